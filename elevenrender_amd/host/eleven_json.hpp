// eleven_json.hpp -- the small JSON the wire protocol needs, dependency-free (the reference uses Boost.JSON:
// Message headers src/Managers.cpp:6-17,167-177; camera / texture / material / config payloads
// src/CommandManager.cpp:8-172; render and device info replies :282-362).
//
// Value model: null, bool, number (double, with an "integral" note so that 3 and 3.0 both satisfy as_int64 /
// as_double the way the plug-in sends them), string, array, object (insertion-ordered).  parse() accepts what
// RFC 8259 defines; text after the value's end is ignored when it is NUL or whitespace (message headers are
// NUL-padded to 1024 bytes, src/TCPInterface.cpp:10-11).  Errors are std::runtime_error with the byte offset.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace eleven {
namespace json {

class Value;
using Member = std::pair<std::string, Value>;

class Value {
public:
    enum class Kind { Null, Bool, Number, String, Array, Object };

    Value() = default;
    Value(std::nullptr_t) {}
    Value(bool b) : kind_(Kind::Bool), b_(b) {}
    Value(double d) : kind_(Kind::Number), n_(d), integral_(false) {}
    Value(int v) : kind_(Kind::Number), n_(v), integral_(true) {}
    Value(unsigned v) : kind_(Kind::Number), n_(v), integral_(true) {}
    Value(long long v) : kind_(Kind::Number), n_((double)v), integral_(true) {}
    Value(unsigned long long v) : kind_(Kind::Number), n_((double)v), integral_(true) {}
    Value(long v) : kind_(Kind::Number), n_((double)v), integral_(true) {}
    Value(unsigned long v) : kind_(Kind::Number), n_((double)v), integral_(true) {}
    Value(const char* s) : kind_(Kind::String), s_(s) {}
    Value(std::string s) : kind_(Kind::String), s_(std::move(s)) {}

    static Value array() { Value v; v.kind_ = Kind::Array; return v; }
    static Value object() { Value v; v.kind_ = Kind::Object; return v; }

    Kind kind() const { return kind_; }
    bool is_null() const { return kind_ == Kind::Null; }
    bool is_object() const { return kind_ == Kind::Object; }
    bool is_array() const { return kind_ == Kind::Array; }
    bool is_string() const { return kind_ == Kind::String; }
    bool is_number() const { return kind_ == Kind::Number; }
    bool is_bool() const { return kind_ == Kind::Bool; }

    // Boost.JSON's as_* throw on a kind mismatch; so do these (the command layer turns that into an error reply).
    // Unlike Boost.JSON a number is a number: {"aperture": 2} satisfies as_double.
    bool as_bool() const { need(Kind::Bool, "bool"); return b_; }
    double as_double() const { need(Kind::Number, "number"); return n_; }
    long long as_int64() const {
        need(Kind::Number, "number");
        if (n_ != std::floor(n_)) throw std::runtime_error("json: integer expected, got " + std::to_string(n_));
        return (long long)n_;
    }
    const std::string& as_string() const { need(Kind::String, "string"); return s_; }
    const std::vector<Value>& as_array() const { need(Kind::Array, "array"); return a_; }
    const std::vector<Member>& as_object() const { need(Kind::Object, "object"); return o_; }

    // object access
    const Value* if_contains(const std::string& key) const {
        if (kind_ != Kind::Object) return nullptr;
        for (const Member& m : o_) if (m.first == key) return &m.second;
        return nullptr;
    }
    const Value& at(const std::string& key) const {
        const Value* v = if_contains(key);
        if (!v) throw std::runtime_error("json: key '" + key + "' is missing");
        return *v;
    }
    Value& operator[](const std::string& key) {          // insert-or-find, like boost::json::object
        if (kind_ == Kind::Null) kind_ = Kind::Object;
        need(Kind::Object, "object");
        for (Member& m : o_) if (m.first == key) return m.second;
        o_.emplace_back(key, Value());
        return o_.back().second;
    }
    void push_back(Value v) {
        if (kind_ == Kind::Null) kind_ = Kind::Array;
        need(Kind::Array, "array");
        a_.push_back(std::move(v));
    }

    std::string serialize() const {
        std::string out;
        write(out);
        return out;
    }

private:
    Kind kind_ = Kind::Null;
    bool b_ = false;
    double n_ = 0;
    bool integral_ = false;
    std::string s_;
    std::vector<Value> a_;
    std::vector<Member> o_;

    void need(Kind k, const char* what) const {
        if (kind_ != k) throw std::runtime_error(std::string("json: ") + what + " expected");
    }
    static void write_string(const std::string& s, std::string& out) {
        out += '"';
        for (unsigned char c : s) {
            switch (c) {
                case '"': out += "\\\""; break;
                case '\\': out += "\\\\"; break;
                case '\n': out += "\\n"; break;
                case '\r': out += "\\r"; break;
                case '\t': out += "\\t"; break;
                case '\b': out += "\\b"; break;
                case '\f': out += "\\f"; break;
                default:
                    if (c < 0x20) { char buf[8]; snprintf(buf, sizeof(buf), "\\u%04x", c); out += buf; }
                    else out += (char)c;
            }
        }
        out += '"';
    }
    void write(std::string& out) const {
        switch (kind_) {
            case Kind::Null: out += "null"; break;
            case Kind::Bool: out += b_ ? "true" : "false"; break;
            case Kind::Number: {
                char buf[40];
                if (integral_ || (n_ == std::floor(n_) && std::fabs(n_) < 9.0e15)) snprintf(buf, sizeof(buf), "%lld", (long long)n_);
                else if (!std::isfinite(n_)) snprintf(buf, sizeof(buf), "null");
                else snprintf(buf, sizeof(buf), "%.17g", n_);
                out += buf;
                break;
            }
            case Kind::String: write_string(s_, out); break;
            case Kind::Array:
                out += '[';
                for (size_t i = 0; i < a_.size(); i++) { if (i) out += ','; a_[i].write(out); }
                out += ']';
                break;
            case Kind::Object:
                out += '{';
                for (size_t i = 0; i < o_.size(); i++) {
                    if (i) out += ',';
                    write_string(o_[i].first, out);
                    out += ':';
                    o_[i].second.write(out);
                }
                out += '}';
                break;
        }
    }
    friend class Parser;
};

class Parser {
public:
    Parser(const char* p, size_t n) : p_(p), n_(n) {}
    Value parse_document() {
        Value v = parse_value(0);
        skip_ws();
        if (i_ < n_ && p_[i_] != '\0') fail("trailing characters after the JSON value");
        return v;
    }

private:
    const char* p_;
    size_t n_, i_ = 0;

    [[noreturn]] void fail(const std::string& what) const { throw std::runtime_error("json: " + what + " at byte " + std::to_string(i_)); }
    void skip_ws() { while (i_ < n_ && (p_[i_] == ' ' || p_[i_] == '\t' || p_[i_] == '\n' || p_[i_] == '\r')) i_++; }
    bool eat(char c) { if (i_ < n_ && p_[i_] == c) { i_++; return true; } return false; }
    void expect_word(const char* w) {
        for (const char* q = w; *q; q++) { if (i_ >= n_ || p_[i_] != *q) fail(std::string("'") + w + "' expected"); i_++; }
    }
    static void append_utf8(unsigned cp, std::string& out) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    unsigned hex4() {
        if (i_ + 4 > n_) fail("truncated \\u escape");
        unsigned v = 0;
        for (int k = 0; k < 4; k++) {
            char c = p_[i_++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else fail("bad hex digit in \\u escape");
        }
        return v;
    }
    std::string parse_string() {
        if (!eat('"')) fail("'\"' expected");
        std::string out;
        while (true) {
            if (i_ >= n_) fail("unterminated string");
            unsigned char c = (unsigned char)p_[i_++];
            if (c == '"') return out;
            if (c < 0x20) fail("control character in string");
            if (c != '\\') { out += (char)c; continue; }
            if (i_ >= n_) fail("unterminated escape");
            char e = p_[i_++];
            switch (e) {
                case '"': out += '"'; break;
                case '\\': out += '\\'; break;
                case '/': out += '/'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'n': out += '\n'; break;
                case 'r': out += '\r'; break;
                case 't': out += '\t'; break;
                case 'u': {
                    unsigned cp = hex4();
                    if (cp >= 0xD800 && cp <= 0xDBFF && i_ + 1 < n_ && p_[i_] == '\\' && p_[i_ + 1] == 'u') {
                        i_ += 2;
                        unsigned lo = hex4();
                        if (lo >= 0xDC00 && lo <= 0xDFFF) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        else fail("unpaired surrogate");
                    }
                    append_utf8(cp, out);
                    break;
                }
                default: fail("unknown escape");
            }
        }
    }
    Value parse_number() {
        size_t start = i_;
        bool integral = true;
        if (i_ < n_ && p_[i_] == '-') i_++;
        if (i_ >= n_ || p_[i_] < '0' || p_[i_] > '9') fail("digit expected");
        if (p_[i_] == '0') i_++;
        else while (i_ < n_ && p_[i_] >= '0' && p_[i_] <= '9') i_++;
        if (i_ < n_ && p_[i_] == '.') {
            integral = false;
            i_++;
            if (i_ >= n_ || p_[i_] < '0' || p_[i_] > '9') fail("digit expected after '.'");
            while (i_ < n_ && p_[i_] >= '0' && p_[i_] <= '9') i_++;
        }
        if (i_ < n_ && (p_[i_] == 'e' || p_[i_] == 'E')) {
            integral = false;
            i_++;
            if (i_ < n_ && (p_[i_] == '+' || p_[i_] == '-')) i_++;
            if (i_ >= n_ || p_[i_] < '0' || p_[i_] > '9') fail("digit expected in exponent");
            while (i_ < n_ && p_[i_] >= '0' && p_[i_] <= '9') i_++;
        }
        std::string tok(p_ + start, i_ - start);
        Value v(strtod(tok.c_str(), nullptr));
        v.integral_ = integral;
        return v;
    }
    Value parse_value(int depth) {
        if (depth > 64) fail("nesting too deep");
        skip_ws();
        if (i_ >= n_) fail("value expected");
        char c = p_[i_];
        if (c == '{') {
            i_++;
            Value v = Value::object();
            skip_ws();
            if (eat('}')) return v;
            while (true) {
                skip_ws();
                std::string key = parse_string();
                skip_ws();
                if (!eat(':')) fail("':' expected");
                Value item = parse_value(depth + 1);
                v.o_.emplace_back(std::move(key), std::move(item));
                skip_ws();
                if (eat(',')) continue;
                if (eat('}')) return v;
                fail("',' or '}' expected");
            }
        }
        if (c == '[') {
            i_++;
            Value v = Value::array();
            skip_ws();
            if (eat(']')) return v;
            while (true) {
                v.a_.push_back(parse_value(depth + 1));
                skip_ws();
                if (eat(',')) continue;
                if (eat(']')) return v;
                fail("',' or ']' expected");
            }
        }
        if (c == '"') return Value(parse_string());
        if (c == 't') { expect_word("true"); return Value(true); }
        if (c == 'f') { expect_word("false"); return Value(false); }
        if (c == 'n') { expect_word("null"); return Value(); }
        return parse_number();
    }
};

inline Value parse(const char* p, size_t n) { return Parser(p, n).parse_document(); }
inline Value parse(const std::string& s) { return parse(s.data(), s.size()); }

}  // namespace json
}  // namespace eleven
