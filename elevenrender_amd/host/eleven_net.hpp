// eleven_net.hpp -- Message + TCPInterface of the reference's wire protocol over plain POSIX sockets
// (reference: Message struct src/Managers.h:101-144, Message:: methods src/Managers.cpp:6-177,
// TCPInterface::write_message / read_message src/TCPInterface.cpp:3-58; Boost.Asio + Boost.JSON there).
//
// Wire format (unchanged, so the Blender plug-in still talks to this host):
//   * every message starts with a MESSAGE_HEADER_SIZE = 1024 byte header: a JSON object
//     {"type": "none|command|status|data", "data_format": "none|float3|float4|string|json", "data_size": N},
//     padded with NUL bytes (src/Managers.h:14, src/TCPInterface.cpp:5-14);
//   * followed by data_size payload bytes when data_size != 0 (:16-21, :44-49).
// Differences from the reference, all on the side of robustness: the payload is owned (std::vector, the reference
// mallocs and leaks), short reads / closed peers / oversized or malformed headers come back as a CloseSession
// message or an exception instead of undefined behaviour, payloads are capped (MAX_PAYLOAD) and string / JSON
// payloads are read by length (the reference reads them with strlen over a buffer it never terminated,
// src/Managers.cpp:138,161).
#pragma once
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "eleven_json.hpp"

namespace eleven {

constexpr size_t MESSAGE_HEADER_SIZE = 1024;            // src/Managers.h:14
constexpr size_t MAX_PAYLOAD = size_t(1) << 32;         // a 4K float4 plane is 133 MB; an .obj of 10M triangles ~1 GB

struct Message {
    enum class Type { NONE, COMMAND, STATUS, DATA };
    enum class DataFormat { NONE, FLOAT3, FLOAT4, STRING, JSON };

    Type type = Type::NONE;
    DataFormat data_format = DataFormat::NONE;
    std::vector<char> data;          // data_size = data.size()

    static Message status(const std::string& text) {
        Message m;
        m.type = Type::STATUS;
        m.data_format = DataFormat::STRING;
        m.data.assign(text.begin(), text.end());
        return m;
    }
    static Message OK() { return status("ok"); }                        // src/Managers.h:113-121
    static Message CloseSession() { return status("close_session"); }   // :123-131
    // not in the reference (it replies OK regardless, src/CommandManager.cpp:500-504): a status the client can test for
    static Message Error(const std::string& what) { return status("error: " + what); }
    static Message command(const std::string& text) { Message m = status(text); m.type = Type::COMMAND; return m; }
    static Message json_data(const json::Value& v) {
        Message m;
        m.type = Type::DATA;
        m.data_format = DataFormat::JSON;
        std::string s = v.serialize();
        m.data.assign(s.begin(), s.end());
        return m;
    }
    static Message float_data(const float* p, size_t count, DataFormat fmt = DataFormat::FLOAT4) {
        Message m;
        m.type = Type::DATA;
        m.data_format = fmt;
        m.data.assign((const char*)p, (const char*)p + count * sizeof(float));
        return m;
    }

    static const char* type2str(Type t) {                                // src/Managers.cpp:43-63
        switch (t) { case Type::COMMAND: return "command"; case Type::STATUS: return "status"; case Type::DATA: return "data"; default: return "none"; }
    }
    static Type str2type(const std::string& s) {                         // :65-83
        if (s == "command") return Type::COMMAND;
        if (s == "status") return Type::STATUS;
        if (s == "data") return Type::DATA;
        return Type::NONE;
    }
    static const char* data_format2str(DataFormat f) {                   // :85-108
        switch (f) { case DataFormat::STRING: return "string"; case DataFormat::JSON: return "json"; case DataFormat::FLOAT3: return "float3";
                     case DataFormat::FLOAT4: return "float4"; default: return "none"; }
    }
    static DataFormat str2data_format(const std::string& s) {            // :110-130
        if (s == "float3") return DataFormat::FLOAT3;
        if (s == "float4") return DataFormat::FLOAT4;
        if (s == "json") return DataFormat::JSON;
        if (s == "string") return DataFormat::STRING;
        return DataFormat::NONE;
    }

    std::string get_string_data() const { return std::string(data.data(), strnlen(data.data(), data.size())); }   // :155-162
    json::Value get_json_data() const { return json::parse(data.data(), data.size()); }                           // :134-148
    const float* get_float_data() const { return (const float*)data.data(); }                                      // :150-153
    size_t float_count() const { return data.size() / sizeof(float); }

    json::Value header() const {                                         // msg2json_header, :165-177
        json::Value h = json::Value::object();
        h["type"] = type2str(type);
        h["data_format"] = data_format2str(data_format);
        h["data_size"] = (unsigned long long)data.size();
        return h;
    }
};

class TCPInterface {
public:
    int fd = -1;
    bool error = false;              // the reference's tcp_interface.error: set when the peer is gone

    TCPInterface() = default;
    explicit TCPInterface(int connected_fd) : fd(connected_fd) {
        int one = 1;
        (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    }
    TCPInterface(const TCPInterface&) = delete;
    TCPInterface& operator=(const TCPInterface&) = delete;
    ~TCPInterface() { close_socket(); }
    void close_socket() { if (fd >= 0) { ::close(fd); fd = -1; } }

    void write_message(const Message& msg) {                             // src/TCPInterface.cpp:3-22
        std::string header = msg.header().serialize();
        if (header.size() > MESSAGE_HEADER_SIZE) throw std::runtime_error("TCP header size exceeded");
        header.resize(MESSAGE_HEADER_SIZE, '\0');
        write_all(header.data(), header.size());
        if (!msg.data.empty() && msg.data_format != Message::DataFormat::NONE) write_all(msg.data.data(), msg.data.size());
    }

    Message read_message() {                                             // src/TCPInterface.cpp:25-58
        char header[MESSAGE_HEADER_SIZE + 1];
        if (!read_all(header, MESSAGE_HEADER_SIZE)) { error = true; return Message::CloseSession(); }   // EOF: :34-37
        header[MESSAGE_HEADER_SIZE] = '\0';
        Message msg;
        json::Value h = json::parse(header, strnlen(header, MESSAGE_HEADER_SIZE));   // Message::json2header, src/Managers.cpp:6-17
        msg.type = Message::str2type(h.at("type").as_string());
        msg.data_format = Message::str2data_format(h.at("data_format").as_string());
        long long size = h.at("data_size").as_int64();
        if (size < 0 || (unsigned long long)size > MAX_PAYLOAD) throw std::runtime_error("message payload of " + std::to_string(size) + " bytes refused");
        if (size > 0) {
            msg.data.resize((size_t)size);
            if (!read_all(msg.data.data(), (size_t)size)) { error = true; return Message::CloseSession(); }
        }
        return msg;
    }

private:
    void write_all(const char* p, size_t n) {
        while (n > 0) {
            ssize_t w = ::send(fd, p, n, MSG_NOSIGNAL);
            if (w < 0) { if (errno == EINTR) continue; error = true; throw std::runtime_error(std::string("socket write: ") + strerror(errno)); }
            p += w;
            n -= (size_t)w;
        }
    }
    bool read_all(char* p, size_t n) {
        while (n > 0) {
            ssize_t r = ::recv(fd, p, n, 0);
            if (r == 0) return false;
            if (r < 0) { if (errno == EINTR) continue; return false; }
            p += r;
            n -= (size_t)r;
        }
        return true;
    }
};

// listening socket on 0.0.0.0:port (the reference: tcp::acceptor on tcp::v4(), 5557 -- src/main.cpp:198); port 0 = any
class Acceptor {
public:
    int fd = -1;
    uint16_t port = 0;
    explicit Acceptor(uint16_t want_port, bool loopback_only = false) {
        fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (fd < 0) throw std::runtime_error(std::string("socket: ") + strerror(errno));
        int one = 1;
        (void)setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in a{};
        a.sin_family = AF_INET;
        a.sin_addr.s_addr = htonl(loopback_only ? INADDR_LOOPBACK : INADDR_ANY);
        a.sin_port = htons(want_port);
        if (::bind(fd, (sockaddr*)&a, sizeof(a)) != 0 || ::listen(fd, 4) != 0) {
            std::string e = strerror(errno);
            ::close(fd);
            fd = -1;
            throw std::runtime_error("bind/listen on port " + std::to_string(want_port) + ": " + e);
        }
        socklen_t len = sizeof(a);
        (void)getsockname(fd, (sockaddr*)&a, &len);
        port = ntohs(a.sin_port);
    }
    Acceptor(const Acceptor&) = delete;
    Acceptor& operator=(const Acceptor&) = delete;
    ~Acceptor() { if (fd >= 0) ::close(fd); }
    int accept_one() {
        while (true) {
            int c = ::accept(fd, nullptr, nullptr);
            if (c >= 0) return c;
            if (errno != EINTR) throw std::runtime_error(std::string("accept: ") + strerror(errno));
        }
    }
};

}  // namespace eleven
