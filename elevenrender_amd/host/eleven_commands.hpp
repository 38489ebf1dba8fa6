// eleven_commands.hpp -- the reference's command layer re-hosted without Boost / SYCL on top of the C ABI
// (SURVEY.md 8(f) rank 3).  Reference: command grammar and the follow-up message protocol src/main.cpp:36-187, JSON ->
// objects src/CommandManager.cpp:8-236, CommandManager::execute_input_command and the load_* / start_render / get_pass /
// get_render_info / get_sycl_info handlers :250-504, the session loop src/main.cpp:190-238.
//
// What stays exactly as the plug-in expects it:
//   * a COMMAND message carries a command line; options are `--name [values]` (the reference parses them with
//     boost::program_options): --load_config --load_texture --load_object --load_camera --load_hdri
//     --load_brdf_material --start --get_info --get_sycl_info --get_pass NAME, modifiers --recompute_normals
//     --mirror_x --mirror_y (accepted and refused as "not implemented", as in the reference: --path, --output, --pause,
//     --abort, --load_osl_material, --sm);
//   * load commands are followed by their DATA messages: camera / config / material = one JSON message; texture /
//     HDRI = metadata JSON then the float payload; object = OBJ text then MTL text (src/main.cpp:137-164);
//   * every load and --start is answered with STATUS "ok"; --get_info with DATA/JSON {"samples": n}; --get_sycl_info
//     with DATA/JSON {"devices": [...]} (same keys, src/CommandManager.cpp:303-362); --get_pass with DATA/FLOAT4 of
//     x_res * y_res * 16 bytes.
// What is different, on purpose (SURVEY.md section 5 and 8(f) "add real error replies and joinable render thread"):
//   * a failed command is answered with STATUS "error: <text>" instead of "ok" (the reference logs and replies ok);
//   * the render thread is joinable and is stopped and joined when a render is restarted or the session ends (the
//     reference never joins it, src/Managers.h:58);
//   * get_pass returns a sample-boundary snapshot (er_read_pass), not a torn read of a second queue;
//   * `denoise: true` runs the library's own edge-avoiding filter (er_denoise) instead of OIDN;
//   * sRGB textures are converted with the real transfer function -- the reference's fast_pow is broken for float
//     (src/Math.hpp:12-20 zeroes every value above 0.04045), which cannot be intended;
//   * config accepts the optional keys max_bounces (default 5), point_lights / mis (false), schedule, and gpus / devices /
//     transport: the frame's pixel tiles are dealt to `gpus` GPUs of this node driven by this one process, and --get_pass
//     gathers the plane to the first of them (er_gather_pass: RCCL send/recv over xGMI, or in-process peer copies).
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <iomanip>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>

#include "eleven_host.hpp"
#include "eleven_net.hpp"
#include "eleven_obj.hpp"

namespace eleven {

// ---- command line -> options (src/main.cpp:12-24, 36-98) ----
inline std::vector<std::string> str_to_argv(const std::string& str) {
    std::vector<std::string> argv;
    std::istringstream iss(str);
    std::string s;
    while (iss >> std::quoted(s)) argv.push_back(s);
    return argv;
}

struct CommandLine {
    std::map<std::string, std::vector<std::string>> opts;
    bool count(const std::string& k) const { return opts.count(k) != 0; }
    std::string value(const std::string& k) const {
        auto it = opts.find(k);
        if (it == opts.end() || it->second.empty()) throw std::runtime_error("option --" + k + " needs a value");
        return it->second[0];
    }
};

inline CommandLine parse_command_line(const std::string& text) {
    // name -> number of values: 0 flag, 1 one value, -1 any number (multitoken)
    static const std::map<std::string, int> known = {
        {"load_config", 0}, {"load_texture", 0}, {"load_object", 0}, {"load_camera", 0}, {"load_hdri", 0}, {"load_brdf_material", 0},
        {"load_osl_material", 0}, {"start", 0}, {"pause", 0}, {"abort", 0}, {"help", 0}, {"path", -1}, {"recompute_normals", 0},
        {"mirror_x", 0}, {"mirror_y", 0}, {"output", 1}, {"get_info", 0}, {"get_sycl_info", 0}, {"get_pass", 1}, {"sm", 0}};
    CommandLine cl;
    std::vector<std::string> argv = str_to_argv(text);
    for (size_t i = 0; i < argv.size(); i++) {
        const std::string& a = argv[i];
        if (a.size() < 3 || a[0] != '-' || a[1] != '-') throw std::runtime_error("unexpected token '" + a + "' in command");
        std::string name = a.substr(2), inline_value;
        size_t eq = name.find('=');
        if (eq != std::string::npos) { inline_value = name.substr(eq + 1); name = name.substr(0, eq); }
        auto k = known.find(name);
        if (k == known.end()) throw std::runtime_error("unrecognised option '--" + name + "'");
        std::vector<std::string>& vals = cl.opts[name];
        if (eq != std::string::npos) vals.push_back(inline_value);
        if (k->second == 1 && vals.empty()) {
            if (i + 1 >= argv.size()) throw std::runtime_error("option --" + name + " needs a value");
            vals.push_back(argv[++i]);
        } else if (k->second == -1) {
            while (i + 1 < argv.size() && argv[i + 1].compare(0, 2, "--") != 0) vals.push_back(argv[++i]);
        }
    }
    return cl;
}

// ---- JSON -> objects (src/CommandManager.cpp:8-112, 154-172) ----
inline Camera parse_camerajson(const json::Value& j) {
    Camera c;
    const json::Value &p = j.at("position"), &r = j.at("rotation");
    c.aperture = (float)j.at("aperture").as_double();
    c.bokeh = j.at("bokeh").as_bool();
    c.focusDistance = (float)j.at("focus_distance").as_double();
    c.focalLength = (float)j.at("focal_length").as_double();
    c.sensorWidth = (float)j.at("sensor_width").as_double();
    c.sensorHeight = (float)j.at("sensor_height").as_double();
    c.position = Vector3((float)p.at("x").as_double(), (float)p.at("y").as_double(), (float)p.at("z").as_double());
    c.rotation = Vector3((float)r.at("x").as_double(), (float)r.at("y").as_double(), (float)r.at("z").as_double());
    return c;
}

inline float sRGBToLinear(float s) {   // src/Texture.cpp:137-144 with a working pow
    return s <= 0.04045f ? s / 12.92f : std::pow((s + 0.055f) / 1.055f, 2.4f);
}

inline Texture parse_texturejson(const json::Value& meta, const float* data, size_t float_count) {
    const long long w = meta.at("width").as_int64(), h = meta.at("height").as_int64(), ch = meta.at("channels").as_int64();
    if (w <= 0 || h <= 0 || ch <= 0 || w > 65536 || h > 65536 || ch > 16) throw std::runtime_error("texture metadata out of range");
    if ((unsigned long long)w * h * ch != float_count)
        throw std::runtime_error("texture payload has " + std::to_string(float_count) + " floats, metadata says " + std::to_string(w * h * ch));
    const std::string& cs = meta.at("color_space").as_string();
    if (cs != "LINEAR" && cs != "sRGB") throw std::runtime_error("texture color_space '" + cs + "' not recognised");
    std::vector<float> d(data, data + float_count);
    if (cs == "sRGB") for (float& v : d) v = sRGBToLinear(v);
    return Texture(meta.at("name").as_string(), (int)w, (int)h, (int)ch, std::move(d), Texture::Filter::NO_FILTER);   // always NO_FILTER, :41
}

inline void texture_mirror_x(Texture& t) {   // src/Texture.cpp:60-72
    std::vector<float> n(t.data.size());
    for (int x = 0; x < t.width; x++) for (int y = 0; y < t.height; y++) for (unsigned c = 0; c < t.channels; c++)
        n[t.channels * (y * t.width + x) + c] = t.data[t.channels * (y * t.width + (t.width - x - 1)) + c];
    t.data.swap(n);
}
inline void texture_mirror_y(Texture& t) {   // :74-86
    std::vector<float> n(t.data.size());
    for (int x = 0; x < t.width; x++) for (int y = 0; y < t.height; y++) for (unsigned c = 0; c < t.channels; c++)
        n[t.channels * (y * t.width + x) + c] = t.data[t.channels * ((t.height - y - 1) * t.width + x) + c];
    t.data.swap(n);
}
inline void texture_pixel_shift(Texture& t, float x_amount, float y_amount) {   // :115-129
    std::vector<float> n(t.data.size());
    for (int x = 0; x < t.width; x++) {
        const int sx = (int)(x + t.width * x_amount) % t.width;
        for (int y = 0; y < t.height; y++) {
            const int sy = (int)(y + t.height * y_amount) % t.height;
            for (unsigned c = 0; c < t.channels; c++) n[t.channels * (sy * t.width + sx) + c] = t.data[t.channels * (y * t.width + x) + c];
        }
    }
    t.data.swap(n);
}

struct MaterialMaps { std::string albedo, emission, roughness, metallic, normal, opacity, transmission; };

inline Material parse_materialjson(const json::Value& j, MaterialMaps& maps) {   // src/CommandManager.cpp:52-112
    Material m;
    auto num = [&](const char* k, float& dst) { if (const json::Value* v = j.if_contains(k)) dst = (float)v->as_double(); };
    auto rgb = [&](const char* k, Vector3& dst) {
        if (const json::Value* v = j.if_contains(k)) dst = Vector3((float)v->at("r").as_double(), (float)v->at("g").as_double(), (float)v->at("b").as_double());
    };
    auto str = [&](const char* k, std::string& dst) { if (const json::Value* v = j.if_contains(k)) dst = v->as_string(); };
    str("name", m.name);
    rgb("albedo", m.albedo);
    rgb("emission", m.emission);
    num("roughness", m.roughness);
    num("metalness", m.metallic);
    num("specular", m.specular);
    num("opacity", m.opacity);
    num("transmission", m.transmission);
    // (the remaining Disney scalars have no key in the reference's parser; accepted here under their member names)
    num("clearcoat", m.clearcoat); num("clearcoat_gloss", m.clearcoatGloss); num("anisotropic", m.anisotropic);
    num("specular_tint", m.specularTint); num("sheen", m.sheen); num("sheen_tint", m.sheenTint); num("subsurface", m.subsurface);
    str("albedo_map", maps.albedo); str("emission_map", maps.emission); str("roughness_map", maps.roughness);
    str("metallic_map", maps.metallic); str("normal_map", maps.normal); str("opacity_map", maps.opacity);
    str("transmission_map", maps.transmission);
    if (const json::Value* v = j.if_contains("albedo_shader_id")) m.albedoShaderID = (int)v->as_int64();
    const float aspect = (float)std::sqrt(1.0 - m.anisotropic * 0.9);           // :108-110
    m.ax = std::max(0.001f, m.roughness / aspect);
    m.ay = std::max(0.001f, m.roughness * aspect);
    return m;
}

// ---- the session's managers (src/CommandManager.h:176-209, src/Managers.h:41-98) ----
class CommandManager {
public:
    TCPInterface* im = nullptr;
    Scene scene;
    RenderingManager rm;
    std::vector<MaterialMaps> material_maps{MaterialMaps()};   // parallel to scene.materials (index 0 = the default material)

    ~CommandManager() { stop_render_thread(); }

    // One COMMAND message: read its follow-up DATA messages (always, so that the stream stays in step even when the
    // command then fails), execute, reply.  Returns false when the peer closed the connection in the middle.
    bool execute(const Message& msg) {
        CommandLine cl;
        std::vector<Message> extra;
        try {
            cl = parse_command_line(msg.get_string_data());
            int follow = 0;
            if (!cl.count("path") && !cl.count("sm")) {
                if (cl.count("load_camera") || cl.count("load_config") || cl.count("load_brdf_material")) follow = 1;
                if (cl.count("load_texture") || cl.count("load_hdri") || cl.count("load_object")) follow = 2;
            }
            for (int i = 0; i < follow; i++) {
                extra.push_back(im->read_message());
                if (im->error) return false;
            }
        } catch (const std::exception& e) {
            im->write_message(Message::Error(e.what()));
            return true;
        }
        try {
            dispatch(cl, extra);
        } catch (const std::exception& e) {
            im->write_message(Message::Error(e.what()));
        }
        return true;
    }

    void stop_render_thread() {
        stop_ = true;
        if (t_rend_.joinable()) t_rend_.join();
        stop_ = false;
    }

private:
    std::thread t_rend_;
    std::atomic<bool> stop_{false};
    std::mutex err_mtx_;
    std::string render_error_;
    std::atomic<unsigned> samples_per_call_{0};      // what the render thread's last call was sized to (reported by --get_info)

    void dispatch(const CommandLine& cl, std::vector<Message>& extra) {
        if (cl.count("path") || cl.count("sm") || cl.count("output"))
            throw std::runtime_error("loading from a filesystem path / shared memory is not implemented (neither is it in the reference: src/CommandManager.cpp:116-150)");
        if (cl.count("load_camera")) { scene.camera = parse_camerajson(extra[0].get_json_data()); return ok(); }
        if (cl.count("load_texture")) {
            Texture t = parse_texturejson(extra[0].get_json_data(), extra[1].get_float_data(), extra[1].float_count());
            scene.addTexture(t);                                            // load_texture, src/CommandManager.cpp:364-370
            pair_textures();
            return ok();
        }
        if (cl.count("load_config")) return load_config(extra[0].get_json_data());
        if (cl.count("load_hdri")) {                                         // src/CommandManager.cpp:178-193
            Texture t = parse_texturejson(extra[0].get_json_data(), extra[1].get_float_data(), extra[1].float_count());
            if (cl.count("mirror_x")) texture_mirror_x(t);
            if (cl.count("mirror_y")) texture_mirror_y(t);
            texture_pixel_shift(t, 0.5f, 0.0f);
            scene.addHDRI(HDRI(t));
            return ok();
        }
        if (cl.count("load_brdf_material")) {                                // :393-398
            MaterialMaps maps;
            scene.addMaterial(parse_materialjson(extra[0].get_json_data(), maps));
            material_maps.push_back(maps);
            scene.pair_materials();
            pair_textures();
            return ok();
        }
        if (cl.count("load_object")) {                                       // :212-227, 489-498
            // the MTL message only contributes material NAMES there (every line that does not start with `newmtl` is
            // stripped before rapidobj sees it, src/ObjLoader.cpp:163-164): the names come with `usemtl` already
            std::istringstream obj(std::string(extra[0].data.data(), extra[0].data.size()));
            for (MeshObject& mo : load_obj(obj, cl.count("recompute_normals"))) scene.addMeshObject(std::move(mo));
            scene.pair_materials();
            return ok();
        }
        if (cl.count("start")) return start_render();
        if (cl.count("get_info")) {                                          // :282-300
            json::Value j = json::Value::object();
            j["samples"] = rm.get_render_info().samples;
            // (extra keys, ignored by the plug-in: how the render is spread and how the render thread sizes its calls)
            j["gpus"] = rm.ranks();
            if (!rm.transport_used.empty()) j["transport"] = rm.transport_used;
            j["samples_per_call"] = samples_per_call_.load();
            im->write_message(Message::json_data(j));
            return;
        }
        if (cl.count("get_sycl_info")) return get_device_info();
        if (cl.count("get_pass")) return get_pass(cl.value("get_pass"));
        if (cl.count("load_osl_material") || cl.count("pause") || cl.count("abort") || cl.count("help"))
            throw std::runtime_error("command accepted by the grammar but not implemented (as in the reference)");
        throw std::runtime_error("input command not recognised");           // src/main.cpp:178-180
    }

    void ok() { im->write_message(Message::OK()); }

    void pair_textures() {   // Scene::pair_textures, src/Scene.cpp:75-102 (transmission is not paired there either)
        for (size_t i = 0; i < scene.materials.size() && i < material_maps.size(); i++)
            for (size_t j = 0; j < scene.textures.size(); j++) {
                const std::string& n = scene.textures[j].name;
                if (n.empty()) continue;
                Material& m = scene.materials[i];
                const MaterialMaps& mm = material_maps[i];
                if (n == mm.albedo) m.albedoTextureID = (int)j;
                if (n == mm.emission) m.emissionTextureID = (int)j;
                if (n == mm.roughness) m.roughnessTextureID = (int)j;
                if (n == mm.metallic) m.metallicTextureID = (int)j;
                if (n == mm.opacity) m.opacityTextureID = (int)j;
                if (n == mm.normal) m.normalTextureID = (int)j;
            }
    }

    void load_config(const json::Value& j) {   // src/CommandManager.cpp:154-172, 378-385
        RenderParameters rp;
        rp.width = (unsigned)j.at("x_res").as_int64();
        rp.height = (unsigned)j.at("y_res").as_int64();
        rp.sampleTarget = (unsigned)j.at("sample_target").as_int64();
        rp.denoise = j.at("denoise").as_bool();
        rp.device = j.at("device").as_string();
        rp.block_size = (unsigned)j.at("block_size").as_int64();
        if (rp.width == 0 || rp.height == 0 || rp.width > 65536 || rp.height > 65536) throw std::runtime_error("config resolution out of range");
        if (const json::Value* v = j.if_contains("max_bounces")) rp.max_bounces = (unsigned)v->as_int64();
        if (const json::Value* v = j.if_contains("point_lights")) if (v->as_bool()) rp.flags |= ER_FLAG_POINT_LIGHTS;
        if (const json::Value* v = j.if_contains("mis")) if (v->as_bool()) rp.flags |= ER_FLAG_MIS;
        if (const json::Value* v = j.if_contains("schedule")) {
            const std::string& s = v->as_string();
            if (s == "stream") rp.flags |= ER_FLAG_STREAM;
            else if (s == "wavefront") rp.flags |= ER_FLAG_WAVEFRONT;
            else if (s == "fused") rp.flags |= ER_FLAG_FUSED;
            else if (s == "megakernel") rp.flags |= ER_FLAG_MEGAKERNEL;
            else if (s != "auto") throw std::runtime_error("config schedule '" + s + "' not recognised");
        }
        // which builder makes the acceleration structure (the library's default: the device build from 20 000 triangles up)
        if (const json::Value* v = j.if_contains("builder")) {
            const std::string& s = v->as_string();
            if (s == "host") rp.flags |= ER_FLAG_HOST_BUILD;
            else if (s == "device") rp.flags |= ER_FLAG_GPU_BUILD;
            else if (s != "auto") throw std::runtime_error("config builder '" + s + "' not recognised");
        }
        // several GPUs of this node behind the one session (SURVEY.md section 5 planned `gpus` beside max_bounces; reference hook
        // src/CommandManager.cpp:154-172): "gpus": N, optionally "devices": [ordinals] and "transport": "auto" | "rccl" | "local"
        if (const json::Value* v = j.if_contains("gpus")) {
            const long long g = v->as_int64();
            if (g < 1 || g > 64) throw std::runtime_error("config gpus out of range (1 .. 64)");
            rp.gpus = (unsigned)g;
        }
        if (const json::Value* v = j.if_contains("devices")) {
            for (const json::Value& o : v->as_array()) rp.devices.push_back((int)o.as_int64());
            if (rp.devices.size() != rp.gpus) throw std::runtime_error("config devices must list one ordinal per gpu");
        }
        if (const json::Value* v = j.if_contains("transport")) rp.transport = v->as_string();
        stop_render_thread();
        rm.pars = rp;
        scene.x_res = rp.width;
        scene.y_res = rp.height;
        ok();
    }

    void start_render() {   // src/CommandManager.cpp:500-504 -> RenderingManager::start_rendering, src/Managers.cpp:234-275
        stop_render_thread();
        { std::lock_guard<std::mutex> lk(err_mtx_); render_error_.clear(); }
        rm.start_rendering(&scene);                     // builds the BVH, uploads, runs setupKernel; throws with er_last_error()
        const unsigned target = rm.pars.sampleTarget;
        t_rend_ = std::thread([this, target] {          // kernel_render_enqueue, src/kernel.cpp:680-706: `target` samples
            try {
                // One sample first (the plug-in's preview wants a first pass soon), then calls sized by TIME: a call of the
                // streaming schedule costs about a millisecond of start-up and tail whatever its length, so each call is given
                // about 50 ms of work -- long enough to lose ~2 % to that, short enough that --get_pass (a sample-boundary
                // snapshot, ordered behind the call in flight) and a restart answer within a frame or two of the viewer.
                unsigned done = 0, n = 1;
                while (done < target && !stop_) {
                    n = std::min(n, target - done);
                    const auto t0 = std::chrono::steady_clock::now();
                    rm.render(n);
                    const double per_sample = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / n;
                    done += n;
                    samples_per_call_ = n;
                    n = (unsigned)std::min(256.0, std::max(1.0, std::ceil(0.050 / std::max(per_sample, 1e-6))));
                }
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> lk(err_mtx_);
                render_error_ = e.what();
            }
        });
        ok();
    }

    void get_pass(const std::string& pass) {   // src/CommandManager.cpp:250-280
        {
            std::lock_guard<std::mutex> lk(err_mtx_);
            if (!render_error_.empty()) throw std::runtime_error("render thread failed: " + render_error_);
        }
        std::vector<float> img;
        const bool want_denoise = parsePass(pass) == ER_PASS_DENOISE || (rm.pars.denoise && parsePass(pass) == ER_PASS_BEAUTY);
        if (want_denoise) {
            img = rm.get_denoised();                    // gathers, fills the DENOISE plane from the current BEAUTY + NORMAL planes and reads it: one step against the render thread
            for (size_t i = 3; i < img.size(); i += 4) img[i] = 1.0f;   // :270-272
        } else {
            img = rm.get_pass(pass);
        }
        im->write_message(Message::float_data(img.data(), img.size(), Message::DataFormat::FLOAT4));
    }

    void get_device_info() {   // get_sycl_info, src/CommandManager.cpp:303-362
        json::Value devices = json::Value::array();
        const int n = er_device_count();
        for (int i = 0; i < n; i++) {
            ErDeviceInfo info;
            if (er_device_info(i, &info) != ER_OK) continue;
            json::Value d = json::Value::object();
            d["name"] = std::string(info.name);
            d["platform"] = std::string(info.platform);
            d["memory"] = (unsigned long long)info.memory_bytes;
            d["max_compute_units"] = info.compute_units;
            d["is_compatible"] = info.compatible != 0;
            d["online_compiler"] = false;
            d["type"] = "gpu";
            devices.push_back(d);
        }
        json::Value j = json::Value::object();
        j["devices"] = devices;
        im->write_message(Message::json_data(j));
    }
};

// One client session (the body of the reference's accept loop, src/main.cpp:201-234).
inline void serve_session(int connected_fd) {
    TCPInterface tcp(connected_fd);
    CommandManager cm;
    cm.im = &tcp;
    tcp.write_message(Message::OK());                                    // :211
    while (!tcp.error) {
        Message msg;
        try {
            msg = tcp.read_message();
        } catch (const std::exception& e) {                              // malformed header: tell the client, drop the session
            try { tcp.write_message(Message::Error(std::string("bad message: ") + e.what())); } catch (...) {}
            break;
        }
        if (tcp.error) break;
        try {
            if (msg.type == Message::Type::COMMAND) {
                if (!cm.execute(msg)) break;
            } else if (msg.type == Message::Type::STATUS) {
                if (msg.get_string_data() == "close_session") break;    // :223-227
                tcp.write_message(Message::Error("message received, expected a command"));
            } else {
                tcp.write_message(Message::Error("message received, expected a command"));
            }
        } catch (const std::exception&) {
            break;                                                       // the socket itself failed
        }
    }
    cm.stop_render_thread();
}

}  // namespace eleven
