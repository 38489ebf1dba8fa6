// eleven_host.hpp -- C++ host mirror of the reference's scene/render objects on top of the C ABI.
//
// The reference is a C++ program; this header gives a reference-shaped host API (same class and member
// names, same call order) whose device side is libeleven_hip.so instead of SYCL:
//
//   reference                                            here
//   Vector3 / Camera / Material / Texture / HDRI / Tri   same names, plain data (src/Vector.h, Camera.h:5-25,
//   MeshObject / PointLight / Scene                       Material.h:6-56, Texture.h:12-73, HDRI.h:9-42, Tri.h:8-21,
//                                                         MeshObject.hpp:14-22, Scene.h:24-73)
//   RenderParameters (src/kernel.h:51-69)                same fields (+ max_bounces, rank, world)
//   RenderingManager::start_rendering / get_pass /       same methods (src/Managers.h:41-66, Managers.cpp:211-302)
//     get_render_info
//
// Only what the per-sample path consumes is mirrored (OBJ ingest: eleven_obj.hpp beside this file; denoise(): the
// library's own filter in place of DenoiseManager's OIDN call); commands and TCP stay the reference's.  Errors surface as std::runtime_error carrying er_last_error(); nothing falls back to the CPU.
#pragma once
#include <mutex>
#include <thread>
#include <algorithm>
#include <cctype>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/eleven_hip.h"

namespace eleven {

struct Vector3 {
    float x = 0, y = 0, z = 0;
    Vector3() = default;
    Vector3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit Vector3(float v) : x(v), y(v), z(v) {}
};

struct Camera {   // defaults of src/Camera.h:9-19
    float focalLength = 35 * 0.001, sensorWidth = 36 * 0.001, sensorHeight = 24 * 0.001, aperture = 2.8, focusDistance = 1000000;
    Vector3 rotation;
    bool bokeh = false;
    Vector3 position;
};

struct Material {   // defaults of src/Material.h:20-47
    std::string name;
    int albedoTextureID = -1, emissionTextureID = -1, roughnessTextureID = -1, metallicTextureID = -1,
        normalTextureID = -1, opacityTextureID = -1, transmissionTextureID = -1, albedoShaderID = -1;
    Vector3 albedo{0.5f, 0.5f, 0.5f}, emission;
    float opacity = 1, roughness = 1, metallic = 0, clearcoatGloss = 0, clearcoat = 0, anisotropic = 0, eta = 0,
          transmission = 0, specular = 0.5, specularTint = 0, sheenTint = 0.5, subsurface = 0, sheen = 0, ax = 0, ay = 0;
    static Material DefaultMaterial() { Material m; m.name = "default"; return m; }
};

struct Texture {
    enum class Filter { NO_FILTER, BILINEAR };
    std::string name;
    int width = 1, height = 1;
    unsigned channels = 3;
    Filter filter = Filter::NO_FILTER;
    std::vector<float> data{0.5f, 0.5f, 0.5f};
    Texture() = default;
    explicit Texture(Vector3 c) : data{c.x, c.y, c.z} {}
    Texture(std::string n, int w, int h, int ch, std::vector<float> d, Filter f = Filter::NO_FILTER)
        : name(std::move(n)), width(w), height(h), channels((unsigned)ch), filter(f), data(std::move(d)) {}
};

struct HDRI {   // HDRI() = 1x1 texel (0.5,0.5,0.5), src/HDRI.cpp:18; the CDF is built by the library (src/HDRI.cpp:62-83)
    Texture texture{Vector3(0.5f)};
    HDRI() = default;
    explicit HDRI(Texture t) : texture(std::move(t)) {}
};

struct Tri {
    Vector3 vertices[3], uv[3], normals[3], tangents[3];
    float tangentsSign = 1;
    int objectID = 0, materialID = 0;
    std::string matName;
};

struct MeshObject {
    std::string name;
    std::vector<Tri> tris;
    int objectID = 0;
};

struct PointLight { Vector3 position, radiance; };

class Scene {
public:
    std::vector<Material> materials{Material::DefaultMaterial()};   // Scene() pushes the default material, src/Scene.h:45
    std::vector<MeshObject> meshObjects;
    std::vector<Texture> textures;
    std::vector<Tri> tris;
    std::vector<PointLight> pointLights;
    unsigned x_res = 1280, y_res = 720;
    HDRI hdri;
    Camera camera;

    void addPointLight(const PointLight& p) { pointLights.push_back(p); }
    void addTexture(const Texture& t) {   // by name, first one wins (src/Scene.cpp:44-51)
        for (const Texture& e : textures) if (e.name == t.name) return;
        textures.push_back(t);
    }
    void addMaterial(const Material& m) { materials.push_back(m); }
    void addMeshObject(MeshObject m) {     // src/Scene.cpp:54-71
        m.objectID = (int)meshObjects.size();
        for (Tri& t : m.tris) { t.objectID = m.objectID; tris.push_back(t); }
        meshObjects.push_back(std::move(m));
    }
    void addHDRI(const HDRI& h) { hdri = h; }
    void pair_materials() {                // src/Scene.cpp:104-120: materialID by name, 0 if none
        for (Tri& t : tris) {
            t.materialID = 0;
            for (size_t j = 0; j < materials.size(); j++) if (materials[j].name == t.matName) t.materialID = (int)j;
        }
    }
};

struct RenderParameters {   // src/kernel.h:51-69
    unsigned width = 1280, height = 720, sampleTarget = 100, block_size = 8;
    std::string device;     // "name|platform" (src/Managers.cpp:201); empty = device 0
    bool denoise = false;
    unsigned max_bounces = 5, rank = 0, world = 1, flags = 0;
    // Several GPUs of this node driven by this one process (north_star: pixel tiles shard over the GPUs of a node, one combine
    // per read-back): `gpus` ranks, rank r on devices[r] (default: ordinals 0 .. gpus-1 after `device`; an ordinal may repeat --
    // several ranks on one GPU, which is how a one-GPU box tests this path).  transport: "rccl" (ncclSend/ncclRecv over xGMI),
    // "local" (in-process peer copies) or "auto" = rccl when every rank has its own device and RCCL loads, else local.
    unsigned gpus = 1;
    std::vector<int> devices;
    std::string transport = "auto";
};

inline int parsePass(std::string s) {   // src/kernel.cpp:50-73: unknown names -> BEAUTY
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    if (s == "denoise") return ER_PASS_DENOISE;
    if (s == "normal") return ER_PASS_NORMAL;
    if (s == "tangent") return ER_PASS_TANGENT;
    if (s == "bitangent") return ER_PASS_BITANGENT;
    return ER_PASS_BEAUTY;
}

class RenderingManager {
public:
    struct RenderInfo { unsigned samples = 0; };
    RenderParameters pars;
    std::string transport_used;      // "" (one GPU), "rccl" or "in-process"

    ~RenderingManager() { release(); }

    void start_rendering(Scene* scene) {   // src/Managers.cpp:234-275 (without spawning the render thread)
        release();
        size_t n = scene->tris.size();
        std::vector<float> v(n * 9), nn(n * 9), tt(n * 9), uv(n * 6), sign(n);
        std::vector<int32_t> mat(n);
        for (size_t i = 0; i < n; i++) {
            const Tri& t = scene->tris[i];
            for (int k = 0; k < 3; k++) {
                const Vector3* src[3] = {&t.vertices[k], &t.normals[k], &t.tangents[k]};
                float* dst[3] = {&v[i * 9 + k * 3], &nn[i * 9 + k * 3], &tt[i * 9 + k * 3]};
                for (int a = 0; a < 3; a++) { dst[a][0] = src[a]->x; dst[a][1] = src[a]->y; dst[a][2] = src[a]->z; }
                uv[i * 6 + k * 2] = t.uv[k].x;
                uv[i * 6 + k * 2 + 1] = t.uv[k].y;
            }
            sign[i] = t.tangentsSign;
            mat[i] = t.materialID;
        }
        std::vector<ErMaterial> ms;
        for (const Material& m : scene->materials) {
            ErMaterial e{};
            e.albedo_tex = m.albedoTextureID; e.emission_tex = m.emissionTextureID; e.roughness_tex = m.roughnessTextureID;
            e.metallic_tex = m.metallicTextureID; e.normal_tex = m.normalTextureID; e.opacity_tex = m.opacityTextureID;
            e.transmission_tex = m.transmissionTextureID; e.albedo_shader_id = m.albedoShaderID;
            e.albedo = {m.albedo.x, m.albedo.y, m.albedo.z}; e.emission = {m.emission.x, m.emission.y, m.emission.z};
            e.opacity = m.opacity; e.roughness = m.roughness; e.metallic = m.metallic; e.clearcoat_gloss = m.clearcoatGloss;
            e.clearcoat = m.clearcoat; e.anisotropic = m.anisotropic; e.eta = m.eta; e.transmission = m.transmission;
            e.specular = m.specular; e.specular_tint = m.specularTint; e.sheen_tint = m.sheenTint; e.subsurface = m.subsurface;
            e.sheen = m.sheen; e.ax = m.ax; e.ay = m.ay;
            ms.push_back(e);
        }
        std::vector<ErTexture> ts;
        for (const Texture& t : scene->textures)
            ts.push_back(ErTexture{t.width, t.height, (int32_t)t.channels, t.filter == Texture::Filter::BILINEAR ? 1 : 0, t.data.data()});
        ErSceneDesc d{};
        d.tri_count = (uint32_t)n;
        d.vertices = v.data(); d.normals = nn.data(); d.tangents = tt.data(); d.uvs = uv.data();
        d.tangent_sign = sign.data(); d.material_id = mat.data();
        d.material_count = (uint32_t)ms.size(); d.materials = ms.data();
        d.texture_count = (uint32_t)ts.size(); d.textures = ts.data();
        const Texture& ht = scene->hdri.texture;
        d.hdri.texture = ErTexture{ht.width, ht.height, (int32_t)ht.channels, ht.filter == Texture::Filter::BILINEAR ? 1 : 0, ht.data.data()};
        const Camera& c = scene->camera;
        d.camera = ErCamera{c.focalLength, c.sensorWidth, c.sensorHeight, c.aperture, c.focusDistance,
                            {c.rotation.x, c.rotation.y, c.rotation.z}, c.bokeh ? 1 : 0, {c.position.x, c.position.y, c.position.z}};
        std::vector<ErPointLight> pls;      // evaluated only with ER_FLAG_POINT_LIGHTS (src/PointLight.h:4-16; no reference command loads one)
        for (const PointLight& p : scene->pointLights) pls.push_back(ErPointLight{{p.position.x, p.position.y, p.position.z}, {p.radiance.x, p.radiance.y, p.radiance.z}});
        d.point_light_count = (uint32_t)pls.size(); d.point_lights = pls.empty() ? nullptr : pls.data();
        d.x_res = scene->x_res; d.y_res = scene->y_res;
        pars.width = scene->x_res; pars.height = scene->y_res;
        if (er_abi_version() != ER_ABI_VERSION)      // the library fills caller-allocated structs completely: never run against another layout
            throw std::runtime_error("libeleven_hip.so has ABI version " + std::to_string(er_abi_version()) + ", this host was built for " + std::to_string(ER_ABI_VERSION));

        // ---- which ranks on which devices ----
        const bool multi = pars.gpus > 1;
        const unsigned world = multi ? pars.gpus : pars.world;
        if (world == 0 || world > 64) throw std::runtime_error("config gpus out of range (1 .. 64)");
        int first = 0;
        if (!pars.device.empty()) { first = er_device_find(pars.device.c_str()); check(first < 0 ? first : ER_OK); }
        std::vector<int> devs;
        if (multi) {
            devs = pars.devices;
            if (devs.empty()) {
                const int have = er_device_count();
                if (have <= 0) check(ER_ERR_NO_DEVICE, "no gfx950 device is visible");
                if ((int)world > have)
                    throw std::runtime_error("config asks for " + std::to_string(world) + " gpus, " + std::to_string(have) + " visible (name the ordinals in \"devices\" to put several ranks on one GPU)");
                for (unsigned r = 0; r < world; r++) devs.push_back((first + (int)r) % have);
            }
            if (devs.size() != world) throw std::runtime_error("config devices must list one ordinal per gpu");
        }
        // ---- one scene per rank (the scene is replicated, the pixel tiles are dealt (tx + ty) % world); begun side by side ----
        const unsigned ranks = multi ? world : 1;
        ers_.assign(ranks, nullptr);
        std::vector<std::string> errs(ranks);
        std::vector<std::thread> th;
        for (unsigned r = 0; r < ranks; r++)
            th.emplace_back([&, r] {
                ErRenderParams p{};
                p.sample_target = pars.sampleTarget; p.block_size = pars.block_size; p.max_bounces = pars.max_bounces;
                p.rank = multi ? r : pars.rank; p.world = world; p.flags = pars.flags;
                p.device = multi ? devs[r] : first;
                int rc = er_scene_create(&d, &ers_[r]);
                if (rc == ER_OK) rc = er_render_begin(ers_[r], &p);
                if (rc != ER_OK) errs[r] = er_last_error();
            });
        for (auto& t : th) t.join();
        for (unsigned r = 0; r < ranks; r++) if (!errs[r].empty()) { std::string e = errs[r]; release(); throw std::runtime_error(e); }
        // ---- the combine's communicators ----
        if (multi) {
            bool distinct = true;
            for (unsigned a = 0; a < world; a++) for (unsigned b = a + 1; b < world; b++) if (devs[a] == devs[b]) distinct = false;
            std::string tr = pars.transport;
            if (tr != "auto" && tr != "rccl" && tr != "local") throw std::runtime_error("config transport '" + tr + "' not recognised");
            comms_.assign(world, nullptr);
            if (tr == "rccl" && !distinct) { release(); throw std::runtime_error("transport rccl needs one device per rank"); }
            if (tr != "local" && distinct) {
                uint8_t id[ER_COMM_ID_BYTES];
                int rc = er_comm_unique_id(id);
                if (rc == ER_OK) {
                    std::vector<int> rcs(world, ER_OK);
                    std::vector<std::thread> ct;        // ncclCommInitRank blocks until every rank has joined: one thread per rank
                    for (unsigned r = 0; r < world; r++) ct.emplace_back([&, r] { rcs[r] = er_comm_create(id, r, world, devs[r], &comms_[r]); if (rcs[r] != ER_OK) errs[r] = er_last_error(); });
                    for (auto& t : ct) t.join();
                    for (unsigned r = 0; r < world; r++) if (rcs[r] != ER_OK) rc = rcs[r];
                }
                if (rc == ER_OK) transport_used = "rccl";
                else if (tr == "rccl") { std::string e = er_last_error(); for (auto& x : errs) if (!x.empty()) e = x; release(); throw std::runtime_error(e); }
                else for (auto*& cm : comms_) { if (cm) er_comm_destroy(cm); cm = nullptr; }
            }
            if (transport_used.empty()) {
                check(er_comm_create_local(world, comms_.data()));
                transport_used = "in-process";
            }
        }
    }
    // body of kernel_render_enqueue's loop: n more samples on every rank (the launches go out side by side, then the waits).
    // With several ranks a read-back is a SEQUENCE of library calls (gathers, denoise, read) that must see one frame: if the
    // render thread enqueued samples on rank 0 between the gathers and er_denoise, the library would refuse the denoise
    // (its gathered planes are stale, ER_ERR_STATE) and the tiles of a preview would mix sample counts.  frame_mtx_
    // makes render() and such a sequence take turns; the render thread sizes its calls to about 50 ms, so a read-back waits
    // that long at most.  One rank needs none of this: er_read_pass is a sample-boundary snapshot by itself.
    void render(unsigned n_samples) {
        std::unique_lock<std::mutex> lk(frame_mtx_, std::defer_lock);
        if (ers_.size() > 1) lk.lock();
        std::string err;
        for (ErScene* e : ers_) if (er_render_samples_async(e, n_samples) != ER_OK && err.empty()) err = er_last_error();
        for (ErScene* e : ers_) if (er_wait(e, nullptr) != ER_OK && err.empty()) err = er_last_error();
        if (!err.empty()) throw std::runtime_error(err);
    }
    std::vector<float> get_pass(const std::string& pass) {   // src/Managers.cpp:287-302
        if (ers_.empty()) throw std::runtime_error("get_pass: no render has been started");
        std::unique_lock<std::mutex> lk(frame_mtx_, std::defer_lock);
        if (ers_.size() > 1) lk.lock();
        return get_pass_unlocked(parsePass(pass));
    }
    // DenoiseManager::denoise (src/Managers.cpp:319-343) called OIDN on the host; here the device fills the DENOISE plane --
    // on a sharded frame on rank 0, after BEAUTY and NORMAL have been gathered there
    void denoise(unsigned levels = 0, float colour_sigma = 0) {
        if (ers_.empty()) throw std::runtime_error("denoise: no render has been started");
        std::unique_lock<std::mutex> lk(frame_mtx_, std::defer_lock);
        if (ers_.size() > 1) lk.lock();
        denoise_unlocked(levels, colour_sigma);
    }
    // denoise + read of the DENOISE plane as ONE step against the render thread (get_pass denoise / `denoise: true`)
    std::vector<float> get_denoised(unsigned levels = 0, float colour_sigma = 0) {
        if (ers_.empty()) throw std::runtime_error("denoise: no render has been started");
        std::unique_lock<std::mutex> lk(frame_mtx_, std::defer_lock);
        if (ers_.size() > 1) lk.lock();
        denoise_unlocked(levels, colour_sigma);
        return get_pass_unlocked(ER_PASS_DENOISE);
    }
    RenderInfo get_render_info() {   // src/Managers.cpp:211-232; several ranks: the one that is furthest behind
        RenderInfo i;
        if (ers_.empty()) throw std::runtime_error("get_info: no render has been started");
        i.samples = ~0u;
        for (ErScene* e : ers_) {
            unsigned s = 0;
            check(er_samples_done(e, &s));
            i.samples = std::min(i.samples, s);
        }
        return i;
    }
    unsigned ranks() const { return (unsigned)ers_.size(); }

private:
    std::vector<ErScene*> ers_;
    std::vector<ErComm*> comms_;
    std::mutex frame_mtx_;
    std::vector<float> get_pass_unlocked(int pass) {
        std::vector<float> out((size_t)pars.width * pars.height * 4);
        gather(pass);
        check(er_read_pass(ers_[0], pass, out.data()));
        return out;
    }
    void denoise_unlocked(unsigned levels, float colour_sigma) {
        gather(ER_PASS_BEAUTY);
        gather(ER_PASS_NORMAL);
        check(er_denoise(ers_[0], levels, colour_sigma));
    }
    static void check(int rc, const char* what = nullptr) { if (rc != ER_OK) throw std::runtime_error(what ? what : er_last_error()); }
    // every rank's owned pixels of one plane -> rank 0's plane: one er_gather_pass per rank, side by side (the RCCL sends block until
    // the root has posted its receives, so the ranks cannot take turns on one thread)
    void gather(int pass) {
        if (ers_.size() < 2 || pass == ER_PASS_DENOISE) return;      // (the DENOISE plane is only ever written on rank 0)
        std::vector<std::string> errs(ers_.size());
        std::vector<std::thread> th;
        for (size_t r = 0; r < ers_.size(); r++)
            th.emplace_back([&, r] { if (er_gather_pass(ers_[r], pass, comms_[r], 0) != ER_OK) errs[r] = er_last_error(); });
        for (auto& t : th) t.join();
        for (auto& e : errs) if (!e.empty()) throw std::runtime_error(e);
    }
    void release() {
        for (ErComm* c : comms_) if (c) er_comm_destroy(c);
        comms_.clear();
        for (ErScene* e : ers_) if (e) er_scene_destroy(e);
        ers_.clear();
        transport_used.clear();
    }
};

}  // namespace eleven
