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

    ~RenderingManager() { if (er_) er_scene_destroy(er_); }

    void start_rendering(Scene* scene) {   // src/Managers.cpp:234-275 (without spawning the render thread)
        if (er_) { er_scene_destroy(er_); er_ = nullptr; }
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
        check(er_scene_create(&d, &er_));
        ErRenderParams p{};
        p.sample_target = pars.sampleTarget; p.block_size = pars.block_size; p.max_bounces = pars.max_bounces;
        p.rank = pars.rank; p.world = pars.world; p.flags = pars.flags;
        p.device = 0;
        if (!pars.device.empty()) { int dev = er_device_find(pars.device.c_str()); check(dev < 0 ? dev : ER_OK); p.device = dev; }
        check(er_render_begin(er_, &p));
    }
    void render(unsigned n_samples) { check(er_render_samples(er_, n_samples)); }   // body of kernel_render_enqueue's loop
    std::vector<float> get_pass(const std::string& pass) {   // src/Managers.cpp:287-302
        std::vector<float> out((size_t)pars.width * pars.height * 4);
        if (!er_) throw std::runtime_error("get_pass: no render has been started");
        check(er_read_pass(er_, parsePass(pass), out.data()));
        return out;
    }
    // DenoiseManager::denoise (src/Managers.cpp:319-343) called OIDN on the host; here the device fills the DENOISE plane
    void denoise(unsigned levels = 0, float colour_sigma = 0) { check(er_denoise(er_, levels, colour_sigma)); }
    RenderInfo get_render_info() {   // src/Managers.cpp:211-232
        RenderInfo i;
        check(er_samples_done(er_, &i.samples));
        return i;
    }

private:
    ErScene* er_ = nullptr;
    static void check(int rc) { if (rc != ER_OK) throw std::runtime_error(er_last_error()); }
};

}  // namespace eleven
