// abi_smoke.cpp -- minimal C++ driver of the C ABI (include/eleven_hip.h): two triangles under a
// constant sky, a few samples, prints a checksum.  Used by tests (GPU) and for debugging.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/eleven_hip.h"

#define CHECK(x) do { int rc__ = (x); if (rc__ != ER_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc__, er_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    int spp = argc > 1 ? atoi(argv[1]) : 4;
    const uint32_t W = 32, H = 24;
    float verts[2][3][3] = {{{-1, -1, 3}, {1, -1, 3}, {1, 1, 3}}, {{-1, -1, 3}, {1, 1, 3}, {-1, 1, 3}}};
    float normals[2][3][3], tangents[2][3][3], uvs[2][3][2] = {{{0, 0}, {1, 0}, {1, 1}}, {{0, 0}, {1, 1}, {0, 1}}};
    for (int t = 0; t < 2; t++) for (int k = 0; k < 3; k++) {
        normals[t][k][0] = 0; normals[t][k][1] = 0; normals[t][k][2] = -1;
        tangents[t][k][0] = 1; tangents[t][k][1] = 0; tangents[t][k][2] = 0;
    }
    float sign[2] = {1, 1};
    int32_t mat_id[2] = {0, 0};
    ErMaterial m;
    memset(&m, 0, sizeof(m));
    m.albedo_tex = m.emission_tex = m.roughness_tex = m.metallic_tex = m.normal_tex = m.opacity_tex = m.transmission_tex = -1;
    m.albedo_shader_id = -1;
    m.albedo = {0.5f, 0.5f, 0.5f};
    m.opacity = 1; m.roughness = 1; m.specular = 0.5f; m.sheen_tint = 0.5f;
    float sky[3] = {0.5f, 0.5f, 0.5f};
    ErSceneDesc d;
    memset(&d, 0, sizeof(d));
    d.tri_count = 2;
    d.vertices = &verts[0][0][0]; d.normals = &normals[0][0][0]; d.tangents = &tangents[0][0][0];
    d.uvs = &uvs[0][0][0]; d.tangent_sign = sign; d.material_id = mat_id;
    d.material_count = 1; d.materials = &m;
    d.hdri.texture = {1, 1, 3, 0, sky};
    d.camera.focal_length = 0.035f; d.camera.sensor_width = 0.036f; d.camera.sensor_height = 0.024f;
    d.camera.aperture = 2.8f; d.camera.focus_distance = 1e6f; d.camera.position = {0.0f, 0.0f, -1.5f};
    d.x_res = W; d.y_res = H;

    printf("abi %d, devices %d\n", er_abi_version(), er_device_count());
    ErScene* s = nullptr;
    CHECK(er_scene_create(&d, &s));
    ErRenderParams p;
    memset(&p, 0, sizeof(p));
    p.sample_target = spp; p.block_size = 8; p.max_bounces = 5; p.world = 1;
    CHECK(er_render_begin(s, &p));
    CHECK(er_render_samples(s, spp));
    std::vector<float> img((size_t)W * H * 4);
    CHECK(er_read_pass(s, ER_PASS_BEAUTY, img.data()));
    uint32_t done = 0;
    CHECK(er_samples_done(s, &done));
    ErCounters c;
    CHECK(er_get_counters(s, &c));
    double sum = 0;
    for (size_t i = 0; i < img.size(); i += 4) sum += img[i] + img[i + 1] + img[i + 2];
    printf("samples_done %u checksum %.6f paths %llu bounce_samples %llu rays %llu\n", done, sum,
           (unsigned long long)c.paths, (unsigned long long)c.bounce_samples, (unsigned long long)c.rays);
    er_scene_destroy(s);
    return 0;
}
