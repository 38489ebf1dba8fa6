// host_cornell.cpp -- drives the C++ host mirror (elevenrender_amd/host/eleven_host.hpp) the way the reference's
// CommandManager drives its Scene/RenderingManager: build the Cornell scene of scenes.cornell(), render, print
// every beauty texel's bits folded into a checksum.  tests/test_gpu_host_cpp.py compares it with the Python path.
// With a third argument the geometry comes from that OBJ file through eleven::load_obj instead (the materials
// keep their names, as after the reference's load_object + pair_materials).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <fstream>

#include "../../elevenrender_amd/host/eleven_host.hpp"
#include "../../elevenrender_amd/host/eleven_obj.hpp"

using namespace eleven;

static void quad(MeshObject& m, Vector3 p0, Vector3 p1, Vector3 p2, Vector3 p3, const char* mat) {
    Vector3 q[2][3] = {{p0, p1, p2}, {p0, p2, p3}};
    for (auto& t3 : q) {
        Tri t;
        for (int k = 0; k < 3; k++) t.vertices[k] = t3[k];
        // face normal (cross(e1,e2) normalised) on all corners, tangent = normalised e1: what scenes.face_frame does
        float e1[3] = {t3[1].x - t3[0].x, t3[1].y - t3[0].y, t3[1].z - t3[0].z}, e2[3] = {t3[2].x - t3[0].x, t3[2].y - t3[0].y, t3[2].z - t3[0].z};
        float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        float nl = __builtin_sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), el = __builtin_sqrtf(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
        for (int k = 0; k < 3; k++) {
            t.normals[k] = Vector3(n[0] / nl, n[1] / nl, n[2] / nl);
            t.tangents[k] = Vector3(e1[0] / el, e1[1] / el, e1[2] / el);
        }
        t.uv[0] = Vector3(0, 0, 0); t.uv[1] = Vector3(1, 0, 0); t.uv[2] = Vector3(0, 1, 0);
        t.matName = mat;
        m.tris.push_back(t);
    }
}

int main(int argc, char** argv) {
    unsigned res = argc > 1 ? (unsigned)atoi(argv[1]) : 64, spp = argc > 2 ? (unsigned)atoi(argv[2]) : 4;
    Scene scene;
    Material red = Material::DefaultMaterial(), green = red, light = red;
    red.name = "red"; red.albedo = Vector3(0.8f, 0.1f, 0.1f);
    green.name = "green"; green.albedo = Vector3(0.1f, 0.8f, 0.1f);
    light.name = "light"; light.emission = Vector3(5, 5, 5);
    scene.addMaterial(red); scene.addMaterial(green); scene.addMaterial(light);
    MeshObject box;
    if (argc > 3) {
        std::ifstream in(argv[3]);
        if (!in) { fprintf(stderr, "error: cannot open %s\n", argv[3]); return 1; }
        for (MeshObject& mo : load_obj(in)) scene.addMeshObject(mo);
    } else {
    const float x0 = -1, x1 = 1, y0 = -1, y1 = 1, z0 = 2, z1 = 4, l = 0.4f, yl = 0.995f;
    quad(box, {x0, y0, z0}, {x0, y0, z1}, {x1, y0, z1}, {x1, y0, z0}, "default");
    quad(box, {x0, y1, z0}, {x1, y1, z0}, {x1, y1, z1}, {x0, y1, z1}, "default");
    quad(box, {x0, y0, z1}, {x0, y1, z1}, {x1, y1, z1}, {x1, y0, z1}, "default");
    quad(box, {x0, y0, z0}, {x0, y1, z0}, {x0, y1, z1}, {x0, y0, z1}, "red");
    quad(box, {x1, y0, z0}, {x1, y0, z1}, {x1, y1, z1}, {x1, y1, z0}, "green");
    quad(box, {-l, yl, 3 - l}, {l, yl, 3 - l}, {l, yl, 3 + l}, {-l, yl, 3 + l}, "light");
    scene.addMeshObject(box);
    }
    scene.pair_materials();
    scene.camera.position = Vector3(0, 0, -1.5f);
    scene.x_res = res; scene.y_res = res;
    try {
        RenderingManager rm;
        rm.pars.sampleTarget = spp;
        rm.start_rendering(&scene);
        rm.render(spp);
        std::vector<float> img = rm.get_pass("Beauty");
        unsigned long long h = 1469598103934665603ull;
        for (float f : img) { unsigned u; memcpy(&u, &f, 4); h = (h ^ u) * 1099511628211ull; }
        printf("samples %u fnv1a %016llx\n", rm.get_render_info().samples, h);
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
