// obj_dump -- loads an OBJ file through elevenrender_amd/host/eleven_obj.hpp and prints what a Scene would receive:
// one line per triangle corner (position, normal, uv, tangent), then the tangent sign and material name.
// Host-only (no GPU): used by tests/test_obj_cpu.py.
#include <cstdio>
#include <fstream>

#include "../../elevenrender_amd/host/eleven_obj.hpp"

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: obj_dump file.obj [recompute_normals] | obj_dump --mtl file.mtl\n"); return 2; }
    if (argc > 2 && std::string(argv[1]) == "--mtl") {     // material library through eleven::parse_mtl
        std::ifstream min(argv[2]);
        if (!min) { fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
        for (const eleven::UnloadedMaterial& u : eleven::parse_mtl(min)) {
            const eleven::Material& m = u.mat;
            printf("m %s Kd %.9g,%.9g,%.9g Ke %.9g,%.9g,%.9g specular %.9g eta %.9g opacity %.9g", m.name.c_str(), m.albedo.x, m.albedo.y, m.albedo.z,
                   m.emission.x, m.emission.y, m.emission.z, m.specular, m.eta, m.opacity);
            for (const auto& kv : u.maps) printf(" %s %s", kv.first.c_str(), kv.second.c_str());
            printf("\n");
        }
        return 0;
    }
    std::ifstream in(argv[1]);
    if (!in) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    std::vector<eleven::MeshObject> mos = eleven::load_obj(in, argc > 2 && argv[2][0] == '1');
    for (const eleven::MeshObject& mo : mos) {
        printf("object %s %zu\n", mo.name.c_str(), mo.tris.size());
        for (const eleven::Tri& t : mo.tris) {
            for (int j = 0; j < 3; j++)
                printf("c %.9g %.9g %.9g  %.9g %.9g %.9g  %.9g %.9g  %.9g %.9g %.9g\n", t.vertices[j].x, t.vertices[j].y, t.vertices[j].z, t.normals[j].x,
                       t.normals[j].y, t.normals[j].z, t.uv[j].x, t.uv[j].y, t.tangents[j].x, t.tangents[j].y, t.tangents[j].z);
            printf("t %.9g %s\n", t.tangentsSign, t.matName.empty() ? "-" : t.matName.c_str());
        }
    }
    return 0;
}
