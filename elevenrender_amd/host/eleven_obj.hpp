// eleven_obj.hpp -- OBJ text -> MeshObjects, dependency-free.
//
// What the reference does at load time (reference src/ObjLoader.cpp:69-147): parse with rapidobj, triangulate,
// flip z on positions and normals, normalise the normals, optionally recompute them face-weighted per position
// (ObjLoader.cpp:53-66), take the material NAME per face (paired with the scene's materials later,
// src/Scene.cpp:104-120), one MeshObject per shape, then generate tangents with MikkTSpace
// (src/mikktspaceCallback.cpp:25-97).  rapidobj and MikkTSpace are third-party and absent here, so this is our own
// reader with the same conventions; where the third-party code decides numbers, parity is UNPINNED:
//   * polygons are triangulated as a fan around their first corner (rapidobj fans convex polygons the same way);
//   * tangents are per triangle, from the uv gradients: T = normalise(dP/du) orthogonalised against each corner
//     normal, sign = handedness of (N, T, dP/dv) -- the quantities MikkTSpace averages per vertex; a face without
//     usable uvs gets the normalised first edge.
// Indices may be negative (relative), `v/vt/vn`, `v//vn`, `v/vt` and `v` are accepted; `o` and `g` start a new shape.
#pragma once
#include <cmath>
#include <istream>
#include <map>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "eleven_host.hpp"

namespace eleven {

namespace obj_detail {
inline Vector3 sub(Vector3 a, Vector3 b) { return Vector3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vector3 add(Vector3 a, Vector3 b) { return Vector3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vector3 mul(Vector3 a, float s) { return Vector3(a.x * s, a.y * s, a.z * s); }
inline float dot(Vector3 a, Vector3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vector3 cross(Vector3 a, Vector3 b) {   // sign convention of the reference's Vector3::cross (src/Vector.h:173-175)
    return Vector3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
inline Vector3 normalized(Vector3 a) {
    float l = std::sqrt(dot(a, a));
    return l > 0 ? mul(a, 1.0f / l) : a;
}
struct PosLess {
    bool operator()(const Vector3& a, const Vector3& b) const { return std::tie(a.x, a.y, a.z) < std::tie(b.x, b.y, b.z); }
};
}  // namespace obj_detail

// Parses OBJ text.  recompute_normals: the reference's face-weighted recomputation (its default for `load_object`).
inline std::vector<MeshObject> load_obj(std::istream& in, bool recompute_normals = false) {
    using namespace obj_detail;
    std::vector<Vector3> P, N, T;
    std::vector<MeshObject> out;
    MeshObject cur;
    std::string mtl, line;
    auto flush = [&]() {
        if (!cur.tris.empty()) out.push_back(std::move(cur));
        cur = MeshObject();
    };
    auto resolve = [](long i, size_t n) -> long { return i > 0 ? i - 1 : (i < 0 ? (long)n + i : -1); };
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag) || tag[0] == '#') continue;
        if (tag == "v") { Vector3 v; ls >> v.x >> v.y >> v.z; P.push_back(v); }
        else if (tag == "vn") { Vector3 v; ls >> v.x >> v.y >> v.z; N.push_back(v); }
        else if (tag == "vt") { Vector3 v; ls >> v.x >> v.y; T.push_back(v); }
        else if (tag == "usemtl") { ls >> mtl; }
        else if (tag == "o" || tag == "g") { flush(); ls >> cur.name; }
        else if (tag == "f") {
            struct Corner { long p, t, n; };
            std::vector<Corner> cs;
            std::string tok;
            while (ls >> tok) {
                Corner c{0, 0, 0};
                size_t a = tok.find('/');
                c.p = std::stol(tok.substr(0, a));
                if (a != std::string::npos) {
                    size_t b = tok.find('/', a + 1);
                    std::string ts = tok.substr(a + 1, b == std::string::npos ? std::string::npos : b - a - 1);
                    if (!ts.empty()) c.t = std::stol(ts);
                    if (b != std::string::npos && b + 1 < tok.size()) c.n = std::stol(tok.substr(b + 1));
                }
                cs.push_back(c);
            }
            for (size_t k = 1; k + 1 < cs.size(); k++) {      // fan
                const Corner f[3] = {cs[0], cs[k], cs[k + 1]};
                Tri tri;
                for (int j = 0; j < 3; j++) {
                    long pi = resolve(f[j].p, P.size()), ti = resolve(f[j].t, T.size()), ni = resolve(f[j].n, N.size());
                    if (pi < 0 || pi >= (long)P.size()) throw std::runtime_error("load_obj: position index out of range");
                    const Vector3 p = P[pi];
                    tri.vertices[j] = Vector3(p.x, p.y, -p.z);                       // ObjLoader.cpp:116
                    if (ni >= 0 && ni < (long)N.size()) tri.normals[j] = normalized(Vector3(N[ni].x, N[ni].y, -N[ni].z));   // :117
                    if (ti >= 0 && ti < (long)T.size()) tri.uv[j] = Vector3(T[ti].x, T[ti].y, 0);                          // :118
                }
                tri.matName = mtl;
                cur.tris.push_back(tri);
            }
        }
    }
    flush();
    for (MeshObject& mo : out) {
        // normals: face-weighted per position (ObjLoader.cpp:53-66), or the face normal where the file gave none
        if (recompute_normals) {
            std::map<Vector3, Vector3, PosLess> acc;
            for (const Tri& t : mo.tris) {
                Vector3 fn = cross(sub(t.vertices[2], t.vertices[0]), sub(t.vertices[1], t.vertices[0]));   // cross(edge2, edge1), :61-63
                for (int j = 0; j < 3; j++) acc[t.vertices[j]] = add(acc[t.vertices[j]], fn);
            }
            for (Tri& t : mo.tris) for (int j = 0; j < 3; j++) t.normals[j] = normalized(acc[t.vertices[j]]);
        } else {
            for (Tri& t : mo.tris) {
                Vector3 fn = normalized(cross(sub(t.vertices[2], t.vertices[0]), sub(t.vertices[1], t.vertices[0])));
                for (int j = 0; j < 3; j++) if (dot(t.normals[j], t.normals[j]) == 0) t.normals[j] = fn;
            }
        }
        // tangents
        for (Tri& t : mo.tris) {
            const Vector3 e1 = sub(t.vertices[1], t.vertices[0]), e2 = sub(t.vertices[2], t.vertices[0]);
            const float du1 = t.uv[1].x - t.uv[0].x, dv1 = t.uv[1].y - t.uv[0].y, du2 = t.uv[2].x - t.uv[0].x, dv2 = t.uv[2].y - t.uv[0].y;
            const float det = du1 * dv2 - du2 * dv1;
            Vector3 dpdu = e1, dpdv = e2;
            if (std::fabs(det) > 1e-20f) {
                dpdu = mul(sub(mul(e1, dv2), mul(e2, dv1)), 1.0f / det);
                dpdv = mul(sub(mul(e2, du1), mul(e1, du2)), 1.0f / det);
            }
            const Vector3 fn = normalized(cross(e1, e2));
            for (int j = 0; j < 3; j++) {
                const Vector3 n = t.normals[j];
                Vector3 tg = sub(dpdu, mul(n, dot(n, dpdu)));
                if (dot(tg, tg) == 0) tg = e1;
                t.tangents[j] = normalized(tg);
            }
            t.tangentsSign = dot(cross(fn, dpdu), dpdv) < 0 ? -1.0f : 1.0f;
        }
    }
    return out;
}

inline std::vector<MeshObject> load_obj(const std::string& text, bool recompute_normals = false) {
    std::istringstream in(text);
    return load_obj(in, recompute_normals);
}

}  // namespace eleven
