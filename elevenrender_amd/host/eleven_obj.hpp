// eleven_obj.hpp -- OBJ + MTL text -> MeshObjects / Materials, dependency-free.
//
// What the reference does at load time (reference src/ObjLoader.cpp:69-147): parse with rapidobj, triangulate,
// flip z on positions and normals, normalise the normals, optionally recompute them face-weighted per position
// (ObjLoader.cpp:53-66; its `faces` map is shared by ALL shapes of the file, so a later shape also sees the faces
// earlier shapes put at a shared position -- mirrored here), take the material NAME per face (paired with the
// scene's materials later, src/Scene.cpp:104-120), one MeshObject per shape, then generate tangents with MikkTSpace
// (src/mikktspaceCallback.cpp:25-97: positions, SMOOTH normals and uvs in, per-corner tangent + one sign per
// triangle out through set_tspace_basic).  rapidobj and MikkTSpace are third-party, un-vendored and absent here, so
// this is our own reader with the same conventions; where the third-party code decides numbers parity is UNPINNED:
//   * polygons are triangulated as a fan around their first corner (rapidobj fans convex polygons the same way);
//   * tangents follow the published MikkTSpace construction (Mikkelsen 2008; mikktspace.c "genTangSpaceDefault"):
//     per triangle dP/du from the uv gradients and an orientation flag (sign of the uv area); triangle corners that
//     share position, normal AND uv, with the same orientation, form a group; within a group the triangles' dP/du,
//     projected into the plane of the shared normal and normalised, are accumulated weighted by the corner's angle
//     (measured between the edges projected into that plane) and normalised; sign = +1 where the uv mapping
//     preserves orientation, -1 where it mirrors.  What is NOT reproduced: MikkTSpace's welding of nearly equal
//     vertices, its edge-connectivity test when forming groups (here: exact equality of the three attributes) and its
//     borrowing of tangents for degenerate triangles (here: the default (1,0,0) / first edge).
// `parse_mtl` reads material libraries with the key handling of ObjLoader::parseMtl (src/ObjLoader.cpp:10-50).
// Indices may be negative (relative), `v/vt/vn`, `v//vn`, `v/vt` and `v` are accepted; `o` and `g` start a new shape.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
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

// Per-corner tangents and the per-triangle sign of one MeshObject, the MikkTSpace way (see the header of this file).
inline void generate_tangents(MeshObject& mo) {
    using namespace obj_detail;
    const size_t n = mo.tris.size();
    struct Face { Vector3 os; bool orient; bool valid; };
    std::vector<Face> faces(n);
    for (size_t i = 0; i < n; i++) {
        const Tri& t = mo.tris[i];
        const Vector3 d1 = sub(t.vertices[1], t.vertices[0]), d2 = sub(t.vertices[2], t.vertices[0]);
        const float s1 = t.uv[1].x - t.uv[0].x, t1 = t.uv[1].y - t.uv[0].y, s2 = t.uv[2].x - t.uv[0].x, t2 = t.uv[2].y - t.uv[0].y;
        const float area2 = s1 * t2 - s2 * t1;                          // twice the signed uv area
        Face f;
        f.orient = area2 > 0;
        f.os = sub(mul(d1, t2), mul(d2, t1));                           // dP/du scaled by the uv area (direction is what counts)
        if (!f.orient) f.os = mul(f.os, -1.0f);
        f.valid = std::fabs(area2) > 1e-30f && dot(f.os, f.os) > 0;
        faces[i] = f;
    }
    // groups: corners with the same (position, normal, uv, orientation)
    struct Key {
        Vector3 p, nrm, uv;
        bool orient;
        bool operator<(const Key& o) const {
            return std::tie(p.x, p.y, p.z, nrm.x, nrm.y, nrm.z, uv.x, uv.y, orient) < std::tie(o.p.x, o.p.y, o.p.z, o.nrm.x, o.nrm.y, o.nrm.z, o.uv.x, o.uv.y, o.orient);
        }
    };
    std::map<Key, Vector3> groups;
    auto corner_key = [&](size_t i, int j) { return Key{mo.tris[i].vertices[j], mo.tris[i].normals[j], mo.tris[i].uv[j], faces[i].orient}; };
    auto in_plane = [](Vector3 v, Vector3 nrm) { return sub(v, mul(nrm, dot(nrm, v))); };
    for (size_t i = 0; i < n; i++) {
        if (!faces[i].valid) continue;
        const Tri& t = mo.tris[i];
        for (int j = 0; j < 3; j++) {
            const Vector3 nrm = t.normals[j];
            const Vector3 os = normalized(in_plane(faces[i].os, nrm));
            // the corner's angle between its two edges, both projected into the plane of the normal
            const Vector3 e0 = normalized(in_plane(sub(t.vertices[(j + 1) % 3], t.vertices[j]), nrm));
            const Vector3 e1 = normalized(in_plane(sub(t.vertices[(j + 2) % 3], t.vertices[j]), nrm));
            const float c = std::min(1.0f, std::max(-1.0f, dot(e0, e1)));
            const float angle = std::acos(c);
            Vector3& g = groups[corner_key(i, j)];
            g = add(g, mul(os, angle));
        }
    }
    for (size_t i = 0; i < n; i++) {
        Tri& t = mo.tris[i];
        for (int j = 0; j < 3; j++) {
            Vector3 tg(1.0f, 0.0f, 0.0f);                                // MikkTSpace's default for a corner without a usable tangent
            if (faces[i].valid) {
                auto it = groups.find(corner_key(i, j));
                if (it != groups.end() && dot(it->second, it->second) > 0) tg = normalized(it->second);
            } else {
                const Vector3 e = in_plane(sub(t.vertices[1], t.vertices[0]), t.normals[j]);
                if (dot(e, e) > 0) tg = normalized(e);
            }
            t.tangents[j] = tg;
        }
        t.tangentsSign = (faces[i].orient || !faces[i].valid) ? 1.0f : -1.0f;    // set_tspace_basic: fSign, src/mikktspaceCallback.cpp:124-132
    }
}

// A material library as ObjLoader::parseMtl reads it (reference src/ObjLoader.cpp:10-50): per `newmtl` block
// Kd -> albedo, Ks -> specular (its first component), Ke -> emission, Ni -> eta, d -> opacity, and the map names
// map_Kd / map_Ns / map_Bump / refl kept by keyword.  Everything else is ignored, as there.
struct UnloadedMaterial {
    Material mat;
    std::map<std::string, std::string> maps;
};
inline std::vector<UnloadedMaterial> parse_mtl(std::istream& in) {
    std::vector<UnloadedMaterial> out;
    std::string line;
    auto second_word = [](const std::string& l) {                        // getSecondWord, :5-8
        std::string::size_type sp = l.find_first_of(" ");
        std::string w = sp == std::string::npos ? std::string() : l.substr(sp + 1);
        while (!w.empty() && (w.back() == '\r' || w.back() == ' ')) w.pop_back();
        return w;
    };
    auto vec3 = [](const std::string& rest) {
        Vector3 v;
        std::istringstream ls(rest);
        ls >> v.x >> v.y >> v.z;
        return v;
    };
    while (std::getline(in, line)) {
        size_t first = line.find_first_not_of(" \t");
        if (first == std::string::npos) continue;
        line = line.substr(first);
        if (line.compare(0, 6, "newmtl") == 0) {
            UnloadedMaterial u;
            u.mat.name = second_word(line);
            out.push_back(u);
            continue;
        }
        if (out.empty() || line.size() < 2) continue;
        UnloadedMaterial& u = out.back();
        if (line[0] == 'K' && line[1] == 'd') u.mat.albedo = vec3(line.substr(2));               // :23-25
        if (line[0] == 'K' && line[1] == 's') u.mat.specular = vec3(line.substr(2)).x;          // :27-29
        if (line[0] == 'K' && line[1] == 'e') u.mat.emission = vec3(line.substr(2));            // :31-33
        if (line[0] == 'N' && line[1] == 'i') u.mat.eta = std::strtof(second_word(line).c_str(), nullptr);       // :35-37
        if (line[0] == 'd' && (line[1] == ' ' || line[1] == '\t')) u.mat.opacity = std::strtof(second_word(line).c_str(), nullptr);   // :39-41
        for (const char* name : {"map_Kd", "map_Ns", "map_Bump", "refl"})                         // :42-48
            if (line.find(name) != std::string::npos) u.maps[name] = second_word(line);
    }
    return out;
}
inline std::vector<UnloadedMaterial> parse_mtl(const std::string& text) {
    std::istringstream in(text);
    return parse_mtl(in);
}

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
    // normals: face-weighted per position (ObjLoader.cpp:53-66), or the face normal where the file gave none.  The
    // reference's `faces` map lives across the shapes of one file (:77) and a shape's normals are recomputed right
    // after the shape is read (:136-137): shape k sees the faces of shapes 0..k.
    std::map<Vector3, Vector3, PosLess> acc;
    for (MeshObject& mo : out) {
        if (recompute_normals) {
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
        generate_tangents(mo);          // CalcTangents::calc(mo), ObjLoader.cpp:141-142
    }
    return out;
}

inline std::vector<MeshObject> load_obj(const std::string& text, bool recompute_normals = false) {
    std::istringstream in(text);
    return load_obj(in, recompute_normals);
}

}  // namespace eleven
