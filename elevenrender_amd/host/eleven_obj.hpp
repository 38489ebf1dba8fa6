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
//   * tangents follow mikktspace.c's published algorithm step by step (generate_tangents below: exact welding, degenerate
//     triangles set aside and served by borrowing, groups grown across shared edges of consistent winding and equal
//     orientation, angle-weighted accumulation per sub-group in triangle order); that source is an un-vendored submodule of
//     the reference, so what is checked is hand-derived cases, not its output.
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
#include <utility>
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

// Per-corner tangents and the per-triangle sign of one MeshObject: the published algorithm of mikktspace.c
// (genTangSpaceDefault; Mikkelsen, "Simulation of Wrinkled Surfaces Revisited", 2008), restated for a triangle list with the
// callbacks of the reference (src/mikktspaceCallback.cpp:25-97: three vertices per face, positions / SMOOTH normals / uvs in,
// set_tspace_basic out).  Steps, in the order of the original (function names of mikktspace.c in brackets):
//   1. weld [GenerateSharedVerticesIndexList]: corners with EXACTLY equal position, normal and uv share one vertex index;
//   2. degenerate triangles [DegenPrologue]: a triangle with two equal POSITIONS takes no part in steps 3-6;
//   3. per triangle [InitTriInfo]: dP/ds and dP/dt from the uv gradients (eq. 18 / 19), normalised and flipped where the uv
//      mapping mirrors; ORIENT_PRESERVING = the signed uv area is > 0; a triangle whose uv area, |dP/ds| or |dP/dt| is not
//      above FLT_MIN is GROUP_WITH_ANY: it contributes nothing and takes the orientation of the first group that reaches it;
//   4. neighbours [BuildNeighborsFast]: two triangles are neighbours across an edge when they use the same two welded
//      indices in OPPOSITE order (consistent winding); every edge gets at most one neighbour, candidates in the order of the
//      sorted edge list (i0 < i1, then triangle number);
//   5. groups [Build4RuleGroups / AssignRecur]: per corner not yet in a group, a new group floods from triangle to triangle
//      across the two edges that meet at the corner's vertex, joining only triangles of the group's orientation -- so corners
//      that merely coincide in attributes but are not edge-connected around the vertex stay in different groups;
//   6. per group [GenerateTSpaces / EvalTspace]: for each triangle of the group, the sub-group of members whose projected
//      dP/ds and dP/dt are not exactly opposite to its own (threshold cos 180 degrees; GROUP_WITH_ANY members always); the
//      sub-group's tangent = sum over its good members, in ascending triangle order, of (the member's dP/ds projected into
//      the plane of the vertex normal and normalised) x (the angle at the vertex between the member's two edges, both
//      projected into that plane), normalised;
//   7. degenerate triangles [DegenEpilogue]: each corner borrows the tangent space of the first good triangle corner with
//      the same welded index, if there is one;
//   8. a corner nothing wrote keeps the initial space: tangent (1, 0, 0), NOT orientation preserving (sign -1);
//   9. out [genTangSpace]: per corner the tangent, sign = +1 / -1 for orientation preserving or not; the reference keeps ONE
//      sign per triangle, whatever its last corner delivered (src/mikktspaceCallback.cpp:124-132).
// All arithmetic in binary32 in the original's expression order (Normalize = v * (1 / length), acos in double rounded once).
// mikktspace.c itself is not in the reference tree (un-vendored submodule): parity UNPINNED, the tests check hand-derived cases.
inline void generate_tangents(MeshObject& mo) {
    using namespace obj_detail;
    const int n = (int)mo.tris.size();
    if (n == 0) return;
    const float FLTMIN = 1.17549435e-38f;
    auto not_zero = [&](float x) { return std::fabs(x) > FLTMIN; };
    auto vnot_zero = [&](Vector3 v) { return not_zero(v.x) || not_zero(v.y) || not_zero(v.z); };
    auto length = [](Vector3 v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); };
    auto normalize = [&](Vector3 v) { return mul(v, 1.0f / length(v)); };
    auto veq = [](Vector3 a, Vector3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; };
    // 1. weld: corner (f, i) -> vertex index; the vertex's attributes are those of its first corner
    struct Vert { Vector3 p, nrm, uv; };
    std::vector<Vert> verts;
    std::vector<int> idx(3 * (size_t)n);
    {
        auto key = [](const Vert& v) {
            auto z = [](float x) { return x == 0.0f ? 0.0f : x; };      // -0 == +0, as the original's float compare
            return std::make_tuple(z(v.p.x), z(v.p.y), z(v.p.z), z(v.nrm.x), z(v.nrm.y), z(v.nrm.z), z(v.uv.x), z(v.uv.y));
        };
        std::map<decltype(key(Vert())), int> seen;
        for (int f = 0; f < n; f++)
            for (int i = 0; i < 3; i++) {
                const Vert v{mo.tris[f].vertices[i], mo.tris[f].normals[i], Vector3(mo.tris[f].uv[i].x, mo.tris[f].uv[i].y, 0.0f)};
                auto it = seen.find(key(v));
                if (it == seen.end()) { it = seen.emplace(key(v), (int)verts.size()).first; verts.push_back(v); }
                idx[3 * f + i] = it->second;
            }
    }
    // 2. degenerate triangles: good triangles keep their order, the degenerate ones come after them
    std::vector<int> order, degen;
    for (int f = 0; f < n; f++) {
        const Vector3 p0 = verts[idx[3 * f]].p, p1 = verts[idx[3 * f + 1]].p, p2 = verts[idx[3 * f + 2]].p;
        (veq(p0, p1) || veq(p0, p2) || veq(p1, p2) ? degen : order).push_back(f);
    }
    const int ng = (int)order.size();          // "triangle t" below = order[t]
    auto vi = [&](int t, int i) { return idx[3 * order[t] + i]; };
    // 3. first-order derivatives
    struct Info { Vector3 os{0, 0, 0}, ot{0, 0, 0}; float mag_s = 0, mag_t = 0; bool orient = false, any = true; int neigh[3] = {-1, -1, -1}; int group[3] = {-1, -1, -1}; };
    std::vector<Info> info(ng);
    for (int t = 0; t < ng; t++) {
        const Vector3 v1 = verts[vi(t, 0)].p, v2 = verts[vi(t, 1)].p, v3 = verts[vi(t, 2)].p;
        const Vector3 t1 = verts[vi(t, 0)].uv, t2 = verts[vi(t, 1)].uv, t3 = verts[vi(t, 2)].uv;
        const float t21x = t2.x - t1.x, t21y = t2.y - t1.y, t31x = t3.x - t1.x, t31y = t3.y - t1.y;
        const Vector3 d1 = sub(v2, v1), d2 = sub(v3, v1);
        const float area2 = t21x * t31y - t21y * t31x;
        const Vector3 os = sub(mul(d1, t31y), mul(d2, t21y));            // eq. 18
        const Vector3 ot = add(mul(d1, -t31x), mul(d2, t21x));           // eq. 19
        Info& I = info[t];
        I.orient = area2 > 0;
        if (not_zero(area2)) {
            const float abs_area = std::fabs(area2), len_s = length(os), len_t = length(ot), fs = I.orient ? 1.0f : -1.0f;
            if (not_zero(len_s)) I.os = mul(os, fs / len_s);
            if (not_zero(len_t)) I.ot = mul(ot, fs / len_t);
            I.mag_s = len_s / abs_area;
            I.mag_t = len_t / abs_area;
            if (not_zero(I.mag_s) && not_zero(I.mag_t)) I.any = false;
        }
    }
    // 4. neighbours across edges: sorted (i0 < i1, triangle), first unassigned candidate with the opposite direction
    {
        struct Edge { int i0, i1, t; };
        std::vector<Edge> edges;
        for (int t = 0; t < ng; t++)
            for (int i = 0; i < 3; i++) {
                const int a = vi(t, i), b = vi(t, (i + 1) % 3);
                edges.push_back(Edge{std::min(a, b), std::max(a, b), t});
            }
        std::stable_sort(edges.begin(), edges.end(), [](const Edge& a, const Edge& b) { return std::tie(a.i0, a.i1, a.t) < std::tie(b.i0, b.i1, b.t); });
        // the edge of triangle t that joins welded indices (i0, i1): its number and its direction
        auto get_edge = [&](int t, int i0, int i1, int& a, int& b, int& e) {
            const int v0 = vi(t, 0), v1 = vi(t, 1), v2 = vi(t, 2);
            if (v0 == i0 || v0 == i1) {
                if (v1 == i0 || v1 == i1) { e = 0; a = v0; b = v1; }        // edge 0: vertex 0 -> 1
                else { e = 2; a = v2; b = v0; }                            // edge 2: vertex 2 -> 0
            } else { e = 1; a = v1; b = v2; }                              // edge 1: vertex 1 -> 2
        };
        for (size_t i = 0; i < edges.size(); i++) {
            int a0, a1, ea;
            get_edge(edges[i].t, edges[i].i0, edges[i].i1, a0, a1, ea);
            if (info[edges[i].t].neigh[ea] != -1) continue;
            for (size_t j = i + 1; j < edges.size() && edges[j].i0 == edges[i].i0 && edges[j].i1 == edges[i].i1; j++) {
                int b0, b1, eb;
                get_edge(edges[j].t, edges[j].i0, edges[j].i1, b0, b1, eb);
                if (a0 == b1 && a1 == b0 && info[edges[j].t].neigh[eb] == -1 && edges[j].t != edges[i].t) {
                    info[edges[i].t].neigh[ea] = edges[j].t;
                    info[edges[j].t].neigh[eb] = edges[i].t;
                    break;
                }
            }
        }
    }
    // 5. groups
    struct Group { int vert; bool orient; std::vector<int> faces; };
    std::vector<Group> groups;
    {
        struct Frame { int t; };
        auto assign = [&](int start, int g) {          // AssignRecur, with an explicit stack (depth-first: left neighbour first)
            std::vector<int> todo{start};
            while (!todo.empty()) {
                const int t = todo.back();
                todo.pop_back();
                Info& I = info[t];
                int i = -1;
                for (int k = 0; k < 3; k++) if (vi(t, k) == groups[g].vert) { i = k; break; }
                if (i < 0 || I.group[i] != -1) continue;                   // (already in this or in another group)
                if (I.any && I.group[0] == -1 && I.group[1] == -1 && I.group[2] == -1) I.orient = groups[g].orient;   // the first group decides
                if (I.orient != groups[g].orient) continue;
                groups[g].faces.push_back(t);
                I.group[i] = g;
                const int left = I.neigh[i], right = I.neigh[i > 0 ? i - 1 : 2];
                if (right >= 0) todo.push_back(right);
                if (left >= 0) todo.push_back(left);                       // (popped first)
            }
        };
        for (int t = 0; t < ng; t++)
            for (int i = 0; i < 3; i++) {
                if (info[t].any || info[t].group[i] != -1) continue;
                const int g = (int)groups.size();
                groups.push_back(Group{vi(t, i), info[t].orient, {t}});
                info[t].group[i] = g;
                const int left = info[t].neigh[i], right = info[t].neigh[i > 0 ? i - 1 : 2];
                if (left >= 0) assign(left, g);
                if (right >= 0) assign(right, g);
            }
    }
    // 6. tangent spaces per group and sub-group
    struct TSpace { Vector3 os{1.0f, 0.0f, 0.0f}; bool orient = false; bool written = false; };
    std::vector<TSpace> ts(3 * (size_t)n);                  // by ORIGINAL triangle and corner
    auto project = [&](Vector3 v, Vector3 nrm) {
        Vector3 r = sub(v, mul(nrm, dot(nrm, v)));
        return vnot_zero(r) ? normalize(r) : r;
    };
    auto eval = [&](const std::vector<int>& members, int vert) {
        Vector3 sum(0, 0, 0);
        for (int t : members) {
            if (info[t].any) continue;                      // only valid triangles get to add their contribution
            int i = -1;
            for (int k = 0; k < 3; k++) if (vi(t, k) == vert) { i = k; break; }
            if (i < 0) continue;
            const Vector3 nrm = verts[vert].nrm;
            const Vector3 os = project(info[t].os, nrm);
            const Vector3 p0 = verts[vi(t, i > 0 ? i - 1 : 2)].p, p1 = verts[vi(t, i)].p, p2 = verts[vi(t, i < 2 ? i + 1 : 0)].p;
            const Vector3 e1 = project(sub(p0, p1), nrm), e2 = project(sub(p2, p1), nrm);
            float c = dot(e1, e2);
            c = c > 1 ? 1 : (c < -1 ? -1 : c);
            const float angle = (float)std::acos((double)c);
            sum = add(sum, mul(os, angle));
        }
        return vnot_zero(sum) ? normalize(sum) : sum;
    };
    for (const Group& G : groups) {
        std::vector<std::pair<std::vector<int>, Vector3>> subs;      // unique sub-groups (sorted members) and their tangents
        for (int t : G.faces) {
            int index = -1;
            for (int k = 0; k < 3; k++) if (info[t].group[k] == (int)(&G - groups.data())) { index = k; break; }
            if (index < 0) continue;
            const Vector3 nrm = verts[G.vert].nrm;
            const Vector3 os = project(info[t].os, nrm), ot = project(info[t].ot, nrm);
            std::vector<int> members;
            for (int u : G.faces) {
                const Vector3 os2 = project(info[u].os, nrm), ot2 = project(info[u].ot, nrm);
                const bool any = info[t].any || info[u].any;
                if (any || u == t || (dot(os, os2) > -1.0f && dot(ot, ot2) > -1.0f)) members.push_back(u);
            }
            std::sort(members.begin(), members.end());
            size_t l = 0;
            while (l < subs.size() && subs[l].first != members) l++;
            if (l == subs.size()) subs.emplace_back(members, eval(members, G.vert));
            TSpace& out = ts[3 * (size_t)order[t] + index];
            out.os = subs[l].second;
            out.orient = G.orient;
            out.written = true;
        }
    }
    // 7. degenerate triangles borrow from the first good corner with the same welded vertex
    for (int f : degen)
        for (int i = 0; i < 3; i++) {
            bool found = false;
            for (int t = 0; t < ng && !found; t++)
                for (int k = 0; k < 3 && !found; k++)
                    if (vi(t, k) == idx[3 * f + i]) { ts[3 * (size_t)f + i] = ts[3 * (size_t)order[t] + k]; found = true; }
        }
    // 8. / 9. out
    for (int f = 0; f < n; f++) {
        Tri& t = mo.tris[f];
        for (int i = 0; i < 3; i++) {
            t.tangents[i] = ts[3 * (size_t)f + i].os;
            t.tangentsSign = ts[3 * (size_t)f + i].orient ? 1.0f : -1.0f;      // the last corner's call stays (set_tspace_basic)
        }
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
