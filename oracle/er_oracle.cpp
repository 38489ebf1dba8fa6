// er_oracle.cpp -- CPU restatement of ElevenRender's per-sample path (renderingKernel and
// everything it calls), written from reading the reference source.  TEST INFRASTRUCTURE ONLY;
// PARITY UNPINNED (see er_oracle.h).  Every function cites the reference lines it follows
// (paths relative to the reference tree).
//
// Arithmetic rules kept from the reference: IEEE binary32 throughout, no FMA contraction
// (build with -ffp-contract=off), expression association exactly as the reference source
// parses, RNG draws in left-to-right argument order (the reference is built with clang/DPC++).
#include "er_oracle.h"
#include "../elevenrender_amd/csrc/er_math.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- math provider
struct M {
    int mode;  // 0 libm, 1 er_math
    float sin(float x) const { return mode ? ermath::er_sin(x) : ::sinf(x); }
    float cos(float x) const { return mode ? ermath::er_cos(x) : ::cosf(x); }
    float acos(float x) const { return mode ? ermath::er_acos(x) : ::acosf(x); }
    float atan2(float y, float x) const { return mode ? ermath::er_atan2(y, x) : ::atan2f(y, x); }
    float pow(float x, float y) const { return mode ? ermath::er_pow(x, y) : ::powf(x, y); }
    float log(float x) const { return mode ? ermath::er_log(x) : ::logf(x); }
};

const float PIF = 3.14159265358979323846f;  // src/Math.hpp:6

// ---------------------------------------------------------------- Vector3 (src/Vector.h)
struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit V3(float a) : x(a), y(a), z(a) {}
    float operator[](int n) const { return n == 1 ? y : (n == 2 ? z : x); }  // Vector.h:114-119
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
inline V3 operator*(V3 a, V3 b) { return V3(b.x * a.x, b.y * a.y, b.z * a.z); }  // Vector.h:205-207
inline V3 operator/(V3 a, V3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline V3 addf(V3 a, float s) { return V3(a.x + s, a.y + s, a.z + s); }  // Vector.h:220-222
inline V3 neg(V3 a) { return a * -1.0f; }                                // Vector.h:109-111
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {  // Vector.h:173-175
    return V3((a.y * b.z - a.z * b.y), -(a.x * b.z - a.z * b.x), (a.x * b.y - a.y * b.x));
}
inline float length(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3 normalized(V3 a) {  // Vector.h:186-189
    if (length(a) == 0) return a;
    return a / length(a);
}
inline void normalize(V3& a) {  // Vector.h:177-184
    float l = length(a);
    if (l == 0) return;
    a.x /= l; a.y /= l; a.z /= l;
}
inline V3 reflectv(V3 v1, V3 v2) { return v1 - (2 * dot(v1, v2)) * v2; }  // Vector.h:238-240

// src/Math.hpp
inline float mapf(float a, float b, float c, float d, float e) { return d + ((a - b) / (c - b)) * (e - d); }
inline float clampf(float a, float b, float c) { return a < b ? b : a > c ? c : a; }
inline float lerpf(float a, float b, float c) { return a + c * (b - a); }  // FAST_LERP
inline V3 lerpv(V3 a, V3 b, float c) { return V3(lerpf(a.x, b.x, c), lerpf(a.y, b.y, c), lerpf(a.z, b.z, c)); }
inline float minf(float a, float b) { return a < b ? a : b; }
inline float maxf(float a, float b) { return a > b ? a : b; }
inline void limitUV(float& u, float& v) {  // Math.hpp:48-51
    u += (float)(-(int)(u > 1) + -(int)(u < 0));
    v += (float)(-(int)(v > 1) + -(int)(v < 0));
}

struct Ray {  // src/Ray.h
    V3 origin, direction;
    Ray() : origin(), direction(0, 0, 1) {}
    Ray(V3 o, V3 d) : origin(o), direction(d) { normalize(direction); }
};

struct Hit {  // src/Hit.h
    V3 position, normal, tangent, bitangent, gnormal;
    bool valid = false;
    int material_id = 0;
    float tu = 0, tv = 0;
    int triIdx = -2;
    int tri = -1;  // original triangle id (ours, for traces)
};

struct Tri {  // src/Tri.h:8-21 (uv keeps only the x,y the kernel reads)
    V3 v[3], n[3], t[3];
    float uvx[3], uvy[3];
    float sign;
    int material;
};

// ---------------------------------------------------------------- RNG (src/kernel.cpp:25-47)
inline uint32_t jenkins_u32(uint32_t seed) {
    uint32_t hash = 0;
    for (int i = 0; i < 4; i++) {
        hash += (seed >> (i * 8)) & 0xFF;
        hash += (hash << 10);
        hash ^= (hash >> 6);
    }
    hash += (hash << 3);
    hash ^= (hash >> 11);
    hash += (hash << 15);
    return hash;
}
struct Rng {
    uint32_t state;
    float next() {
        state ^= state << 13;
        state ^= state >> 17;
        state ^= state << 5;
        return (float)state / 4294967296.0f;  // static_cast<float>(UINT_MAX) == 2^32
    }
};

// ---------------------------------------------------------------- Tri::hit (src/Tri.h:41-144)
inline V3 projectOnPlane(V3 position, V3 origin, V3 normal) {
    return position - dot(position - origin, normal) * normal;
}
inline bool tri_hit(const Tri& T, const Ray& ray, Hit& hit) {
    const float EPSILON = 0.0000001f;
    V3 edge1 = T.v[1] - T.v[0];
    V3 edge2 = T.v[2] - T.v[0];
    V3 pvec = cross(ray.direction, edge2);
    float det = dot(edge1, pvec);
    float inv_det = 1.0f / det;
    if (det > -EPSILON && det < EPSILON) return false;
    V3 tvec = ray.origin - T.v[0];
    float u = dot(tvec, pvec) * inv_det;
    if (u < 0 || u > 1) return false;
    V3 qvec = cross(tvec, edge1);
    float v = dot(ray.direction, qvec) * inv_det;
    if (v < 0 || (u + v) > 1) return false;
    float t = dot(edge2, qvec) * inv_det;
    if (t < 0) return false;

    float tu = T.uvx[0] + (T.uvx[1] - T.uvx[0]) * u + (T.uvx[2] - T.uvx[0]) * v;
    float tv = T.uvy[0] + (T.uvy[1] - T.uvy[0]) * u + (T.uvy[2] - T.uvy[0]) * v;
    V3 geomPosition = ray.origin + ray.direction * t;

    V3 n0 = T.n[0], n1 = T.n[1], n2 = T.n[2];
    V3 shadingNormal = normalized(n0 + (n1 - n0) * u + (n2 - n0) * v);
    V3 compNormal = normalized(cross(edge1, edge2));
    if (dot(compNormal, ray.direction) > 0) compNormal = compNormal * -1.0f;
    V3 shadingTangent = T.t[0] + (T.t[1] - T.t[0]) * u + (T.t[2] - T.t[0]) * v;

    V3 p0 = projectOnPlane(geomPosition, T.v[0], n0);
    V3 p1 = projectOnPlane(geomPosition, T.v[1], n1);
    V3 p2 = projectOnPlane(geomPosition, T.v[2], n2);
    V3 shadingPosition = p0 + (p1 - p0) * u + (p2 - p0) * v;
    bool convex = dot(shadingPosition - geomPosition, shadingNormal) > 0;

    hit.tangent = shadingTangent;
    hit.position = convex ? shadingPosition : geomPosition;
    hit.normal = shadingNormal;
    hit.gnormal = compNormal;
    hit.bitangent = T.sign * cross(hit.normal, hit.tangent);
    hit.valid = true;
    hit.tu = tu;
    hit.tv = tv;
    hit.material_id = T.material;
    return true;
}

// ---------------------------------------------------------------- BVH (src/BVH.h, src/BVH.cpp)
const int BVH_DEPTH = 18;    // src/Definitions.h:13
const int BVH_SAHBINS = 14;  // src/Definitions.h:14

struct Node {  // src/BVH.h:24-33
    V3 b1, b2;
    int from = 0, to = 0, idx = 0, depth = 0;
    bool valid = false;
};

inline float boundsArea(V3 b1, V3 b2) {  // BVH.cpp:469-477
    float x = b2.x - b1.x, y = b2.y - b1.y, z = b2.z - b1.z;
    return 2 * (x * y + x * z + y * z);
}
inline void boundsUnion(V3 b1, V3 b2, V3 b3, V3 b4, V3& b5, V3& b6) {  // BVH.cpp:442-467
    if (boundsArea(b1, b2) <= 0 || boundsArea(b3, b4) <= 0) {
        if (boundsArea(b1, b2) <= 0) { b5 = b3; b6 = b4; }
        if (boundsArea(b3, b4) <= 0) { b5 = b1; b6 = b2; }
    } else {
        b5.x = minf(b1.x, minf(b2.x, minf(b3.x, b4.x)));
        b5.y = minf(b1.y, minf(b2.y, minf(b3.y, b4.y)));
        b5.z = minf(b1.z, minf(b2.z, minf(b3.z, b4.z)));
        b6.x = maxf(b1.x, maxf(b2.x, maxf(b3.x, b4.x)));
        b6.y = maxf(b1.y, maxf(b2.y, maxf(b3.y, b4.y)));
        b6.z = maxf(b1.z, maxf(b2.z, maxf(b3.z, b4.z)));
    }
}
inline void triBounds(const Tri& t, V3& b1, V3& b2) {  // BVH.cpp:479-488
    b1.x = minf(t.v[0].x, minf(t.v[1].x, t.v[2].x));
    b1.y = minf(t.v[0].y, minf(t.v[1].y, t.v[2].y));
    b1.z = minf(t.v[0].z, minf(t.v[1].z, t.v[2].z));
    b2.x = maxf(t.v[0].x, maxf(t.v[1].x, t.v[2].x));
    b2.y = maxf(t.v[0].y, maxf(t.v[1].y, t.v[2].y));
    b2.z = maxf(t.v[0].z, maxf(t.v[1].z, t.v[2].z));
}
inline V3 centroid(const Tri& t) {  // Tri.h:30-35
    return V3(t.v[0].x + t.v[1].x + t.v[2].x, t.v[0].y + t.v[1].y + t.v[2].y, t.v[0].z + t.v[1].z + t.v[2].z) / 3.0f;
}

struct RefBVH {
    const std::vector<Tri>* tris = nullptr;
    std::vector<Node> nodes;      // 2 << BVH_DEPTH entries, pre-order (BVH.h:47)
    std::vector<int> triIndices;  // leaf order -> original tri id
    int nodeIdx = 0, triIdx = 0;

    // BVH.cpp:490-534: bounds of a list; an EMPTY list leaves b1,b2 as they were (zero).
    void listBounds(const std::vector<int>& l, V3& b1, V3& b2) const {
        if (l.empty()) return;
        const std::vector<Tri>& T = *tris;
        b1 = T[l[0]].v[0];
        b2 = T[l[0]].v[0];
        for (int id : l) {
            for (int k = 0; k < 3; k++) {
                b1.x = minf(T[id].v[k].x, b1.x); b1.y = minf(T[id].v[k].y, b1.y); b1.z = minf(T[id].v[k].z, b1.z);
            }
            for (int k = 0; k < 3; k++) {
                b2.x = maxf(T[id].v[k].x, b2.x); b2.y = maxf(T[id].v[k].y, b2.y); b2.z = maxf(T[id].v[k].z, b2.z);
            }
        }
    }

    // BVH.cpp:327-415
    void divideSAH(const std::vector<int>& l, std::vector<int>& left, std::vector<int>& right) const {
        if (l.empty()) return;
        const std::vector<Tri>& T = *tris;
        V3 totalB1, totalB2;
        int bestBin = 0, bestAxis = 0;
        float bestHeuristic = 3.402823466e+38f;
        listBounds(l, totalB1, totalB2);
        // centroids and per-tri bounds are pure functions of the tri: compute once per call
        for (int axis = 0; axis < 3; axis++) {
            V3 b1s[BVH_SAHBINS], b2s[BVH_SAHBINS];
            int count[BVH_SAHBINS];
            for (int i = 0; i < BVH_SAHBINS; i++) count[i] = 0;
            for (int id : l) {
                int bin = 0;
                V3 b1, b2;
                if (totalB1[axis] != totalB2[axis]) {
                    float c = centroid(T[id])[axis];
                    bin = ermath::f2i(mapf(c, totalB1[axis], totalB2[axis], 0, BVH_SAHBINS - 1));
                }
                count[bin]++;
                triBounds(T[id], b1, b2);
                boundsUnion(b1s[bin], b2s[bin], b1, b2, b1s[bin], b2s[bin]);
            }
            for (int i = 0; i < BVH_SAHBINS; i++) {
                int count1 = 0, count2 = 0;
                V3 b1, b2, b3, b4;
                for (int j = 0; j < i; j++) { count1 += count[j]; boundsUnion(b1, b2, b1s[j], b2s[j], b1, b2); }
                for (int k = i; k < BVH_SAHBINS; k++) { count2 += count[k]; boundsUnion(b3, b4, b1s[k], b2s[k], b3, b4); }
                float heuristic = boundsArea(b1, b2) * (float)count1 + boundsArea(b3, b4) * (float)count2;
                if (heuristic < bestHeuristic) { bestHeuristic = heuristic; bestBin = i; bestAxis = axis; }
            }
        }
        for (int id : l) {
            float c = centroid(T[id])[bestAxis];
            int bin = ermath::f2i(mapf(c, totalB1[bestAxis], totalB2[bestAxis], 0, BVH_SAHBINS - 1));
            if (bin < bestBin) left.push_back(id); else right.push_back(id);
        }
    }

    // BVH.cpp:239-281
    void buildAux(int depth, const std::vector<int>& l) {
        V3 b1, b2;
        listBounds(l, b1, b2);
        Node n;
        n.idx = nodeIdx; n.b1 = b1; n.b2 = b2; n.depth = depth; n.valid = true;
        if (depth == BVH_DEPTH) {
            n.from = triIdx; n.to = triIdx + (int)l.size();
            nodes[nodeIdx++] = n;
            for (int id : l) triIndices[triIdx++] = id;
        } else {
            n.from = 0; n.to = 0;
            nodes[nodeIdx++] = n;
            std::vector<int> left, right;
            divideSAH(l, left, right);
            buildAux(depth + 1, left);
            buildAux(depth + 1, right);
        }
    }

    void build(const std::vector<Tri>& T) {  // BVH.cpp:132-156, Scene.cpp:122-143
        tris = &T;
        nodes.assign((size_t)2 << BVH_DEPTH, Node());
        triIndices.assign(T.size(), 0);
        std::vector<int> all(T.size());
        for (size_t i = 0; i < T.size(); i++) all[i] = (int)i;
        buildAux(0, all);
    }

    // BVH.cpp:27-61
    static bool intersect(const Ray& ray, V3 b1, V3 b2) {
        V3 dirfrac;
        dirfrac.x = 1.0f / ray.direction.x;
        dirfrac.y = 1.0f / ray.direction.y;
        dirfrac.z = 1.0f / ray.direction.z;
        float t1 = (b1.x - ray.origin.x) * dirfrac.x;
        float t2 = (b2.x - ray.origin.x) * dirfrac.x;
        float t3 = (b1.y - ray.origin.y) * dirfrac.y;
        float t4 = (b2.y - ray.origin.y) * dirfrac.y;
        float t5 = (b1.z - ray.origin.z) * dirfrac.z;
        float t6 = (b2.z - ray.origin.z) * dirfrac.z;
        float tmin = maxf(maxf(minf(t1, t2), minf(t3, t4)), minf(t5, t6));
        float tmax = minf(minf(maxf(t1, t2), maxf(t3, t4)), maxf(t5, t6));
        if (tmax < 0) return false;
        if (tmin > tmax) return false;
        return true;
    }
};

// ---------------------------------------------------------------- Texture (src/Texture.cpp:172-292)
struct Tex {
    int width = 1, height = 1, channels = 1, filter = 0;
    std::vector<float> data;
};
inline V3 texCoords(const Tex& t, int x, int y, uint64_t* fetches) {  // Texture.cpp:172-200
    V3 pixel;
    x %= t.width;
    y %= t.height;
    if (x < 0) x *= -1;
    if (y < 0) y *= -1;
    if (fetches) (*fetches)++;
    const float* d = t.data.data();
    if (t.channels == 0) {
        pixel = V3();
    } else if (t.channels == 1) {
        pixel = V3(d[y * t.width + x]);
    } else if (t.channels == 2) {
        pixel.x = d[t.channels * (y * t.width + x) + 0];
        pixel.y = d[t.channels * (y * t.width + x) + 1];
    } else {
        pixel.x = d[t.channels * (y * t.width + x) + 0];
        pixel.y = d[t.channels * (y * t.width + x) + 1];
        pixel.z = d[t.channels * (y * t.width + x) + 2];
    }
    return pixel;
}
inline V3 texUV(const Tex& t, float u, float v, uint64_t* f) {  // Texture.cpp:202-204
    return texCoords(t, ermath::f2i(u * t.width), ermath::f2i(v * t.height), f);
}
inline V3 texBilinear(const Tex& t, float u, float v, uint64_t* f) {  // Texture.cpp:206-227
    float x = u * t.width, y = v * t.height;
    float t1x = std::floor(x), t1y = std::floor(y);
    float t2x = t1x + 1, t2y = t1y + 1;
    float a = (x - t1x) / (t2x - t1x);
    float b = (y - t1y) / (t2y - t1y);
    V3 v1 = texCoords(t, ermath::f2i(t1x), ermath::f2i(t1y), f);
    V3 v2 = texCoords(t, ermath::f2i(t2x), ermath::f2i(t1y), f);
    V3 v3 = texCoords(t, ermath::f2i(t1x), ermath::f2i(t2y), f);
    V3 v4 = texCoords(t, ermath::f2i(t2x), ermath::f2i(t2y), f);
    return lerpv(lerpv(v1, v2, a), lerpv(v3, v4, a), b);
}
inline V3 texFiltered(const Tex& t, float u, float v, uint64_t* f) {  // Texture.cpp:229-236
    return t.filter == 1 ? texBilinear(t, u, v, f) : texUV(t, u, v, f);
}
inline void sphericalMapping(const M& m, V3 origin, V3 point, float radius, float& u, float& v) {  // Texture.cpp:239-251
    V3 p = (point - origin) / radius;
    float theta = m.acos(-p.y);
    float phi = m.atan2(-p.z, p.x) + PIF;
    u = phi / (2 * PIF);
    v = theta / PIF;
    limitUV(u, v);
}
inline V3 inverseTransformUV(const Tex& t, float u, float v) {  // Texture.cpp:267-278
    int x = ermath::f2i(u * t.width);
    int y = ermath::f2i(v * t.height);
    float nu = (float)x / (float)t.width;
    float nv = (float)y / (float)t.height;
    limitUV(nu, nv);
    return V3(nu, nv, 0);
}
inline V3 reverseSphericalMapping(const M& m, float u, float v) {  // Texture.cpp:280-292
    float phi = u * 2 * PIF;
    float theta = v * PIF;
    float px = m.cos(phi - PIF);
    float py = -m.cos(theta);
    float pz = -m.sin(phi - PIF);
    float a = std::sqrt(1 - py * py);
    return V3(a * px, py, a * pz);
}

// ---------------------------------------------------------------- HDRI (src/HDRI.cpp)
struct Hdri {
    Tex texture;
    std::vector<float> cdf;
    float radianceSum = 0;
};
inline void generateCDF(const Tex& t, std::vector<float>& cdf, float& radianceSum) {  // HDRI.cpp:62-83
    int c = 0;
    radianceSum = 0;
    cdf.assign((size_t)t.width * t.height + 1, 0.0f);
    cdf[0] = 0;
    for (int j = 0; j < t.height; j++)
        for (int i = 0; i < t.width; i++) {
            V3 d = texCoords(t, i, j, nullptr);
            radianceSum += d.x + d.y + d.z;
        }
    for (int j = 0; j < t.height; j++)
        for (int i = 0; i < t.width; i++) {
            V3 d = texCoords(t, i, j, nullptr);
            cdf[c + 1] = cdf[c] + (d.x + d.y + d.z) / radianceSum;
            c++;
        }
}
inline int binarySearch(const float* arr, float value, int length) {  // HDRI.cpp:85-98
    int from = 0;
    int to = length - 1;
    while (to - from > 0) {
        int m = from + (to - from) / 2;
        if (m >= length || m < 0) return 0;
        if (value == arr[m]) return m;
        if (value < arr[m]) to = m - 1;
        if (value > arr[m]) from = m + 1;
    }
    return to;
}
inline float hdriPdf(const M& m, const Hdri& h, int x, int y, uint64_t* f) {  // HDRI.cpp:101-107
    V3 dv = texCoords(h.texture, x, y, f);
    float theta = (((float)y / (float)h.texture.height)) * PIF;
    return ((dv.x + dv.y + dv.z) / h.radianceSum) * h.texture.width * h.texture.height / (2.0f * PIF * m.sin(theta));
}
inline V3 hdriSample(const Hdri& h, float r1) {  // HDRI.cpp:109-117
    int count = binarySearch(h.cdf.data(), r1, h.texture.width * h.texture.height);
    int x = count % h.texture.width;
    int y = count / h.texture.width;
    return V3((float)x, (float)y, 0);
}

// ---------------------------------------------------------------- HitData + Disney (src/kernel.h:92-121, src/Disney.cpp)
struct HitData {
    float metallic, roughness, clearcoatGloss, clearcoat, anisotropic, eta, transmission, specular,
        specularTint, sheenTint, subsurface, sheen, opacity, ax, ay;
    V3 emission, albedo;
    int triIdx;
    V3 position, normal, gnormal, tangent, bitangent;
};

inline float SchlickFresnel(float u) {  // Disney.cpp:34-38
    float m = clampf(1.0f - u, 0.0f, 1.0f);
    float m2 = m * m;
    return m2 * m2 * m;
}
inline float GTR1(const M& mm, float NDotH, float a) {  // Disney.cpp:55-62
    if (a >= 1.0f) return (1.0f / PIF);
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    return (a2 - 1.0f) / (PIF * mm.log(a2) * t);
}
inline float GTR2_aniso(float NDotH, float HDotX, float HDotY, float ax, float ay) {  // Disney.cpp:70-75
    float a = HDotX / ax;
    float b = HDotY / ay;
    float c = a * a + b * b + NDotH * NDotH;
    return 1.0f / (PIF * ax * ay * c * c);
}
inline float SmithG_GGX(float NDotV, float alphaG) {  // Disney.cpp:77-81
    float a = alphaG * alphaG;
    float b = NDotV * NDotV;
    return 1.0f / (NDotV + std::sqrt(a + b - a * b));
}
inline float SmithG_GGX_aniso(float NDotV, float VDotX, float VDotY, float ax, float ay) {  // Disney.cpp:83-88
    float a = VDotX * ax;
    float b = VDotY * ay;
    float c = NDotV;
    return 1.0f / (NDotV + std::sqrt(a * a + b * b + c * c));
}

float DisneyPdf(const M& mm, const HitData& hd, V3 V, V3 N, V3 L) {  // Disney.cpp:97-133
    V3 H = normalized(L + V);
    V3 T = hd.tangent, B = hd.bitangent;
    float NDotH = std::fabs(dot(N, H));
    if (dot(N, L) <= 0.0f) return 1.0f;
    float clearcoatAlpha = lerpf(0.1f, 0.001f, hd.clearcoatGloss);
    float diffuseRatio = 0.5f * (1.0f - hd.metallic);
    float specularRatio = 1.0f - diffuseRatio;
    float aspect = std::sqrt(1.0f - hd.anisotropic * 0.9f);
    float ax = maxf(0.001f, hd.roughness / aspect);
    float ay = maxf(0.001f, hd.roughness * aspect);
    float pdfGTR2_aniso = GTR2_aniso(NDotH, dot(H, T), dot(H, B), ax, ay) * NDotH;
    float pdfGTR1 = GTR1(mm, NDotH, clearcoatAlpha) * NDotH;
    float ratio = 1.0f / (1.0f + hd.clearcoat);
    float pdfSpec = lerpf(pdfGTR1, pdfGTR2_aniso, ratio) / (4.0f * std::fabs(dot(L, H)));
    float pdfDiff = std::fabs(dot(L, N)) * (1.0f / PIF);
    return diffuseRatio * pdfDiff + specularRatio * pdfSpec;
}

inline V3 CosineSampleHemisphere(const M& mm, float u1, float u2) {  // Sampling.h:30-40
    V3 dir;
    float r = std::sqrt(u1);
    float phi = 2.0f * PIF * u2;
    dir.x = r * mm.cos(phi);
    dir.y = r * mm.sin(phi);
    dir.z = std::sqrt(maxf(0.0f, 1.0f - dir.x * dir.x - dir.y * dir.y));
    return dir;
}
inline V3 ImportanceSampleGGX(const M& mm, float rgh, float r1, float r2) {  // Sampling.h:42-53
    float a = maxf(0.001f, rgh);
    float phi = r1 * PIF * 2;
    float cosTheta = std::sqrt((1.0f - r2) / (1.0f + (a * a - 1.0f) * r2));
    float sinTheta = clampf(std::sqrt(1.0f - (cosTheta * cosTheta)), 0.0f, 1.0f);
    float sinPhi = mm.sin(phi);
    float cosPhi = mm.cos(phi);
    return V3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
}
V3 DisneySample(const M& mm, const HitData& hd, V3 V, V3 N, float r1, float r2, float r3) {  // Disney.cpp:136-158
    V3 T = hd.tangent, B = hd.bitangent;
    V3 dir;
    float diffuseRatio = 0.5f * (1.0f - hd.metallic);
    if (r3 < diffuseRatio) {
        V3 H = CosineSampleHemisphere(mm, r1, r2);
        H = T * H.x + B * H.y + N * H.z;
        dir = H;
    } else {
        V3 H = ImportanceSampleGGX(mm, hd.roughness, r1, r2);
        H = T * H.x + B * H.y + N * H.z;
        dir = reflectv(-1 * V, H);
    }
    return dir;
}
V3 DisneyEval(const M& mm, const HitData& hd, V3 V, V3 N, V3 L) {  // Disney.cpp:160-230
    V3 T = hd.tangent, B = hd.bitangent;
    V3 H = normalized(L + V);
    float NDotL = std::fabs(dot(N, L));
    float NDotV = std::fabs(dot(N, V));
    float NDotH = std::fabs(dot(N, H));
    float LDotH = std::fabs(dot(L, H));
    V3 brdf(0.0f);
    if (hd.transmission < 1.0f && dot(N, L) > 0.0f && dot(N, V) > 0.0f) {
        V3 Cdlin = hd.albedo;
        float Cdlum = 0.3f * Cdlin.x + 0.6f * Cdlin.y + 0.1f * Cdlin.z;
        V3 Ctint = Cdlum > 0.0f ? Cdlin / Cdlum : V3(1.0f);
        V3 Cspec0 = lerpv(hd.specular * 0.08f * lerpv(V3(1.0f), Ctint, hd.specularTint), Cdlin, hd.metallic);
        V3 Csheen = lerpv(V3(1.0f), Ctint, hd.sheenTint);
        float FL = SchlickFresnel(NDotL);
        float FV = SchlickFresnel(NDotV);
        float Fd90 = 0.5f + 2.0f * LDotH * LDotH * hd.roughness;
        float Fd = lerpf(1.0f, Fd90, FL) * lerpf(1.0f, Fd90, FV);
        float Fss90 = LDotH * LDotH * hd.roughness;
        float Fss = lerpf(1.0f, Fss90, FL) * lerpf(1.0f, Fss90, FV);
        float ss = 1.25f * (Fss * (1.0f / (NDotL + NDotV) - 0.5f) + 0.5f);
        float aspect = std::sqrt(1.0f - hd.anisotropic * 0.9f);
        float ax = maxf(0.001f, hd.roughness / aspect);
        float ay = maxf(0.001f, hd.roughness * aspect);
        float Ds = GTR2_aniso(NDotH, dot(H, T), dot(H, B), ax, ay);
        float FH = SchlickFresnel(LDotH);
        V3 Fs = lerpv(Cspec0, V3(1.0f), FH);
        float Gs = SmithG_GGX_aniso(NDotL, dot(L, T), dot(L, B), ax, ay);
        Gs *= SmithG_GGX_aniso(NDotV, dot(V, T), dot(V, B), ax, ay);
        V3 Fsheen = FH * hd.sheen * Csheen;
        float Dr = GTR1(mm, NDotH, lerpf(0.1f, 0.001f, hd.clearcoatGloss));
        float Fr = lerpf(0.04f, 1.0f, FH);
        float Gr = SmithG_GGX(NDotL, 0.25f) * SmithG_GGX(NDotV, 0.25f);
        brdf = addf(((1.0f / PIF) * lerpf(Fd, ss, hd.subsurface) * Cdlin + Fsheen) * (1.0f - hd.metallic) + Gs * Fs * Ds,
                    0.25f * hd.clearcoat * Gr * Fr * Dr);
    }
    return brdf;
}

}  // namespace

// ---------------------------------------------------------------- the scene + kernel
struct Oracle {
    M m;
    int max_bounces = 5;
    int traversal = 0;
    int threads = 1;
    ErCamera cam;
    uint32_t x_res = 0, y_res = 0;
    std::vector<Tri> tris;
    std::vector<ErPointLight> lights;
    uint32_t flags = 0;
    std::vector<ErMaterial> materials;
    std::vector<Tex> textures;
    Hdri hdri;
    RefBVH bvh;
    double build_s = 0;
    std::vector<float> passes;
    std::vector<uint32_t> samples;
    std::vector<uint32_t> rng;
    OracleCounters ctr;

    // BVH.cpp:105-120 ; nearestHit.triIdx = SLOT in triIndices
    void intersectRange(const Ray& ray, int from, int to, Hit& nearestHit, OracleCounters& c) const {
        for (int i = from; i < to; i++) {
            Hit hit;
            int id = traversal == 0 ? bvh.triIndices[i] : i;
            c.tri_tests++;
            if (tri_hit(tris[id], ray, hit)) {
                c.tri_hits++;
                if (!nearestHit.valid || length(hit.position - ray.origin) < length(nearestHit.position - ray.origin)) {
                    nearestHit = hit;
                    nearestHit.triIdx = i;
                    nearestHit.tri = id;
                }
            }
        }
    }
    // BVH.cpp:63-103 (stack of node copies restated as a stack of node indices; -1 = the invalid sentinel)
    void transverse(const Ray& ray, Hit& nearestHit, OracleCounters& c) const {
        int stack[64];
        int sp = 0;
        stack[sp++] = -1;
        int node = 0;
        const Node* N = bvh.nodes.data();
        do {
            const Node& cur = N[node];
            c.node_visits++;
            const Node& lChild = N[cur.idx + 1];
            const Node& rChild = N[cur.idx + (2 << (BVH_DEPTH - cur.depth - 1))];
            bool lOverlap = RefBVH::intersect(ray, lChild.b1, lChild.b2);
            bool rOverlap = RefBVH::intersect(ray, rChild.b1, rChild.b2);
            if (cur.depth == (BVH_DEPTH - 1) && rOverlap) intersectRange(ray, rChild.from, rChild.to, nearestHit, c);
            if (cur.depth == (BVH_DEPTH - 1) && lOverlap) intersectRange(ray, lChild.from, lChild.to, nearestHit, c);
            bool traverseL = (lOverlap && cur.depth != (BVH_DEPTH - 1));
            bool traverseR = (rOverlap && cur.depth != (BVH_DEPTH - 1));
            if (!traverseL && !traverseR) {
                node = stack[--sp];
            } else {
                node = traverseL ? lChild.idx : rChild.idx;
                if (traverseL && traverseR) stack[sp++] = rChild.idx;
            }
        } while (node >= 0);
    }
    Hit throwRay(const Ray& ray, OracleCounters& c) const {  // kernel.cpp:218-240
        Hit nearestHit;
        c.rays++;
        if (traversal == 0) transverse(ray, nearestHit, c);
        else intersectRange(ray, 0, (int)tris.size(), nearestHit, c);
        return nearestHit;
    }

    // kernel.cpp:76-172
    void generateHitData(const ErMaterial& mat, HitData& hd, const Hit& hit, OracleCounters& c) const {
        V3 tangent = hit.tangent, bitangent = hit.bitangent;
        uint64_t* f = &c.texel_fetches;
        if (mat.albedo_tex < 0) hd.albedo = V3(mat.albedo.x, mat.albedo.y, mat.albedo.z);
        else hd.albedo = texFiltered(textures[mat.albedo_tex], hit.tu, hit.tv, f);
        if (mat.opacity_tex < 0) hd.opacity = mat.opacity;
        else hd.opacity = texFiltered(textures[mat.opacity_tex], hit.tu, hit.tv, f).x;
        if (mat.emission_tex < 0) hd.emission = V3(mat.emission.x, mat.emission.y, mat.emission.z);
        else hd.emission = texFiltered(textures[mat.emission_tex], hit.tu, hit.tv, f);
        if (mat.roughness_tex < 0) hd.roughness = mat.roughness;
        else hd.roughness = texFiltered(textures[mat.roughness_tex], hit.tu, hit.tv, f).x;
        if (mat.metallic_tex < 0) hd.metallic = mat.metallic;
        else hd.metallic = texFiltered(textures[mat.metallic_tex], hit.tu, hit.tv, f).x;
        if (mat.transmission_tex < 0) hd.transmission = mat.transmission;
        else hd.transmission = texFiltered(textures[mat.transmission_tex], hit.tu, hit.tv, f).x;
        if (mat.normal_tex < 0) {
            hd.normal = hit.normal;
        } else {
            V3 ncolor = texUV(textures[mat.normal_tex], hit.tu, hit.tv, f);
            V3 localNormal = (ncolor * 2) - V3(1.0f);
            hd.normal = normalized(localNormal.x * tangent - localNormal.y * bitangent + localNormal.z * hit.normal);
        }
        hd.roughness = m.pow(hd.roughness, 2.2f);
        hd.metallic = m.pow(hd.metallic, 2.2f);
        hd.clearcoatGloss = mat.clearcoat_gloss;
        hd.clearcoat = mat.clearcoat;
        hd.anisotropic = mat.anisotropic;
        hd.eta = mat.eta;
        hd.specular = mat.specular;
        hd.specularTint = mat.specular_tint;
        hd.sheenTint = mat.sheen_tint;
        hd.subsurface = mat.subsurface;
        hd.sheen = mat.sheen;
        hd.ax = mat.ax;
        hd.ay = mat.ay;
        hd.gnormal = hit.gnormal;
        hd.tangent = tangent;
        hd.bitangent = bitangent;
        hd.position = hit.position;
        hd.triIdx = hit.triIdx;
    }

    // kernel.cpp:371-473
    static void calculateCameraRay(const M& m, int x, int y, uint32_t x_res, uint32_t y_res, const ErCamera& cam,
                                   Ray& ray, float r1, float r2, float r3, float r4, float r5) {
        V3 cpos(cam.position.x, cam.position.y, cam.position.z);
        float dx = cpos.x + ((float)x) / ((float)x_res) * cam.sensor_width;
        float dy = cpos.y + ((float)y) / ((float)y_res) * cam.sensor_height;
        float odx = (-cam.sensor_width / 2.0f) + dx;
        float ody = (-cam.sensor_height / 2.0f) + dy;
        float rx = (1.0f / (float)x_res) * (r1 - 0.5f) * cam.sensor_width;
        float ry = (1.0f / (float)y_res) * (r2 - 0.5f) * cam.sensor_height;
        float SPx = odx + rx;
        float SPy = ody + ry;
        float SPz = cpos.z + cam.focal_length;
        V3 rotation(cam.rotation.x, cam.rotation.y, cam.rotation.z);
        rotation = rotation * (PIF / 180.0f);
        V3 dir = V3(SPx, SPy, SPz) - cpos;
        V3 dirXRot(dir.x, dir.y * m.cos(rotation.x) - dir.z * m.sin(rotation.x),
                   dir.y * m.sin(rotation.x) + dir.z * m.cos(rotation.x));
        V3 dirYRot(dirXRot.x * m.cos(rotation.y) + dirXRot.z * m.sin(rotation.y), dirXRot.y,
                   dirXRot.z * m.cos(rotation.y) - dirXRot.x * m.sin(rotation.y));
        V3 dirZRot(dirYRot.x * m.cos(rotation.z) - dirYRot.y * m.sin(rotation.z),
                   dirYRot.x * m.sin(rotation.z) + dirYRot.y * m.cos(rotation.z), dirYRot.z);
        ray = Ray(cpos, dirZRot);
        if (cam.bokeh) {
            float diameter = cam.focal_length / cam.aperture;
            float l = cam.focus_distance + cam.focal_length;
            V3 focusPoint = ray.origin + ray.direction * l;
            // uniformCircleSampling, Sampling.h:20-28
            float t = 2 * PIF * r3;
            float u = r4 + r5;
            float r = u > 1 ? 2 - u : u;
            float rIPx = r * m.cos(t);
            float rIPy = r * m.sin(t);
            rIPx *= diameter * 0.5f;
            rIPy *= diameter * 0.5f;
            V3 rIP(rIPx, rIPy, 0);
            V3 bX(rIP.x, rIP.y * m.cos(rotation.x) - rIP.z * m.sin(rotation.x),
                  rIP.y * m.sin(rotation.x) + rIP.z * m.cos(rotation.x));
            V3 bY(bX.x * m.cos(rotation.y) + bX.z * m.sin(rotation.y), bX.y,
                  bX.z * m.cos(rotation.y) - bX.x * m.sin(rotation.y));
            V3 bZ(bY.x * m.cos(rotation.z) - bY.y * m.sin(rotation.z),
                  bY.x * m.sin(rotation.z) + bY.y * m.cos(rotation.z), bY.z);
            V3 orig = cpos + bZ;
            ray = Ray(orig, focusPoint - orig);
        }
    }

    // kernel.cpp:477-646 ; the literal 5 of :508 is the max_bounces parameter
    void renderingKernel(uint32_t idx, OracleCounters& c, OracleTraceRec* recs, int max_recs, int* nrec) {
        if (idx >= x_res * y_res) return;
        Rng rnd{rng[idx]};
        uint32_t sa = samples[idx];
        Ray ray;
        int x = (int)(idx % x_res);
        int y = (int)(idx / x_res);
        float c1 = rnd.next(), c2 = rnd.next(), c3 = rnd.next(), c4 = rnd.next(), c5 = rnd.next();
        calculateCameraRay(m, x, y, x_res, y_res, cam, ray, c1, c2, c3, c4, c5);

        V3 light(0.0f), normal(0.0f), tangent(0.0f), bitangent(0.0f), reduction(1.0f);
        // the two build-defined extensions (rules: elevenrender_amd/csrc/er_shade.h; off = the reference)
        const bool mis = (flags & ER_FLAG_MIS) != 0;
        const bool use_lights = (flags & ER_FLAG_POINT_LIGHTS) != 0 && !lights.empty();
        float prev_pdf = -1.0f;     // ER_FLAG_MIS: brdfpdf of the last opaque bounce
        for (int i = 0; i < max_bounces; i++) {
            c.bounce_samples++;
            HitData hd;
            Hit nearestHit = throwRay(ray, c);
            OracleTraceRec* rec = (recs && *nrec < max_recs) ? &recs[(*nrec)++] : nullptr;
            if (rec) { memset(rec, 0, sizeof(*rec)); rec->bounce = i; rec->tri = nearestHit.valid ? nearestHit.tri : -1; rec->shadow_tri = -1; rec->shadow_occ = -1; rec->light_occ = -1; }
            if (!nearestHit.valid) {
                float u, v;
                sphericalMapping(m, V3(), -1 * ray.direction, 1, u, v);
                V3 env = texFiltered(hdri.texture, u, v, &c.texel_fetches);
                if (mis && prev_pdf >= 0.0f) {   // balance heuristic for the BRDF-sampled direction that left the scene
                    float p_h = hdriPdf(m, hdri, ermath::f2i(u * hdri.texture.width), ermath::f2i(v * hdri.texture.height), &c.texel_fetches);
                    env = env * (1.0f / (1.0f + p_h / prev_pdf));
                }
                light = light + reduction * env;
                if (rec) { rec->light[0] = light.x; rec->light[1] = light.y; rec->light[2] = light.z;
                           rec->reduction[0] = reduction.x; rec->reduction[1] = reduction.y; rec->reduction[2] = reduction.z; }
                break;
            }
            c.shaded_hits++;
            const ErMaterial& material = materials[nearestHit.material_id];
            generateHitData(material, hd, nearestHit, c);
            if (material.albedo_shader_id != -1) {
                hd.albedo = V3(0, 0, 0);
                // asl_shade: shader.cpp:6-10 dispatches ids 0..3 to the placeholder body (shader.h:10-11)
                if (material.albedo_shader_id >= 0 && material.albedo_shader_id < 4) hd.albedo = V3(1, 1, 0);
            }
            if (rnd.next() <= hd.opacity) {
                V3 wo = neg(ray.direction);
                c.hdri_samples++;
                V3 textCoordinate = hdriSample(hdri, rnd.next());
                float d1 = rnd.next(), d2 = rnd.next(), d3 = rnd.next();
                float rl = 0.0f;
                if (use_lights) rl = rnd.next();     // the light pick: one extra draw, right after DisneySample's three
                V3 wibrdf = DisneySample(m, hd, wo, hd.normal, d1, d2, d3);
                float nu = textCoordinate.x / (float)hdri.texture.width;
                float nv = textCoordinate.y / (float)hdri.texture.height;
                float iu = inverseTransformUV(hdri.texture, nu, nv).x;
                float iv = inverseTransformUV(hdri.texture, nu, nv).y;
                V3 wihdri = neg(normalized(reverseSphericalMapping(m, iu, iv)));
                Ray shadowRay(hd.position + hd.normal * 0.001f, wihdri);
                Hit shadowHit = throwRay(shadowRay, c);
                V3 hdriValue = texUV(hdri.texture, iu, iv, &c.texel_fetches);
                if (shadowHit.valid && shadowHit.triIdx != hd.triIdx) hdriValue = V3();
                float hdripdf = hdriPdf(m, hdri, ermath::f2i(iu * hdri.texture.width), ermath::f2i(iv * hdri.texture.height), &c.texel_fetches);
                V3 hdriInt = hdriValue * DisneyEval(m, hd, wo, hd.normal, wihdri) * std::fabs(dot(wihdri, hd.normal)) / hdripdf;
                if (mis) hdriInt = hdriInt * (1.0f / (1.0f + DisneyPdf(m, hd, wo, hd.normal, wihdri) / hdripdf));   // balance heuristic, NEE direction
                float brdfpdf = DisneyPdf(m, hd, wo, hd.normal, wibrdf);
                light = light + reduction * (hd.emission + hdriInt);
                if (use_lights) {
                    // the author's sketch pointLight(), kernel.cpp:269-301
                    int count = (int)lights.size();
                    int k = ermath::f2i((float)count * rl);                       // :280
                    if (k > count - 1) k = count - 1;                             // rnd.next() can return exactly 1.0
                    const ErPointLight& L = lights[k];
                    V3 lpos(L.position.x, L.position.y, L.position.z);
                    V3 point = hd.position;
                    V3 newDir = normalized(lpos - point);                         // :282
                    float dist = length(lpos - point);                            // :284
                    Ray shadowRay2(point + newDir * 0.001f, newDir);              // :287
                    V3 pointLightValue = V3(L.radiance.x, L.radiance.y, L.radiance.z) / (dist * dist);   // :295
                    V3 brdfDisney = DisneyEval(m, hd, wo, hd.normal, newDir);     // :297
                    float lpdf = ((float)count) / (2.0f * PIF);                   // :275
                    if (brdfDisney.x != 0.0f || brdfDisney.y != 0.0f || brdfDisney.z != 0.0f) {   // below the horizon: skipped
                        V3 pl = pointLightValue * brdfDisney * std::fabs(dot(newDir, hd.normal)) / lpdf;   // :299-300
                        Hit lh = throwRay(shadowRay2, c);                         // :288
                        // :289-291, both distances from the shadow ray's origin
                        const bool locc = lh.valid && length(lh.position - shadowRay2.origin) < length(lpos - shadowRay2.origin);
                        if (locc) pl = V3();
                        if (rec) rec->light_occ = locc ? 1 : 0;
                        light = light + reduction * pl;
                    }
                }
                reduction = reduction * (DisneyEval(m, hd, wo, hd.normal, wibrdf) * std::fabs(dot(wibrdf, hd.normal)) / brdfpdf);
                if (mis) prev_pdf = brdfpdf;
                if (i == 0) { normal = hd.normal; tangent = hd.tangent; bitangent = hd.bitangent; }
                ray = Ray(nearestHit.position + wibrdf * 0.001f, wibrdf);
                if (rec) { rec->opaque = 1; rec->shadow_tri = shadowHit.valid ? shadowHit.tri : -1;
                           rec->shadow_occ = (shadowHit.valid && shadowHit.triIdx != hd.triIdx) ? 1 : 0; }
            } else {
                ray = Ray(nearestHit.position + ray.direction * 0.001f, ray.direction);
            }
            if (rec) {
                rec->position[0] = nearestHit.position.x; rec->position[1] = nearestHit.position.y; rec->position[2] = nearestHit.position.z;
                rec->wi[0] = ray.direction.x; rec->wi[1] = ray.direction.y; rec->wi[2] = ray.direction.z;
                rec->light[0] = light.x; rec->light[1] = light.y; rec->light[2] = light.z;
                rec->reduction[0] = reduction.x; rec->reduction[1] = reduction.y; rec->reduction[2] = reduction.z;
            }
        }
        light = V3(clampf(light.x, 0, 10), clampf(light.y, 0, 10), clampf(light.z, 0, 10));
        size_t plane = (size_t)x_res * y_res * 4;
        float* P = passes.data();
        if (!(light.x != light.x) && !(light.y != light.y) && !(light.z != light.z)) {
            if (sa > 0) {
                for (int pass = 0; pass < ER_PASS_COUNT; pass++) {
                    if (pass != ER_PASS_DENOISE) {
                        P[pass * plane + 4 * (size_t)idx + 0] *= ((float)sa) / ((float)(sa + 1));
                        P[pass * plane + 4 * (size_t)idx + 1] *= ((float)sa) / ((float)(sa + 1));
                        P[pass * plane + 4 * (size_t)idx + 2] *= ((float)sa) / ((float)(sa + 1));
                    }
                }
            }
            const V3* vals[4] = {&light, &normal, &tangent, &bitangent};
            const int pl[4] = {ER_PASS_BEAUTY, ER_PASS_NORMAL, ER_PASS_TANGENT, ER_PASS_BITANGENT};
            for (int k = 0; k < 4; k++) {
                P[pl[k] * plane + 4 * (size_t)idx + 0] += vals[k]->x / ((float)(sa + 1));
                P[pl[k] * plane + 4 * (size_t)idx + 1] += vals[k]->y / ((float)(sa + 1));
                P[pl[k] * plane + 4 * (size_t)idx + 2] += vals[k]->z / ((float)(sa + 1));
            }
            samples[idx]++;
        }
        rng[idx] = rnd.state;
        c.paths++;
    }

    void setup() {  // setupKernel, kernel.cpp:176-213, for every pixel
        size_t npx = (size_t)x_res * y_res;
        passes.assign(npx * 4 * ER_PASS_COUNT, 0.0f);
        for (int p = 0; p < ER_PASS_COUNT; p++)
            for (size_t i = 0; i < npx; i++) passes[p * npx * 4 + 4 * i + 3] = 1.0f;
        samples.assign(npx, 1u);
        rng.resize(npx);
        for (size_t i = 0; i < npx; i++) rng[i] = jenkins_u32((uint32_t)i + 1);
        memset(&ctr, 0, sizeof(ctr));
    }
};

static void add_ctr(OracleCounters& a, const OracleCounters& b) {
    a.paths += b.paths; a.bounce_samples += b.bounce_samples; a.rays += b.rays; a.node_visits += b.node_visits;
    a.tri_tests += b.tri_tests; a.tri_hits += b.tri_hits; a.shaded_hits += b.shaded_hits;
    a.texel_fetches += b.texel_fetches; a.hdri_samples += b.hdri_samples;
}

static Tex make_tex(const ErTexture& t) {
    Tex r;
    r.width = t.width; r.height = t.height; r.channels = t.channels; r.filter = t.filter;
    size_t n = (size_t)t.width * t.height * (t.channels > 0 ? t.channels : 0);
    r.data.assign(t.data, t.data + n);
    return r;
}

extern "C" {

Oracle* oracle_create(const ErSceneDesc* d, const OracleOpts* opts) {
    Oracle* o = new Oracle();
    o->m.mode = opts ? opts->math_mode : 0;
    o->max_bounces = (opts && opts->max_bounces > 0) ? opts->max_bounces : 5;
    o->traversal = opts ? opts->traversal : 0;
    o->threads = (opts && opts->threads > 0) ? opts->threads : 1;
    o->flags = opts ? opts->flags : 0;
    if (d->point_light_count && d->point_lights) o->lights.assign(d->point_lights, d->point_lights + d->point_light_count);
    o->cam = d->camera;
    o->x_res = d->x_res; o->y_res = d->y_res;
    o->tris.resize(d->tri_count);
    for (uint32_t i = 0; i < d->tri_count; i++) {
        Tri& t = o->tris[i];
        for (int k = 0; k < 3; k++) {
            t.v[k] = V3(d->vertices[i * 9 + k * 3], d->vertices[i * 9 + k * 3 + 1], d->vertices[i * 9 + k * 3 + 2]);
            t.n[k] = V3(d->normals[i * 9 + k * 3], d->normals[i * 9 + k * 3 + 1], d->normals[i * 9 + k * 3 + 2]);
            t.t[k] = V3(d->tangents[i * 9 + k * 3], d->tangents[i * 9 + k * 3 + 1], d->tangents[i * 9 + k * 3 + 2]);
            t.uvx[k] = d->uvs[i * 6 + k * 2];
            t.uvy[k] = d->uvs[i * 6 + k * 2 + 1];
        }
        t.sign = d->tangent_sign[i];
        t.material = d->material_id[i];
    }
    o->materials.assign(d->materials, d->materials + d->material_count);
    for (uint32_t i = 0; i < d->texture_count; i++) o->textures.push_back(make_tex(d->textures[i]));
    o->hdri.texture = make_tex(d->hdri.texture);
    if (d->hdri.cdf) {
        o->hdri.cdf.assign(d->hdri.cdf, d->hdri.cdf + (size_t)d->hdri.texture.width * d->hdri.texture.height + 1);
        o->hdri.radianceSum = d->hdri.radiance_sum;
    } else {
        generateCDF(o->hdri.texture, o->hdri.cdf, o->hdri.radianceSum);
    }
    auto t0 = std::chrono::steady_clock::now();
    if (o->traversal == 0) o->bvh.build(o->tris);
    o->build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    o->setup();
    return o;
}
void oracle_destroy(Oracle* o) { delete o; }
double oracle_build_seconds(const Oracle* o) { return o->build_s; }

void oracle_render(Oracle* o, uint32_t n_samples, uint32_t idx0, uint32_t idx1) {
    uint32_t npx = o->x_res * o->y_res;
    if (idx1 == 0 || idx1 > npx) idx1 = npx;
    int nt = o->threads;
    std::vector<OracleCounters> ctrs(nt);
    for (auto& c : ctrs) memset(&c, 0, sizeof(c));
    auto work = [&](int tid) {
        uint32_t span = idx1 - idx0;
        uint32_t a = idx0 + (uint32_t)((uint64_t)span * tid / nt), b = idx0 + (uint32_t)((uint64_t)span * (tid + 1) / nt);
        OracleCounters local;   // thread-private (adjacent vector slots would false-share)
        memset(&local, 0, sizeof(local));
        for (uint32_t s = 0; s < n_samples; s++)
            for (uint32_t idx = a; idx < b; idx++) o->renderingKernel(idx, local, nullptr, 0, nullptr);
        ctrs[tid] = local;
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    for (auto& c : ctrs) add_ctr(o->ctr, c);
}
void oracle_read_pass(const Oracle* o, int pass, float* dst) {
    size_t plane = (size_t)o->x_res * o->y_res * 4;
    memcpy(dst, o->passes.data() + pass * plane, plane * sizeof(float));
}
void oracle_read_samples(const Oracle* o, uint32_t* dst) { memcpy(dst, o->samples.data(), o->samples.size() * 4); }
void oracle_read_rng(const Oracle* o, uint32_t* dst) { memcpy(dst, o->rng.data(), o->rng.size() * 4); }
void oracle_counters(const Oracle* o, OracleCounters* out) { *out = o->ctr; }

int oracle_trace_pixel(Oracle* o, uint32_t idx, OracleTraceRec* recs, int max_recs) {
    int n = 0;
    OracleCounters c;
    memset(&c, 0, sizeof(c));
    o->renderingKernel(idx, c, recs, max_recs, &n);
    add_ctr(o->ctr, c);
    return n;
}
void oracle_closest_hit(Oracle* o, const float* origins, const float* dirs, int n, int32_t* out_tri, float* out_pos) {
    OracleCounters c;
    memset(&c, 0, sizeof(c));
    for (int i = 0; i < n; i++) {
        Ray r;  // direction taken as given (callers pass what Ray's ctor would have produced)
        r.origin = V3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
        r.direction = V3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
        Hit h = o->throwRay(r, c);
        out_tri[i] = h.valid ? h.tri : -1;
        out_pos[3 * i] = h.position.x; out_pos[3 * i + 1] = h.position.y; out_pos[3 * i + 2] = h.position.z;
    }
}

uint32_t oracle_jenkins_oaat_u32(uint32_t seed) { return jenkins_u32(seed); }
uint32_t oracle_jenkins_oaat_bytes(const uint8_t* key, size_t len) {
    // Bob Jenkins' one-at-a-time hash as published (Dr. Dobb's, 1997)
    uint32_t hash = 0;
    for (size_t i = 0; i < len; i++) { hash += key[i]; hash += (hash << 10); hash ^= (hash >> 6); }
    hash += (hash << 3); hash ^= (hash >> 11); hash += (hash << 15);
    return hash;
}
void oracle_xorshift32(uint32_t* state) { Rng r{*state}; r.next(); *state = r.state; }
void oracle_rng_stream(uint32_t pixel_idx, int n, uint32_t* states, float* values) {
    Rng r{jenkins_u32(pixel_idx + 1)};  // RngGenerator(idx), kernel.cpp:38-40,183
    for (int i = 0; i < n; i++) { values[i] = r.next(); states[i] = r.state; }
}
void oracle_camera_ray(const ErCamera* cam, uint32_t x_res, uint32_t y_res, int x, int y, const float r[5],
                       int math_mode, float out_origin[3], float out_dir[3]) {
    M m{math_mode};
    Ray ray;
    Oracle::calculateCameraRay(m, x, y, x_res, y_res, *cam, ray, r[0], r[1], r[2], r[3], r[4]);
    out_origin[0] = ray.origin.x; out_origin[1] = ray.origin.y; out_origin[2] = ray.origin.z;
    out_dir[0] = ray.direction.x; out_dir[1] = ray.direction.y; out_dir[2] = ray.direction.z;
}
int oracle_tri_hit(const float* verts, const float* normals, const float* tangents, const float* uvs,
                   float tangent_sign, const float origin[3], const float dir[3], float out[17]) {
    Tri t;
    for (int k = 0; k < 3; k++) {
        t.v[k] = V3(verts[3 * k], verts[3 * k + 1], verts[3 * k + 2]);
        t.n[k] = V3(normals[3 * k], normals[3 * k + 1], normals[3 * k + 2]);
        t.t[k] = V3(tangents[3 * k], tangents[3 * k + 1], tangents[3 * k + 2]);
        t.uvx[k] = uvs[2 * k]; t.uvy[k] = uvs[2 * k + 1];
    }
    t.sign = tangent_sign; t.material = 0;
    Ray r;
    r.origin = V3(origin[0], origin[1], origin[2]);
    r.direction = V3(dir[0], dir[1], dir[2]);
    Hit h;
    if (!tri_hit(t, r, h)) return 0;
    const V3* v[5] = {&h.position, &h.normal, &h.gnormal, &h.tangent, &h.bitangent};
    for (int k = 0; k < 5; k++) { out[3 * k] = v[k]->x; out[3 * k + 1] = v[k]->y; out[3 * k + 2] = v[k]->z; }
    out[15] = h.tu; out[16] = h.tv;
    return 1;
}
int oracle_box_hit(const float origin[3], const float dir[3], const float b1[3], const float b2[3]) {
    Ray r;
    r.origin = V3(origin[0], origin[1], origin[2]);
    r.direction = V3(dir[0], dir[1], dir[2]);
    return RefBVH::intersect(r, V3(b1[0], b1[1], b1[2]), V3(b2[0], b2[1], b2[2])) ? 1 : 0;
}
static HitData hd_from(const float h[20]) {
    HitData d;
    memset(&d, 0, sizeof(d));
    d.metallic = h[0]; d.roughness = h[1]; d.clearcoatGloss = h[2]; d.clearcoat = h[3]; d.anisotropic = h[4];
    d.transmission = h[5]; d.specular = h[6]; d.specularTint = h[7]; d.sheenTint = h[8]; d.subsurface = h[9];
    d.sheen = h[10]; d.albedo = V3(h[11], h[12], h[13]); d.tangent = V3(h[14], h[15], h[16]);
    d.bitangent = V3(h[17], h[18], h[19]);
    return d;
}
void oracle_disney_eval(const float hd[20], const float V[3], const float N[3], const float L[3], int mode, float out[3]) {
    M m{mode};
    V3 r = DisneyEval(m, hd_from(hd), V3(V[0], V[1], V[2]), V3(N[0], N[1], N[2]), V3(L[0], L[1], L[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float oracle_disney_pdf(const float hd[20], const float V[3], const float N[3], const float L[3], int mode) {
    M m{mode};
    return DisneyPdf(m, hd_from(hd), V3(V[0], V[1], V[2]), V3(N[0], N[1], N[2]), V3(L[0], L[1], L[2]));
}
void oracle_disney_sample(const float hd[20], const float V[3], const float N[3], float r1, float r2, float r3,
                          int mode, float out[3]) {
    M m{mode};
    V3 r = DisneySample(m, hd_from(hd), V3(V[0], V[1], V[2]), V3(N[0], N[1], N[2]), r1, r2, r3);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oracle_spherical_mapping(const float p[3], int mode, float* u, float* v) {
    M m{mode};
    sphericalMapping(m, V3(), V3(p[0], p[1], p[2]), 1, *u, *v);
}
void oracle_reverse_spherical_mapping(float u, float v, int mode, float out[3]) {
    M m{mode};
    V3 r = reverseSphericalMapping(m, u, v);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oracle_texture_fetch(const ErTexture* tex, float u, float v, int filtered, float out[3]) {
    Tex t = make_tex(*tex);
    V3 r = filtered ? texFiltered(t, u, v, nullptr) : texUV(t, u, v, nullptr);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void oracle_hdri_cdf(const ErTexture* tex, float* cdf, float* radiance_sum) {
    Tex t = make_tex(*tex);
    std::vector<float> c;
    generateCDF(t, c, *radiance_sum);
    memcpy(cdf, c.data(), c.size() * sizeof(float));
}
int oracle_hdri_binary_search(const float* cdf, float value, int length) { return binarySearch(cdf, value, length); }
float oracle_hdri_pdf(const ErTexture* tex, float radiance_sum, int x, int y, int mode) {
    M m{mode};
    Hdri h;
    h.texture = make_tex(*tex);
    h.radianceSum = radiance_sum;
    return hdriPdf(m, h, x, y, nullptr);
}
void oracle_math(int kind, int mode, const float* x, const float* y, float* out, int n) {
    M m{mode};
    for (int i = 0; i < n; i++) {
        switch (kind) {
            case 0: out[i] = m.sin(x[i]); break;
            case 1: out[i] = m.cos(x[i]); break;
            case 2: out[i] = m.acos(x[i]); break;
            case 3: out[i] = m.log(x[i]); break;
            case 4: out[i] = m.pow(x[i], y[i]); break;
            default: out[i] = m.atan2(x[i], y[i]); break;
        }
    }
}

}  // extern "C"
