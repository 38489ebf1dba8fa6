// er_device.h -- device-side scene view and the per-sample path functions for gfx950.
//
// Arithmetic contract: every float expression below associates exactly as the reference
// source parses (file:line cited per function) and is compiled with -ffp-contract=off, so
// that the only differences from a CPU evaluation of the reference are (a) the
// transcendental functions (er_math.h) and (b) the acceleration structure, which changes
// no result except the winner among hits at EXACTLY equal distance.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/eleven_hip.h"
#include "er_bvh.h"
#include "er_cdf.h"
#include "er_math.h"

// ISA region marks for tools/isa_budget.py: with -DER_ISA_MARKS every ER_MARK("name") leaves an assembler comment in the
// device assembly (hipcc -S) at that point of the instruction stream; the product build compiles them to nothing.
#ifdef ER_ISA_MARKS
#define ER_MARK(name) asm volatile("; ER_MARK " name ::: "memory")
#else
#define ER_MARK(name) ((void)0)
#endif

#ifndef ER_TP
#define ER_TP(n) ((void)0)       // section stamp of the streaming kernel's diagnostic build (-DER_TIME_PROBE), nothing otherwise
#endif

#define ER_TILE 8                 // 8x8 pixel tile = one 64-lane wavefront
#define ER_STACK ER_BVH_MAX_DEPTH

struct DevTex {
    int32_t width, height, channels, filter;
    uint32_t offset;   // into tex_pool (floats)
};

// A material whose albedo, roughness and metallic textures have one size and one filter: the three texel by texel, five floats per texel
// (albedo r g b as Texture::getValueFromCoordinates returns it, roughness, metallic = the first channels), made by er_api.cpp.  One fetch
// (one cache line, one coordinate computation) instead of three per hit.  filter: 1 bilinear, 0 unfiltered, 2 unfiltered with roughness
// and metallic already to the power 2.2 (src/kernel.cpp:152-153); width = 0: the material is not fused.
struct DevFused {
    int32_t width, height, filter;
    uint32_t offset;   // into tex_pool (floats)
};

struct DevCounters {
    unsigned long long paths, bounce_samples, rays, node_visits, tri_tests, shaded_hits, texel_fetches, hdri_samples;
    unsigned long long trace_wave_steps, trace_busy_lanes, trace_node_lanes, trace_tri_lanes;
};

struct DevScene {
    // acceleration structure + geometry (leaf order)
    const float4* nodes;        // 4 x float4 per node (binary tree: megakernel + exact re-trace)
    const float4* nodes8;       // 5 x float4 per node (8-wide compressed tree: er_wf_trace)
    const float4* tri_isect;    // 3 x float4 per slot
    const float4* tri_attr;     // 7 x float4 per slot
    uint32_t tri_count, node_count;
    uint32_t node8_count;       // wide nodes in nodes8
    uint32_t tri_base_pieces;   // tri_isect == nodes8 + tri_base_pieces (16-byte pieces): one buffer, 32-bit offsets
    float prune_margin;         // lift bound + rounding slack, see trace()
    float max_lift, scene_scale; // inputs of the interval bookkeeping in er_wf_trace
    // shading
    const ErMaterial* materials;
    // per material, evaluated once on the host with the same ermath functions (one implementation for both sides: the same bits):
    // x = er_pow(roughness, 2.2f), y = er_pow(metallic, 2.2f) -- used when that channel has no texture (src/kernel.cpp:152-153 computes
    // them at every hit) --, z = er_log(a * a) for a = lerp(0.1, 0.001, clearcoatGloss), the logarithm GTR1 takes at every BRDF
    // evaluation (src/Disney.cpp:40-46); w = 1 if z is valid (a < 1)
    const float4* mat_pre;
    const DevTex* textures;
    const DevFused* mat_fused;  // per material (above)
    uint32_t fused_any;         // 1 if any material is fused
    const float* tex_pool;
    DevTex hdri_tex;
    const float* hdri_cdf;
    const uint32_t* hdri_guide;  // er_cdf.h
    int32_t hdri_buckets;
    float hdri_radiance_sum;
    ErCamera cam;
    uint32_t x_res, y_res, tiles_x, tiles_y;
    uint32_t max_bounces;
    uint32_t tex_pow2;          // 1 when every texture and the HDRI have power-of-two sides (tex_coords then wraps with a mask)
    // the camera's six rotation sines / cosines (src/kernel.cpp:371-473 recomputes them for every sample): evaluated ONCE on the host
    // with the same ermath functions -- er_math.h is one implementation for both sides, binary64 without FMA: the same bits -- by
    // camera_trig() below; cam_trig_valid = 0 makes camera_ray compute them itself (the other schedules' kernels leave it 0 or 1 alike)
    float cam_cx, cam_sx, cam_cy, cam_sy, cam_cz, cam_sz;
    uint32_t cam_trig_valid;
    // build-defined extensions (er_shade.h): ER_FLAG_POINT_LIGHTS / ER_FLAG_MIS bits of the render flags
    uint32_t ext_flags;
    const ErPointLight* lights;
    uint32_t light_count;
    // per-pixel state
    float4* passes;             // ER_PASS_COUNT x (x_res*y_res) float4 in the layout of er_pass_index (below): NOT plane after plane
    uint32_t* samples;
    uint32_t* rng;
    const uint32_t* owned_tiles;
    uint32_t owned_tile_count;
    DevCounters* counters;
    // streaming schedule, while the deal of tiles to the XCDs is undecided (er_api.cpp er_stream_adapt): per tile of the frame, the sum
    // of the path lengths of its finished samples -- WORK counted by the kernel, the same on every run of the same frame; NULL = do not count
    uint32_t* tile_cost;
    // streaming schedule, 12-wave form (er_stream.hip, speculative sample pipelining): per pixel, the draw count its samples are guessed to
    // have (bits 0-6: how many random numbers a sample draws; 0 = none yet) and a saturating confidence in it (bits 8-10: + 1 for every
    // accumulated sample that drew that many, - 2 for every other one; at 0 the latest count becomes the guess).  A guess, never a
    // result: not part of the progressive state (er_state_export), zeroed by er_render_begin
    uint32_t* px_draws;
};

// Where pass `pass` of pixel `idx` lives in DevScene::passes.  The four planes a finished sample is accumulated into -- beauty, normal,
// tangent, bitangent -- are INTERLEAVED per pixel (64 bytes: one half cache line instead of four lines for the read-modify-write of the
// accumulate step: +1.8 % on C2, profiles/r03_ab_interleaved_planes_probe.log); the DENOISE plane, which only er_denoise writes, follows
// as a plane of its own.  Every access goes through this function; the ABI's plane-after-plane view is made at the boundary
// (er_read_pass gathers a plane, er_pack / er_unpack address pixels one by one, the checkpoint blob is the raw buffer).
__host__ __device__ __forceinline__ size_t er_pass_index(size_t npx, int pass, size_t idx) {
    return pass == ER_PASS_DENOISE ? 4 * npx + idx : idx * 4 + (size_t)(pass == ER_PASS_BEAUTY ? 0 : pass - 1);
}


namespace erd {

#define ERD __device__ __forceinline__

const float PIF = 3.14159265358979323846f;   // src/Math.hpp:6

struct F3 { float x, y, z; };
ERD F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
ERD F3 f3s(float s) { return f3(s, s, s); }
ERD F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
ERD F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
ERD F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
ERD F3 operator*(float s, F3 a) { return f3(a.x * s, a.y * s, a.z * s); }
ERD F3 operator/(F3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
ERD F3 operator*(F3 a, F3 b) { return f3(b.x * a.x, b.y * a.y, b.z * a.z); }
ERD F3 addf(F3 a, float s) { return f3(a.x + s, a.y + s, a.z + s); }
ERD float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
ERD F3 cross(F3 a, F3 b) {   // sign convention of src/Vector.h:173-175
    return f3((a.y * b.z - a.z * b.y), -(a.x * b.z - a.z * b.x), (a.x * b.y - a.y * b.x));
}
ERD float length(F3 a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
ERD F3 normalized(F3 a) {    // src/Vector.h:186-189
    float l = length(a);
    if (l == 0) return a;
    return a / l;
}
ERD float clampf(float a, float b, float c) { return a < b ? b : a > c ? c : a; }   // src/Math.hpp:30-32
ERD float lerpf(float a, float b, float c) { return a + c * (b - a); }              // src/Math.hpp:38-41
ERD F3 lerpv(F3 a, F3 b, float c) { return f3(lerpf(a.x, b.x, c), lerpf(a.y, b.y, c), lerpf(a.z, b.z, c)); }
ERD float minf(float a, float b) { return a < b ? a : b; }
ERD float maxf(float a, float b) { return a > b ? a : b; }
ERD void limitUV(float& u, float& v) {   // src/Math.hpp:48-51
    u += (float)(-(int)(u > 1) + -(int)(u < 0));
    v += (float)(-(int)(v > 1) + -(int)(v < 0));
}

struct Ray { F3 o, d; };
ERD Ray make_ray(F3 o, F3 d) {   // Ray ctor normalises, src/Ray.h:13-17
    Ray r;
    r.o = o;
    float l = length(d);
    if (l != 0) { d.x /= l; d.y /= l; d.z /= l; }
    r.d = d;
    return r;
}

// ---- RNG, src/kernel.cpp:25-47 -------------------------------------------------
ERD uint32_t jenkins_u32(uint32_t seed) {
    uint32_t hash = 0;
#pragma unroll
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
ERD float rng_next(uint32_t& state) {
    state ^= state << 13;
    state ^= state >> 17;
    state ^= state << 5;
    return (float)state / 4294967296.0f;
}

// ---- camera, src/kernel.cpp:371-473 ----------------------------------------------
struct CamTrig { float cx, sx, cy, sy, cz, sz; };
// the rotation's sines and cosines exactly as camera_ray evaluates them (host and device: the same expressions on the same ermath)
__host__ __device__ inline CamTrig camera_trig(const ErCamera& cam) {
    using namespace ermath;
    const float k = 3.14159265358979323846f / 180.0f;      // PIF / 180.0f
    const float rx = cam.rotation.x * k, ry = cam.rotation.y * k, rz = cam.rotation.z * k;
    CamTrig t;
    t.cx = er_cos(rx); t.sx = er_sin(rx); t.cy = er_cos(ry); t.sy = er_sin(ry); t.cz = er_cos(rz); t.sz = er_sin(rz);
    return t;
}
ERD Ray camera_ray(const ErCamera& cam, int x, int y, uint32_t x_res, uint32_t y_res,
                   float r1, float r2, float r3, float r4, float r5, const CamTrig* pre = nullptr) {
    using namespace ermath;
    F3 cpos = f3(cam.position.x, cam.position.y, cam.position.z);
    float dx = cpos.x + ((float)x) / ((float)x_res) * cam.sensor_width;
    float dy = cpos.y + ((float)y) / ((float)y_res) * cam.sensor_height;
    float odx = (-cam.sensor_width / 2.0f) + dx;
    float ody = (-cam.sensor_height / 2.0f) + dy;
    float rx = (1.0f / (float)x_res) * (r1 - 0.5f) * cam.sensor_width;
    float ry = (1.0f / (float)y_res) * (r2 - 0.5f) * cam.sensor_height;
    float SPx = odx + rx, SPy = ody + ry, SPz = cpos.z + cam.focal_length;
    const CamTrig tr = pre ? *pre : camera_trig(cam);      // (rot = rotation * (PI / 180), then the six er_cos / er_sin: camera_trig)
    const float cx = tr.cx, sx = tr.sx, cy = tr.cy, sy = tr.sy, cz = tr.cz, sz = tr.sz;
    F3 dir = f3(SPx, SPy, SPz) - cpos;
    F3 dX = f3(dir.x, dir.y * cx - dir.z * sx, dir.y * sx + dir.z * cx);
    F3 dY = f3(dX.x * cy + dX.z * sy, dX.y, dX.z * cy - dX.x * sy);
    F3 dZ = f3(dY.x * cz - dY.y * sz, dY.x * sz + dY.y * cz, dY.z);
    Ray ray = make_ray(cpos, dZ);
    if (cam.bokeh) {
        float diameter = cam.focal_length / cam.aperture;
        float l = cam.focus_distance + cam.focal_length;
        F3 focusPoint = ray.o + ray.d * l;
        float t = 2 * PIF * r3;                 // uniformCircleSampling, src/Sampling.h:20-28
        float u = r4 + r5;
        float r = u > 1 ? 2 - u : u;
        float rIPx = r * er_cos(t), rIPy = r * er_sin(t);
        rIPx *= diameter * 0.5f;
        rIPy *= diameter * 0.5f;
        F3 rIP = f3(rIPx, rIPy, 0);
        F3 bX = f3(rIP.x, rIP.y * cx - rIP.z * sx, rIP.y * sx + rIP.z * cx);
        F3 bY = f3(bX.x * cy + bX.z * sy, bX.y, bX.z * cy - bX.x * sy);
        F3 bZ = f3(bY.x * cz - bY.y * sz, bY.x * sz + bY.y * cz, bY.z);
        F3 orig = cpos + bZ;
        ray = make_ray(orig, focusPoint - orig);
    }
    return ray;
}

// ---- triangle, src/Tri.h:41-144 ----------------------------------------------------
struct HitFull {
    F3 position, normal, tangent, bitangent, gnormal;
    float tu, tv;
    int material;
};

ERD F3 project_on_plane(F3 position, F3 origin, F3 normal) {   // src/Tri.h:37-39
    return position - dot(position - origin, normal) * normal;
}

// Moller-Trumbore exactly as the reference evaluates it; returns false on reject.
ERD bool tri_mt(F3 v0, F3 v1, F3 v2, const Ray& ray, float& u, float& v, float& t) {
    const float EPSILON = 0.0000001f;
    F3 edge1 = v1 - v0, edge2 = v2 - v0;
    F3 pvec = cross(ray.d, edge2);
    float det = dot(edge1, pvec);
    float inv_det = 1.0f / det;
    if (det > -EPSILON && det < EPSILON) return false;
    F3 tvec = ray.o - v0;
    u = dot(tvec, pvec) * inv_det;
    if (u < 0 || u > 1) return false;
    F3 qvec = cross(tvec, edge1);
    v = dot(ray.d, qvec) * inv_det;
    if (v < 0 || (u + v) > 1) return false;
    t = dot(edge2, qvec) * inv_det;
    if (t < 0) return false;
    return true;
}

// Hit.position of the reference (shading position when "convex") and the shading normal.
ERD F3 hit_position(F3 v0, F3 v1, F3 v2, F3 n0, F3 n1, F3 n2, const Ray& ray, float u, float v, float t, F3& shadingNormal) {
    F3 geomPosition = ray.o + ray.d * t;
    shadingNormal = normalized(n0 + (n1 - n0) * u + (n2 - n0) * v);
    F3 p0 = project_on_plane(geomPosition, v0, n0);
    F3 p1 = project_on_plane(geomPosition, v1, n1);
    F3 p2 = project_on_plane(geomPosition, v2, n2);
    F3 shadingPosition = p0 + (p1 - p0) * u + (p2 - p0) * v;
    bool convex = dot(shadingPosition - geomPosition, shadingNormal) > 0;
    return convex ? shadingPosition : geomPosition;
}

ERD void load_verts(const DevScene& S, uint32_t slot, F3& v0, F3& v1, F3& v2, float4& a, float4& b, float4& c) {
    const float4* p = S.tri_isect + (size_t)slot * 3;
    a = p[0]; b = p[1]; c = p[2];
    v0 = f3(a.x, a.y, a.z); v1 = f3(b.x, b.y, b.z); v2 = f3(c.x, c.y, c.z);
}
ERD void load_normals(const DevScene& S, uint32_t slot, F3& n0, F3& n1, F3& n2) {
    const float4* p = S.tri_attr + (size_t)slot * ER_ATTR_PIECES;
    float4 a = p[0], b = p[1], c = p[2];
    n0 = f3(a.x, a.y, a.z); n1 = f3(a.w, b.x, b.y); n2 = f3(b.z, b.w, c.x);
}

// the reference metric of a candidate: |Hit.position - ray.origin| (src/BVH.cpp:114)
ERD float candidate_distance(const DevScene& S, uint32_t slot, F3 v0, F3 v1, F3 v2, const Ray& ray, float u, float v, float t) {
    F3 n0, n1, n2, sn;
    load_normals(S, slot, n0, n1, n2);
    F3 pos = hit_position(v0, v1, v2, n0, n1, n2, ray, u, v, t, sn);
    return length(pos - ray.o);
}

// full Hit record of the winning slot (recomputed: same expressions, same values)
ERD void full_hit(const DevScene& S, uint32_t slot, const Ray& ray, HitFull& h) {
    F3 v0, v1, v2;
    float4 a, b, c;
    load_verts(S, slot, v0, v1, v2, a, b, c);
    float u = 0, v = 0, t = 0;
    tri_mt(v0, v1, v2, ray, u, v, t);
    const float4* p = S.tri_attr + (size_t)slot * ER_ATTR_PIECES;
    float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4], q5 = p[5], q6 = p[6];
    F3 n0 = f3(q0.x, q0.y, q0.z), n1 = f3(q0.w, q1.x, q1.y), n2 = f3(q1.z, q1.w, q2.x);
    F3 t0 = f3(q2.y, q2.z, q2.w), t1 = f3(q3.x, q3.y, q3.z), t2 = f3(q3.w, q4.x, q4.y);
    float uv0x = q4.z, uv0y = q4.w, uv1x = q5.x, uv1y = q5.y, uv2x = q5.z, uv2y = q5.w;
    F3 edge1 = v1 - v0, edge2 = v2 - v0;
    h.tu = uv0x + (uv1x - uv0x) * u + (uv2x - uv0x) * v;
    h.tv = uv0y + (uv1y - uv0y) * u + (uv2y - uv0y) * v;
    F3 sn;
    h.position = hit_position(v0, v1, v2, n0, n1, n2, ray, u, v, t, sn);
    h.normal = sn;
    F3 compNormal = normalized(cross(edge1, edge2));
    if (dot(compNormal, ray.d) > 0) compNormal = compNormal * -1.0f;
    h.gnormal = compNormal;
    h.tangent = t0 + (t1 - t0) * u + (t2 - t0) * v;
    h.bitangent = __builtin_bit_cast(float, c.w) * cross(h.normal, h.tangent);
    h.material = __builtin_bit_cast(int, q6.x);
}

// ---- box test, src/BVH.cpp:27-61 (formula kept; dirfrac hoisted out of the loop) ----
ERD bool box_test(F3 lo, F3 hi, F3 o, F3 dirfrac, float bound, float& tmin_out) {
    float t1 = (lo.x - o.x) * dirfrac.x;
    float t2 = (hi.x - o.x) * dirfrac.x;
    float t3 = (lo.y - o.y) * dirfrac.y;
    float t4 = (hi.y - o.y) * dirfrac.y;
    float t5 = (lo.z - o.z) * dirfrac.z;
    float t6 = (hi.z - o.z) * dirfrac.z;
    float tmin = maxf(maxf(minf(t1, t2), minf(t3, t4)), minf(t5, t6));
    float tmax = minf(minf(maxf(t1, t2), maxf(t3, t4)), maxf(t5, t6));
    tmin_out = tmin;
    if (tmax < 0) return false;
    if (tmin > tmax) return false;
    // ordered traversal prunes what the reference would test and then lose on distance
    return !(tmin > bound);
}

// Closest hit under the reference's metric.  `skip_slot`/`limit`: used by the shadow query
// (occluded()) -- ignore one slot, and stop at the first candidate closer than `limit`.
// Returns the winning slot or -1.
template <bool COUNT, bool ANY>
ERD int trace(const DevScene& S, int* stack /* LDS, stride 64 ints */, const Ray& ray, int skip_slot, float limit,
              float& best_dist, unsigned& node_visits, unsigned& tri_tests) {
    int best = -1;
    best_dist = limit;
    if (S.node_count == 0) return -1;
    float bound = ANY ? limit * 1.00001f + S.prune_margin : __builtin_inff();
    F3 dirfrac = f3(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    int sp = 0;
    int cur = 0;
    while (true) {
        while ((unsigned)cur < (unsigned)ER_BVH_NO_CHILD) {   // inner node
            const float4* np = S.nodes + (size_t)cur * 4;
            float4 a = np[0], b = np[1], c = np[2], d = np[3];
            if (COUNT) node_visits++;
            int c0 = __builtin_bit_cast(int, d.x), c1 = __builtin_bit_cast(int, d.y);
            float tm0, tm1;
            bool h0 = box_test(f3(a.x, a.y, a.z), f3(a.w, b.x, b.y), ray.o, dirfrac, bound, tm0) && c0 != ER_BVH_NO_CHILD;
            bool h1 = box_test(f3(b.z, b.w, c.x), f3(c.y, c.z, c.w), ray.o, dirfrac, bound, tm1) && c1 != ER_BVH_NO_CHILD;
            if (h0 && h1) {
                bool swap = tm1 < tm0;
                int nearc = swap ? c1 : c0, farc = swap ? c0 : c1;
                stack[sp * 64] = farc;
                sp++;
                cur = nearc;
            } else if (h0) {
                cur = c0;
            } else if (h1) {
                cur = c1;
            } else {
                if (sp == 0) return best;
                sp--;
                cur = stack[sp * 64];
            }
        }
        // leaf (cur < 0)
        {
            unsigned v = (unsigned)~cur;
            unsigned first = v >> 3, count = (v & 7u) + 1u;
            for (unsigned i = 0; i < count; i++) {
                unsigned slot = first + i;
                if ((int)slot == skip_slot) continue;
                F3 v0, v1, v2;
                float4 qa, qb, qc;
                load_verts(S, slot, v0, v1, v2, qa, qb, qc);
                if (COUNT) tri_tests++;
                float u, vv, t;
                if (!tri_mt(v0, v1, v2, ray, u, vv, t)) continue;
                float dist = candidate_distance(S, slot, v0, v1, v2, ray, u, vv, t);
                if (ANY) {
                    if (dist < limit) { best_dist = dist; return (int)slot; }
                } else if (best < 0 || dist < best_dist) {   // src/BVH.cpp:114
                    best = (int)slot;
                    best_dist = dist;
                    bound = dist * 1.00001f + S.prune_margin;
                }
            }
        }
        if (sp == 0) return best;
        sp--;
        cur = stack[sp * 64];
    }
}

// ---- textures, src/Texture.cpp:172-236 -------------------------------------------
// |x % w| of C's truncating `%` (what `x %= w; if (x < 0) x *= -1;` of src/Texture.cpp:176-180 leaves) equals |x| % w, and for a
// power-of-two w that is |x| & (w - 1): two operations instead of the ~35 of a signed division by a run-time value -- six of them ran
// per shading step (HDRI texel, its pdf, the CDF index), profiles/r04_shader_function_budget.txt.  Same integers for every input:
// |INT_MIN| wraps to INT_MIN in both forms (its low bits are zero, so the mask gives 0 = |INT_MIN % w|).
ERD int wrap_abs(int x, int w, bool pow2) {      // pow2: EVERY texture of the scene has power-of-two sides (DevScene::tex_pow2, wave-uniform)
    if (pow2) return (int)((x < 0 ? 0u - (unsigned)x : (unsigned)x) & (unsigned)(w - 1));
    x %= w;
    return x < 0 ? -x : x;
}
ERD F3 tex_coords(const DevScene& S, const DevTex& t, int x, int y) {
    x = wrap_abs(x, t.width, S.tex_pow2 != 0u);
    y = wrap_abs(y, t.height, S.tex_pow2 != 0u);
    const float* d = S.tex_pool + t.offset;
    F3 pixel = f3s(0);
    if (t.channels == 1) {
        pixel = f3s(d[y * t.width + x]);
    } else if (t.channels == 2) {
        pixel.x = d[t.channels * (y * t.width + x) + 0];
        pixel.y = d[t.channels * (y * t.width + x) + 1];
    } else if (t.channels >= 3) {
        pixel.x = d[t.channels * (y * t.width + x) + 0];
        pixel.y = d[t.channels * (y * t.width + x) + 1];
        pixel.z = d[t.channels * (y * t.width + x) + 2];
    }
    return pixel;
}
ERD F3 tex_uv(const DevScene& S, const DevTex& t, float u, float v) {
    return tex_coords(S, t, ermath::f2i(u * t.width), ermath::f2i(v * t.height));
}
ERD F3 tex_bilinear(const DevScene& S, const DevTex& t, float u, float v) {
    float x = u * t.width, y = v * t.height;
    float t1x = __builtin_floorf(x), t1y = __builtin_floorf(y);
    float t2x = t1x + 1, t2y = t1y + 1;
    float a = (x - t1x) / (t2x - t1x);
    float b = (y - t1y) / (t2y - t1y);
    F3 v1 = tex_coords(S, t, ermath::f2i(t1x), ermath::f2i(t1y));
    F3 v2 = tex_coords(S, t, ermath::f2i(t2x), ermath::f2i(t1y));
    F3 v3 = tex_coords(S, t, ermath::f2i(t1x), ermath::f2i(t2y));
    F3 v4 = tex_coords(S, t, ermath::f2i(t2x), ermath::f2i(t2y));
    return lerpv(lerpv(v1, v2, a), lerpv(v3, v4, a), b);
}
ERD F3 tex_filtered(const DevScene& S, const DevTex& t, float u, float v) {
    return t.filter == 1 ? tex_bilinear(S, t, u, v) : tex_uv(S, t, u, v);
}
// the fused texel of a material (DevFused): the statements of tex_coords / tex_bilinear on all five channels at once
ERD void fused_coords(const DevScene& S, const DevFused& t, int x, int y, F3& alb, F3& rm) {
    x = wrap_abs(x, t.width, S.tex_pow2 != 0u);
    y = wrap_abs(y, t.height, S.tex_pow2 != 0u);
    const float* d = S.tex_pool + t.offset + 5u * (uint32_t)(y * t.width + x);
    alb = f3(d[0], d[1], d[2]);
    rm = f3(d[3], d[4], 0.0f);
}
ERD void fused_fetch(const DevScene& S, const DevFused& t, float u, float v, F3& alb, F3& rm) {
    if (t.filter == 1) {
        float x = u * t.width, y = v * t.height;
        float t1x = __builtin_floorf(x), t1y = __builtin_floorf(y);
        float t2x = t1x + 1, t2y = t1y + 1;
        float a = (x - t1x) / (t2x - t1x);
        float b = (y - t1y) / (t2y - t1y);
        F3 a1, a2, a3, a4, r1, r2, r3, r4;
        fused_coords(S, t, ermath::f2i(t1x), ermath::f2i(t1y), a1, r1);
        fused_coords(S, t, ermath::f2i(t2x), ermath::f2i(t1y), a2, r2);
        fused_coords(S, t, ermath::f2i(t1x), ermath::f2i(t2y), a3, r3);
        fused_coords(S, t, ermath::f2i(t2x), ermath::f2i(t2y), a4, r4);
        alb = lerpv(lerpv(a1, a2, a), lerpv(a3, a4, a), b);
        rm = lerpv(lerpv(r1, r2, a), lerpv(r3, r4, a), b);
    } else {
        fused_coords(S, t, ermath::f2i(u * t.width), ermath::f2i(v * t.height), alb, rm);
    }
}
ERD void spherical_mapping(F3 point, float& u, float& v) {   // src/Texture.cpp:239-251 (origin 0, radius 1)
    F3 p = (point - f3s(0)) / 1.0f;
    float theta = ermath::er_acos(-p.y);
    float phi = ermath::er_atan2(-p.z, p.x) + PIF;
    u = phi / (2 * PIF);
    v = theta / PIF;
    limitUV(u, v);
}
ERD void inverse_transform_uv(const DevTex& t, float u, float v, float& nu, float& nv) {   // src/Texture.cpp:267-278
    int x = ermath::f2i(u * t.width);
    int y = ermath::f2i(v * t.height);
    nu = (float)x / (float)t.width;
    nv = (float)y / (float)t.height;
    limitUV(nu, nv);
}
ERD F3 reverse_spherical_mapping(float u, float v) {   // src/Texture.cpp:280-292
    float phi = u * 2 * PIF;
    float theta = v * PIF;
    float px = ermath::er_cos(phi - PIF);
    float py = -ermath::er_cos(theta);
    float pz = -ermath::er_sin(phi - PIF);
    float a = __builtin_sqrtf(1 - py * py);
    return f3(a * px, py, a * pz);
}

// ---- HDRI, src/HDRI.cpp:85-117 ---------------------------------------------------
ERD int hdri_binary_search(const float* arr, float value, int length) {
    int from = 0;
    int to = length - 1;
    while (to - from > 0) {
        int m = from + (to - from) / 2;
        if (m >= length || m < 0) return 0;
        float am = arr[m];
        if (value == am) return m;
        if (value < am) to = m - 1;
        if (value > am) from = m + 1;
    }
    return to;
}
ERD float hdri_pdf(const DevScene& S, int x, int y) {
    F3 dv = tex_coords(S, S.hdri_tex, x, y);
    float theta = (((float)y / (float)S.hdri_tex.height)) * PIF;
    return ((dv.x + dv.y + dv.z) / S.hdri_radiance_sum) * S.hdri_tex.width * S.hdri_tex.height / (2.0f * PIF * ermath::er_sin(theta));
}

// ---- Disney BRDF, src/Disney.cpp:34-230 ---------------------------------------------
struct HitData {
    float metallic, roughness, clearcoatGloss, clearcoat, anisotropic, transmission, specular,
        specularTint, sheenTint, subsurface, sheen, opacity;
    float gtr1_log;      // er_log(a * a) of the clearcoat lobe's alpha (a per-material constant: DevScene::mat_pre), NaN = compute it
    F3 emission, albedo;
    F3 position, normal, tangent, bitangent;
};

ERD float SchlickFresnel(float u) {
    float m = clampf(1.0f - u, 0.0f, 1.0f);
    float m2 = m * m;
    return m2 * m2 * m;
}
ERD float GTR1(float NDotH, float a, float log_a2) {      // log_a2 = er_log(a * a), or NaN: evaluated here
    if (a >= 1.0f) return (1.0f / PIF);
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    const float lg = log_a2 == log_a2 ? log_a2 : ermath::er_log(a2);
    return (a2 - 1.0f) / (PIF * lg * t);
}
ERD float GTR2_aniso(float NDotH, float HDotX, float HDotY, float ax, float ay) {
    float a = HDotX / ax;
    float b = HDotY / ay;
    float c = a * a + b * b + NDotH * NDotH;
    return 1.0f / (PIF * ax * ay * c * c);
}
ERD float SmithG_GGX(float NDotV, float alphaG) {
    float a = alphaG * alphaG;
    float b = NDotV * NDotV;
    return 1.0f / (NDotV + __builtin_sqrtf(a + b - a * b));
}
ERD float SmithG_GGX_aniso(float NDotV, float VDotX, float VDotY, float ax, float ay) {
    float a = VDotX * ax;
    float b = VDotY * ay;
    float c = NDotV;
    return 1.0f / (NDotV + __builtin_sqrtf(a * a + b * b + c * c));
}
ERD float DisneyPdf(const HitData& hd, F3 V, F3 N, F3 L) {   // src/Disney.cpp:97-133
    F3 H = normalized(L + V);
    F3 T = hd.tangent, B = hd.bitangent;
    float NDotH = __builtin_fabsf(dot(N, H));
    if (dot(N, L) <= 0.0f) return 1.0f;
    float clearcoatAlpha = lerpf(0.1f, 0.001f, hd.clearcoatGloss);
    float diffuseRatio = 0.5f * (1.0f - hd.metallic);
    float specularRatio = 1.0f - diffuseRatio;
    float aspect = __builtin_sqrtf(1.0f - hd.anisotropic * 0.9f);
    float ax = maxf(0.001f, hd.roughness / aspect);
    float ay = maxf(0.001f, hd.roughness * aspect);
    float pdfGTR2_aniso = GTR2_aniso(NDotH, dot(H, T), dot(H, B), ax, ay) * NDotH;
    float pdfGTR1 = GTR1(NDotH, clearcoatAlpha, hd.gtr1_log) * NDotH;
    float ratio = 1.0f / (1.0f + hd.clearcoat);
    float pdfSpec = lerpf(pdfGTR1, pdfGTR2_aniso, ratio) / (4.0f * __builtin_fabsf(dot(L, H)));
    float pdfDiff = __builtin_fabsf(dot(L, N)) * (1.0f / PIF);
    return diffuseRatio * pdfDiff + specularRatio * pdfSpec;
}
ERD F3 DisneySample(const HitData& hd, F3 V, F3 N, float r1, float r2, float r3) {   // src/Disney.cpp:136-158
    F3 T = hd.tangent, B = hd.bitangent;
    float diffuseRatio = 0.5f * (1.0f - hd.metallic);
    if (r3 < diffuseRatio) {
        // CosineSampleHemisphere, src/Sampling.h:30-40
        F3 H;
        float r = __builtin_sqrtf(r1);
        float phi = 2.0f * PIF * r2;
        H.x = r * ermath::er_cos(phi);
        H.y = r * ermath::er_sin(phi);
        H.z = __builtin_sqrtf(maxf(0.0f, 1.0f - H.x * H.x - H.y * H.y));
        return T * H.x + B * H.y + N * H.z;
    } else {
        // ImportanceSampleGGX, src/Sampling.h:42-53
        float a = maxf(0.001f, hd.roughness);
        float phi = r1 * PIF * 2;
        float cosTheta = __builtin_sqrtf((1.0f - r2) / (1.0f + (a * a - 1.0f) * r2));
        float sinTheta = clampf(__builtin_sqrtf(1.0f - (cosTheta * cosTheta)), 0.0f, 1.0f);
        float sinPhi = ermath::er_sin(phi);
        float cosPhi = ermath::er_cos(phi);
        F3 Hl = f3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
        F3 H = T * Hl.x + B * Hl.y + N * Hl.z;
        F3 v1 = -1 * V;
        return v1 - (2 * dot(v1, H)) * H;   // reflect, src/Vector.h:238-240
    }
}
ERD F3 DisneyEval(const HitData& hd, F3 V, F3 N, F3 L) {   // src/Disney.cpp:160-230
    F3 T = hd.tangent, B = hd.bitangent;
    F3 H = normalized(L + V);
    float NDotL = __builtin_fabsf(dot(N, L));
    float NDotV = __builtin_fabsf(dot(N, V));
    float NDotH = __builtin_fabsf(dot(N, H));
    float LDotH = __builtin_fabsf(dot(L, H));
    F3 brdf = f3s(0.0f);
    if (hd.transmission < 1.0f && dot(N, L) > 0.0f && dot(N, V) > 0.0f) {
        F3 Cdlin = hd.albedo;
        float Cdlum = 0.3f * Cdlin.x + 0.6f * Cdlin.y + 0.1f * Cdlin.z;
        F3 Ctint = Cdlum > 0.0f ? Cdlin / Cdlum : f3s(1.0f);
        F3 Cspec0 = lerpv(hd.specular * 0.08f * lerpv(f3s(1.0f), Ctint, hd.specularTint), Cdlin, hd.metallic);
        F3 Csheen = lerpv(f3s(1.0f), Ctint, hd.sheenTint);
        float FL = SchlickFresnel(NDotL);
        float FV = SchlickFresnel(NDotV);
        float Fd90 = 0.5f + 2.0f * LDotH * LDotH * hd.roughness;
        float Fd = lerpf(1.0f, Fd90, FL) * lerpf(1.0f, Fd90, FV);
        float Fss90 = LDotH * LDotH * hd.roughness;
        float Fss = lerpf(1.0f, Fss90, FL) * lerpf(1.0f, Fss90, FV);
        float ss = 1.25f * (Fss * (1.0f / (NDotL + NDotV) - 0.5f) + 0.5f);
        float aspect = __builtin_sqrtf(1.0f - hd.anisotropic * 0.9f);
        float ax = maxf(0.001f, hd.roughness / aspect);
        float ay = maxf(0.001f, hd.roughness * aspect);
        float Ds = GTR2_aniso(NDotH, dot(H, T), dot(H, B), ax, ay);
        float FH = SchlickFresnel(LDotH);
        F3 Fs = lerpv(Cspec0, f3s(1.0f), FH);
        float Gs = SmithG_GGX_aniso(NDotL, dot(L, T), dot(L, B), ax, ay);
        Gs *= SmithG_GGX_aniso(NDotV, dot(V, T), dot(V, B), ax, ay);
        F3 Fsheen = FH * hd.sheen * Csheen;
        float Dr = GTR1(NDotH, lerpf(0.1f, 0.001f, hd.clearcoatGloss), hd.gtr1_log);
        float Fr = lerpf(0.04f, 1.0f, FH);
        float Gr = SmithG_GGX(NDotL, 0.25f) * SmithG_GGX(NDotV, 0.25f);
        brdf = addf(((1.0f / PIF) * lerpf(Fd, ss, hd.subsurface) * Cdlin + Fsheen) * (1.0f - hd.metallic) + Gs * Fs * Ds,
                    0.25f * hd.clearcoat * Gr * Fr * Dr);
    }
    return brdf;
}

// ---- generateHitData, src/kernel.cpp:76-172 ----------------------------------------
// FUSE = false compiles the fused-texel path out: the streaming kernel has an instance of its own for scenes in which no material is
// fused (C2, C4), whose shading step is then the code it was before -- with the path compiled in, C2 ran 0.4 ... 0.6 % slower for a
// branch it never takes (register allocation of the whole kernel; profiles/r04_ab_fused_path_on_untextured_scenes.log).
template <bool COUNT, bool FUSE = true>
ERD void generate_hit_data(const DevScene& S, const ErMaterial& mat, const HitFull& hit, HitData& hd, unsigned& texels) {
    DevFused fu = {0, 0, 0, 0};
    if (FUSE && S.fused_any) fu = S.mat_fused[hit.material];
    const bool fused = FUSE && fu.width > 0;      // albedo, roughness and metallic of this material come from one fused texel
    F3 fu_rm = f3s(0);
    if (fused) { fused_fetch(S, fu, hit.tu, hit.tv, hd.albedo, fu_rm); if (COUNT) texels += 3; }
    else if (mat.albedo_tex < 0) hd.albedo = f3(mat.albedo.x, mat.albedo.y, mat.albedo.z);
    else { hd.albedo = tex_filtered(S, S.textures[mat.albedo_tex], hit.tu, hit.tv); if (COUNT) texels++; }
    if (mat.opacity_tex < 0) hd.opacity = mat.opacity;
    else { hd.opacity = tex_filtered(S, S.textures[mat.opacity_tex], hit.tu, hit.tv).x; if (COUNT) texels++; }
    if (mat.emission_tex < 0) hd.emission = f3(mat.emission.x, mat.emission.y, mat.emission.z);
    else { hd.emission = tex_filtered(S, S.textures[mat.emission_tex], hit.tu, hit.tv); if (COUNT) texels++; }
    if (fused) hd.roughness = fu_rm.x;
    else if (mat.roughness_tex < 0) hd.roughness = mat.roughness;
    else { hd.roughness = tex_filtered(S, S.textures[mat.roughness_tex], hit.tu, hit.tv).x; if (COUNT) texels++; }
    if (fused) hd.metallic = fu_rm.y;
    else if (mat.metallic_tex < 0) hd.metallic = mat.metallic;
    else { hd.metallic = tex_filtered(S, S.textures[mat.metallic_tex], hit.tu, hit.tv).x; if (COUNT) texels++; }
    if (mat.transmission_tex < 0) hd.transmission = mat.transmission;
    else { hd.transmission = tex_filtered(S, S.textures[mat.transmission_tex], hit.tu, hit.tv).x; if (COUNT) texels++; }
    if (mat.normal_tex < 0) {
        hd.normal = hit.normal;
    } else {
        F3 ncolor = tex_uv(S, S.textures[mat.normal_tex], hit.tu, hit.tv);
        if (COUNT) texels++;
        F3 localNormal = (ncolor * 2) - f3s(1.0f);
        hd.normal = normalized(localNormal.x * hit.tangent - localNormal.y * hit.bitangent + localNormal.z * hit.normal);
    }
    // roughness and metallic to the power 2.2 (src/kernel.cpp:152-153): of a constant channel once per material on the host, of a
    // textured one here
    const float4 pre = S.mat_pre[hit.material];
    // (... and of a texture the host already holds to that power: DevTex::filter == 2, er_api.cpp)
    if (fused) {
        if (fu.filter != 2) { hd.roughness = ermath::er_pow(hd.roughness, 2.2f); hd.metallic = ermath::er_pow(hd.metallic, 2.2f); }
    } else {
        if (mat.roughness_tex < 0) hd.roughness = pre.x;
        else if (S.textures[mat.roughness_tex].filter != 2) hd.roughness = ermath::er_pow(hd.roughness, 2.2f);
        if (mat.metallic_tex < 0) hd.metallic = pre.y;
        else if (S.textures[mat.metallic_tex].filter != 2) hd.metallic = ermath::er_pow(hd.metallic, 2.2f);
    }
    hd.gtr1_log = pre.w != 0.0f ? pre.z : __builtin_nanf("");
    hd.clearcoatGloss = mat.clearcoat_gloss;
    hd.clearcoat = mat.clearcoat;
    hd.anisotropic = mat.anisotropic;
    hd.specular = mat.specular;
    hd.specularTint = mat.specular_tint;
    hd.sheenTint = mat.sheen_tint;
    hd.subsurface = mat.subsurface;
    hd.sheen = mat.sheen;
    hd.tangent = hit.tangent;
    hd.bitangent = hit.bitangent;
    hd.position = hit.position;
}

}  // namespace erd
