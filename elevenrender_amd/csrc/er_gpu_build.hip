// er_gpu_build.hip -- device-side builder of the acceleration structure.
//
// The reference builds its tree on the host, one node at a time (reference src/BVH.cpp:132-415, 11.7 s for
// 1M triangles); the contract is only "same nearest hit", so the tree is free to differ.  This file builds the whole
// structure on the GPU and is the library's default builder since round 5 (er_api.cpp: scenes of at least
// ER_GPU_BUILD_MIN_TRIS triangles; ER_FLAG_HOST_BUILD keeps the host build of er_bvh.cpp, which is also what a declined or
// failed device build falls back to):
//
//   1. bounds      per-triangle padded box and lift bound (the same arithmetic as er_build_bvh), largest coordinate;
//   2. binary tree top-down binned SAH, the host builder's algorithm, one tree level per round (comment above k_sah_cen);
//   3. collapse    the SAH-optimal collapse into 8-wide compressed nodes (the dynamic programme and the breadth-first
//                  layout of er_collapse_bvh8, er_bvh.cpp): the DP bottom-up with a last-arrival rule, the emission one
//                  tree level per launch with prefix sums for the node and triangle positions, so the layout never
//                  depends on thread timing;
//   4. records     the binary tree's leaf references follow the new slot order; the 48-byte intersection and
//                  112-byte attribute records are written in that order straight from the scene arrays.
//
// Everything the kernels read is produced in place on the device; a few counters come back.  Measured on MI355X (round 5,
// profiles/r05_ab_device_sah_builder.log): 1M triangles 65-73 ms against 420-435 ms for the host build (16 threads) and 9 ms
// instead of 112 ms of upload; 10M triangles 325-340 ms against 5.0-5.2 s (+ 17 ms instead of 830 ms) -- and the SAME tree:
// 119 146 wide nodes and 20.67 node visits per ray on C2, 1 169 040 and 18.43 on C4, the same Msamples/s.  Round 1's builder
// here was a radix tree over Morton codes (Karras 2012): 5 % more node visits per ray, 2.5 % slower frames, and too deep for
// the traversal stacks on the 10 M-triangle scene, where it silently declined; a locally-ordered clustering (Meister and
// Bittner 2018) tried in round 5 made a worse wide tree still (profiles/r05_ab_builder_ploc.log).  Every image is identical
// bit for bit whichever builder made the tree (tests/test_gpu_build.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>     // (before rocprim: its texture iterator calls the host memset)
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "er_bvh.h"
#include "er_gpu_build.h"

namespace {

struct Box3 { float lo[3], hi[3]; };

__device__ __forceinline__ unsigned f2ord(float f) {   // order-preserving map float -> unsigned
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// pass 1: largest |coordinate| (the absolute part of the box padding, as er_build_bvh)
__global__ __launch_bounds__(256) void k_scene_bounds(const float* __restrict__ v, uint32_t n, unsigned* g /* [0] vmax bits */) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float vm = 0;
    if (i < n) {
        const float* p = v + (size_t)i * 9;
        for (int k = 0; k < 9; k++) vm = fmaxf(vm, fabsf(p[k]));
    }
    for (int off = 32; off >= 1; off >>= 1) vm = fmaxf(vm, __shfl_xor(vm, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&g[0], __float_as_uint(vm));     // vm >= 0: the bit pattern orders like the value
}

// pass 2: padded box, lift bound (er_build_bvh's arithmetic, reference src/Tri.h:106-112)
__global__ __launch_bounds__(256) void k_prims(const float* __restrict__ v, const float* __restrict__ nrm, uint32_t n, const unsigned* __restrict__ g,
                                                Box3* __restrict__ boxes, float* __restrict__ lift, unsigned* __restrict__ lift_max) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float my_lift = 0;
    if (i < n) {
        const float pad_abs = __uint_as_float(g[0]) * 1e-6f;
        const float* p = v + (size_t)i * 9;
        Box3 b;
        for (int a = 0; a < 3; a++) {
            float lo = fminf(fminf(p[a], p[3 + a]), p[6 + a]), hi = fmaxf(fmaxf(p[a], p[3 + a]), p[6 + a]);
            float m = fmaxf(fabsf(lo), fabsf(hi));
            float pad = fmaxf(m * 4e-7f + 1e-37f, pad_abs);
            b.lo[a] = lo - pad;
            b.hi[a] = hi + pad;
        }
        boxes[i] = b;
        const float* nn = nrm + (size_t)i * 9;
        double tl = 0;
        for (int j = 0; j < 3; j++) {
            double nx = nn[3 * j], ny = nn[3 * j + 1], nz = nn[3 * j + 2];
            double nl = sqrt(nx * nx + ny * ny + nz * nz);
            for (int k = 0; k < 3; k++) {
                if (k == j) continue;
                double dx = (double)p[3 * k] - p[3 * j], dy = (double)p[3 * k + 1] - p[3 * j + 1], dz = (double)p[3 * k + 2] - p[3 * j + 2];
                double l = fabs(dx * nx + dy * ny + dz * nz) * nl;
                if (l > tl) tl = l;
            }
        }
        lift[i] = (float)(tl * 1.01) + 1e-30f;
        my_lift = (float)(tl * 1.01);
    }
    for (int off = 32; off >= 1; off >>= 1) my_lift = fmaxf(my_lift, __shfl_xor(my_lift, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(lift_max, __float_as_uint(my_lift));
}

// ---------------------------------------------------------------------------------------------------------
// Round 5: the host builder's own algorithm on the device -- top-down binned SAH (er_bvh.cpp: 16 bins per axis over the
// centroid bounds, cost = area x count on either side, leaves of <= ER_BVH_LEAF_MAX triangles), one tree LEVEL per round:
//   k_sah_cb        centroid bounds of every active node (atomic min / max on order-preserving integers);
//   k_sah_bin       every triangle of an active node into its bin on each axis: count + box (per-wave LDS bins when the
//                   whole wave works on one node -- the top levels --, global atomics otherwise);
//   k_sah_split     one thread per active node sweeps the 3 x 15 candidate planes exactly as Builder::split does and
//                   writes the node: child boxes, the plane, the size of the left side;
//   (scan)          which children are inner nodes -> their indices in the next round and their node ids (breadth-first);
//   k_sah_children  child references (inner id or leaf range), parent links, the next round's node list;
//   k_sah_flags + (scan) + k_sah_scatter   stable partition of every active node's triangles by its plane.
// The triangles of a subtree are consecutive slots by construction, two-triangle leaves come out directly (no pairing
// pass), the root is node 0, the depth is that of the host build (26 at 1 M triangles, ~30 at 10 M) and bounded by the same
// guard, and the tree is of the host build's quality because it is the host build's tree up to float rounding of the bins.
// The radix tree over Morton codes of round 1 and a locally-ordered clustering (profiles/r05_ab_builder_ploc.log, r05_ab_device_sah_builder.log)
// make trees that a ray visits 5 % / 20 % more nodes of.
// ---------------------------------------------------------------------------------------------------------
#ifndef SAH_BINS
#define SAH_BINS 16
#endif
#define SAH_BIN_WORDS 7                                   // count, lo[3], hi[3] (order-preserving integers)
#define SAH_NODE_WORDS (3 * SAH_BINS * SAH_BIN_WORDS)     // 336 words = 1 344 bytes of bins per active node

struct SahAct { uint32_t node, lo, hi, depth; float blo[3], bhi[3]; };      // an active node: its id, triangle range, depth and box
struct SahSplit { int axis, bin; uint32_t left; uint32_t in0, in1; };        // the plane (axis < 0: by position), the left side's size, which children are inner

__device__ __forceinline__ int sah_bin_of(float c, float lo, float ext) {
    const float scale = (float)SAH_BINS / ext;
    int b = (int)((c - lo) * scale);
    return b < 0 ? 0 : (b >= SAH_BINS ? SAH_BINS - 1 : b);
}

__global__ __launch_bounds__(256) void k_sah_cen(const float* __restrict__ v, uint32_t n, float* __restrict__ cen) {      // the host builder's centroid
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* p = v + (size_t)i * 9;
    for (int a = 0; a < 3; a++) cen[(size_t)i * 3 + a] = (p[a] + p[3 + a] + p[6 + a]) * (1.0f / 3.0f);
}

__global__ __launch_bounds__(256) void k_sah_init(uint32_t n, uint32_t* __restrict__ idx, int* __restrict__ node_of) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p < n) { idx[p] = p; node_of[p] = 0; }
}

__global__ __launch_bounds__(256) void k_sah_clear(unsigned* __restrict__ cbv, uint32_t n_act, unsigned* __restrict__ bins, uint32_t bin_nodes) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)n_act * 6) cbv[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
    if (i < (size_t)bin_nodes * SAH_NODE_WORDS) { const uint32_t w = (uint32_t)(i % SAH_BIN_WORDS); bins[i] = w == 0 ? 0u : (w < 4 ? 0xffffffffu : 0u); }
}

__global__ __launch_bounds__(256) void k_sah_cb(uint32_t n, const uint32_t* __restrict__ idx, const int* __restrict__ node_of, const float* __restrict__ cen,
                                                 unsigned* __restrict__ cbv) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    const int a = p < n ? node_of[p] : -1;
    float c[3] = {0, 0, 0};
    if (a >= 0) { const float* q = cen + (size_t)idx[p] * 3; c[0] = q[0]; c[1] = q[1]; c[2] = q[2]; }
    const int a0 = __builtin_amdgcn_readfirstlane(a);
    if (__all(a == a0)) {      // the whole wave in one node (the top levels): one set of atomics per wave
        if (a0 < 0) return;
        float lo[3] = {c[0], c[1], c[2]}, hi[3] = {c[0], c[1], c[2]};
        for (int off = 32; off >= 1; off >>= 1)
            for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64)); }
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 3; k++) { atomicMin(&cbv[(size_t)a0 * 6 + k], f2ord(lo[k])); atomicMax(&cbv[(size_t)a0 * 6 + 3 + k], f2ord(hi[k])); }
        return;
    }
    if (a < 0) return;
    for (int k = 0; k < 3; k++) { atomicMin(&cbv[(size_t)a * 6 + k], f2ord(c[k])); atomicMax(&cbv[(size_t)a * 6 + 3 + k], f2ord(c[k])); }
}

// bins of the active nodes [a_lo, a_hi) (a round whose nodes outnumber the bin buffer runs in several such chunks)
__global__ __launch_bounds__(256) void k_sah_bin(uint32_t n, const uint32_t* __restrict__ idx, const int* __restrict__ node_of, const float* __restrict__ cen,
                                                  const Box3* __restrict__ boxes, const unsigned* __restrict__ cbv, unsigned* __restrict__ bins, int a_lo, int a_hi) {
    __shared__ unsigned s_bins[4][SAH_NODE_WORDS];
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    int a = p < n ? node_of[p] : -1;
    if (a < a_lo || a >= a_hi) a = -1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a0 = __builtin_amdgcn_readfirstlane(a);
    const bool uni = __all(a == a0);
    if (uni && a0 < 0) return;
    int b[3] = {-1, -1, -1};
    Box3 bx;
    if (a >= 0) {
        const uint32_t id = idx[p];
        bx = boxes[id];
        const float* q = cen + (size_t)id * 3;
        for (int k = 0; k < 3; k++) {
            const float lo = ord2f(cbv[(size_t)a * 6 + k]), hi = ord2f(cbv[(size_t)a * 6 + 3 + k]);
            const float ext = hi - lo;
            if (ext > 0) b[k] = sah_bin_of(q[k], lo, ext);
        }
    }
    if (uni) {
        unsigned* sb = s_bins[wave];
        for (int i = lane; i < SAH_NODE_WORDS; i += 64) { const int w = i % SAH_BIN_WORDS; sb[i] = w == 0 ? 0u : (w < 4 ? 0xffffffffu : 0u); }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int k = 0; k < 3; k++) {
            if (b[k] < 0) continue;
            unsigned* t = sb + (k * SAH_BINS + b[k]) * SAH_BIN_WORDS;
            atomicAdd(&t[0], 1u);
            for (int m = 0; m < 3; m++) { atomicMin(&t[1 + m], f2ord(bx.lo[m])); atomicMax(&t[4 + m], f2ord(bx.hi[m])); }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        unsigned* g = bins + (size_t)(a0 - a_lo) * SAH_NODE_WORDS;
        for (int i = lane; i < 3 * SAH_BINS; i += 64) {
            const unsigned* t = sb + i * SAH_BIN_WORDS;
            if (t[0] == 0u) continue;
            unsigned* o = g + i * SAH_BIN_WORDS;
            atomicAdd(&o[0], t[0]);
            for (int m = 0; m < 3; m++) { atomicMin(&o[1 + m], t[1 + m]); atomicMax(&o[4 + m], t[4 + m]); }
        }
        return;
    }
    if (a < 0) return;
    unsigned* g = bins + (size_t)(a - a_lo) * SAH_NODE_WORDS;
    for (int k = 0; k < 3; k++) {
        if (b[k] < 0) continue;
        unsigned* o = g + (k * SAH_BINS + b[k]) * SAH_BIN_WORDS;
        atomicAdd(&o[0], 1u);
        for (int m = 0; m < 3; m++) { atomicMin(&o[1 + m], f2ord(bx.lo[m])); atomicMax(&o[4 + m], f2ord(bx.hi[m])); }
    }
}

__device__ __forceinline__ float sah_area(const float* lo, const float* hi) {      // Box::area of er_bvh.cpp
    const float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    if (!(x >= 0) || !(y >= 0) || !(z >= 0)) return 0;
    return 2 * (x * y + x * z + y * z);
}
__device__ __forceinline__ int sah_levels_needed(uint32_t n) {      // Builder::levels_needed
    const uint32_t v = (n + ER_BVH_LEAF_MAX - 1) / ER_BVH_LEAF_MAX;
    int r = 0;
    while ((1u << r) < v) r++;
    return r;
}

// Builder::split of er_bvh.cpp for the active nodes [a_lo, a_hi), from their bins
__global__ __launch_bounds__(64) void k_sah_split(const SahAct* __restrict__ act, const unsigned* __restrict__ cbv, const unsigned* __restrict__ bins, int a_lo, int a_hi,
                                                   SahSplit* __restrict__ split, ErNode* __restrict__ nodes) {
    const int a = a_lo + (int)(blockIdx.x * 64 + threadIdx.x);
    if (a >= a_hi) return;
    const SahAct A = act[a];
    const uint32_t n = A.hi - A.lo;
    const unsigned* g = bins + (size_t)(a - a_lo) * SAH_NODE_WORDS;
    const bool force_median = (int)A.depth + 1 + sah_levels_needed(n) > ER_BVH_MAX_DEPTH - 1;
    int best_axis = -1, best_bin = -1;
    float best_cost = INFINITY;
    for (int axis = 0; axis < 3 && !force_median; axis++) {
        const float ext = ord2f(cbv[(size_t)a * 6 + 3 + axis]) - ord2f(cbv[(size_t)a * 6 + axis]);
        if (!(ext > 0)) continue;
        const unsigned* t = g + axis * SAH_BINS * SAH_BIN_WORDS;
        float ra[SAH_BINS];
        uint32_t rc[SAH_BINS];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t c = 0;
        for (int b = SAH_BINS - 1; b >= 1; b--) {
            const unsigned* q = t + b * SAH_BIN_WORDS;
            if (q[0]) for (int m = 0; m < 3; m++) { lo[m] = fminf(lo[m], ord2f(q[1 + m])); hi[m] = fmaxf(hi[m], ord2f(q[4 + m])); }
            c += q[0];
            ra[b] = sah_area(lo, hi);
            rc[b] = c;
        }
        for (int m = 0; m < 3; m++) { lo[m] = INFINITY; hi[m] = -INFINITY; }
        c = 0;
        for (int b = 0; b < SAH_BINS - 1; b++) {
            const unsigned* q = t + b * SAH_BIN_WORDS;
            if (q[0]) for (int m = 0; m < 3; m++) { lo[m] = fminf(lo[m], ord2f(q[1 + m])); hi[m] = fmaxf(hi[m], ord2f(q[4 + m])); }
            c += q[0];
            if (c == 0 || rc[b + 1] == 0) continue;
            const float cost = sah_area(lo, hi) * (float)c + ra[b + 1] * (float)rc[b + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
        }
    }
    SahSplit S;
    S.axis = best_axis; S.bin = best_bin;
    ErNode nd;
    if (best_axis >= 0) {
        const unsigned* t = g + best_axis * SAH_BINS * SAH_BIN_WORDS;
        float l0[3] = {INFINITY, INFINITY, INFINITY}, h0[3] = {-INFINITY, -INFINITY, -INFINITY}, l1[3] = {INFINITY, INFINITY, INFINITY}, h1[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t c = 0;
        for (int b = 0; b < SAH_BINS; b++) {
            const unsigned* q = t + b * SAH_BIN_WORDS;
            if (!q[0]) continue;
            if (b <= best_bin) { c += q[0]; for (int m = 0; m < 3; m++) { l0[m] = fminf(l0[m], ord2f(q[1 + m])); h0[m] = fmaxf(h0[m], ord2f(q[4 + m])); } }
            else for (int m = 0; m < 3; m++) { l1[m] = fminf(l1[m], ord2f(q[1 + m])); h1[m] = fmaxf(h1[m], ord2f(q[4 + m])); }
        }
        S.left = c;
        for (int m = 0; m < 3; m++) { nd.lo0[m] = l0[m]; nd.hi0[m] = h0[m]; nd.lo1[m] = l1[m]; nd.hi1[m] = h1[m]; }
    } else {
        // every centroid in one place, or the depth guard: the two halves of the range, each under the node's whole box (rare)
        S.left = n / 2;
        for (int m = 0; m < 3; m++) { nd.lo0[m] = A.blo[m]; nd.hi0[m] = A.bhi[m]; nd.lo1[m] = A.blo[m]; nd.hi1[m] = A.bhi[m]; }
    }
    S.in0 = S.left > ER_BVH_LEAF_MAX ? 1u : 0u;
    S.in1 = n - S.left > ER_BVH_LEAF_MAX ? 1u : 0u;
    nd.c0 = 0; nd.c1 = 0; nd.pad[0] = 0; nd.pad[1] = 0;
    split[a] = S;
    nodes[A.node] = nd;
}

__global__ __launch_bounds__(256) void k_sah_count(const SahSplit* __restrict__ split, uint32_t n_act, uint32_t* __restrict__ kids) {
    const uint32_t a = blockIdx.x * 256 + threadIdx.x;
    if (a < n_act) kids[a] = split[a].in0 + split[a].in1;
}

__global__ __launch_bounds__(256) void k_sah_children(const SahAct* __restrict__ act, const SahSplit* __restrict__ split, const uint32_t* __restrict__ kid_off, uint32_t n_act,
                                                       uint32_t node_base, ErNode* __restrict__ nodes, int* __restrict__ parent_inner, SahAct* __restrict__ next,
                                                       int2* __restrict__ child_act, unsigned* __restrict__ totals /* [0] next actives, [1] leaves */) {
    const uint32_t a = blockIdx.x * 256 + threadIdx.x;
    unsigned leaves = 0;
    if (a < n_act) {
        const SahAct A = act[a];
        const SahSplit S = split[a];
        ErNode* nd = nodes + A.node;
        const uint32_t off = kid_off[a], cnt = A.hi - A.lo;
        int ca[2] = {-1, -1};
        for (int k = 0; k < 2; k++) {
            const uint32_t lo = k == 0 ? A.lo : A.lo + S.left, hi = k == 0 ? A.lo + S.left : A.hi;
            const bool inner = k == 0 ? S.in0 != 0 : S.in1 != 0;
            int ref;
            if (inner) {
                const uint32_t na = off + (k == 1 ? S.in0 : 0u);
                const uint32_t id = node_base + na;
                ref = (int)id;
                parent_inner[id] = (int)A.node * 2 + k;
                SahAct C;
                C.node = id; C.lo = lo; C.hi = hi; C.depth = A.depth + 1;
                for (int m = 0; m < 3; m++) { C.blo[m] = k == 0 ? nd->lo0[m] : nd->lo1[m]; C.bhi[m] = k == 0 ? nd->hi0[m] : nd->hi1[m]; }
                next[na] = C;
                ca[k] = (int)na;
            } else {
                ref = ~(int)((lo << 3) | (hi - lo - 1u));
                leaves++;
            }
            if (k == 0) nd->c0 = ref; else nd->c1 = ref;
        }
        child_act[a] = make_int2(ca[0], ca[1]);
        if (a == n_act - 1) totals[0] = off + S.in0 + S.in1;
        (void)cnt;
    }
    for (int off = 32; off >= 1; off >>= 1) leaves += __shfl_xor(leaves, off, 64);
    if ((threadIdx.x & 63) == 0 && leaves) atomicAdd(&totals[1], leaves);
}

__global__ __launch_bounds__(256) void k_sah_flags(uint32_t n, const uint32_t* __restrict__ idx, const int* __restrict__ node_of, const float* __restrict__ cen,
                                                    const SahAct* __restrict__ act, const SahSplit* __restrict__ split, const unsigned* __restrict__ cbv, uint32_t* __restrict__ flag) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int a = node_of[p];
    uint32_t f = 0;
    if (a >= 0) {
        const SahSplit S = split[a];
        if (S.axis < 0) f = p < act[a].lo + S.left ? 1u : 0u;
        else {
            const float lo = ord2f(cbv[(size_t)a * 6 + S.axis]), hi = ord2f(cbv[(size_t)a * 6 + 3 + S.axis]);
            f = sah_bin_of(cen[(size_t)idx[p] * 3 + S.axis], lo, hi - lo) <= S.bin ? 1u : 0u;
        }
    }
    flag[p] = f;
}

__global__ __launch_bounds__(256) void k_sah_scatter(uint32_t n, const uint32_t* __restrict__ idx, const int* __restrict__ node_of, const uint32_t* __restrict__ flag,
                                                      const uint32_t* __restrict__ scan, const SahAct* __restrict__ act, const SahSplit* __restrict__ split,
                                                      const int2* __restrict__ child_act, uint32_t* __restrict__ idx2, int* __restrict__ node_of2) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int a = node_of[p];
    if (a < 0) { idx2[p] = idx[p]; node_of2[p] = -1; return; }
    const uint32_t lo = act[a].lo, left = split[a].left;
    const uint32_t before = scan[p] - scan[lo];      // triangles of this node that go left and stand before p
    const bool l = flag[p] != 0;
    const uint32_t q = l ? lo + before : lo + left + ((p - lo) - before);
    idx2[q] = idx[p];
    const int2 ca = child_act[a];
    node_of2[q] = l ? ca.x : ca.y;
}

// ---------------------------------------------------------------------------------------------------------
// Wide-node collapse on the device: the same SAH-optimal dynamic programme and the same breadth-first layout
// as er_collapse_bvh8 (er_bvh.cpp; Ylitie, Karras, Laine 2017, section 4.1), level by level, with prefix sums
// instead of a queue so that the layout does not depend on thread timing.
// ---------------------------------------------------------------------------------------------------------
#ifndef ER_C_PRIM
#define ER_C_PRIM 0.3f
#endif
struct DpD {
    float C[8];
    unsigned char dec[8];
    uint32_t first, count;
};
__device__ __forceinline__ float area_d(const float* lo, const float* hi) {
    float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    return 2.0f * (x * y + x * z + y * z);
}
__device__ __forceinline__ uint32_t leaf_first_d(int ref) { return ((uint32_t)~ref) >> 3; }
__device__ __forceinline__ uint32_t leaf_count_d(int ref) { return (((uint32_t)~ref) & 7u) + 1u; }
__device__ __forceinline__ float child_cost(const DpD* dp, int ref, float area, int i) {
    if (ref < 0) return area * (float)leaf_count_d(ref) * ER_C_PRIM;
    return dp[ref].C[i];
}

// bottom-up: a node is evaluated by the last of its inner children to finish (or at once if it has none)
__global__ __launch_bounds__(256) void k_dp(const ErNode* __restrict__ nodes, uint32_t n_inner,
                                             const int* __restrict__ parent_inner, DpD* dp, unsigned* arrived) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_inner) return;
    if (nodes[i].c0 >= 0 || nodes[i].c1 >= 0) return;      // has inner children: one of them gets here
    while (true) {
        const ErNode nd = nodes[i];
        DpD D;
        const float a0 = area_d(nd.lo0, nd.hi0), a1 = area_d(nd.lo1, nd.hi1);
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) { lo[a] = fminf(nd.lo0[a], nd.lo1[a]); hi[a] = fmaxf(nd.hi0[a], nd.hi1[a]); }
        const float an = area_d(lo, hi);
        const uint32_t f0 = nd.c0 < 0 ? leaf_first_d(nd.c0) : dp[nd.c0].first, k0 = nd.c0 < 0 ? leaf_count_d(nd.c0) : dp[nd.c0].count;
        const uint32_t f1 = nd.c1 < 0 ? leaf_first_d(nd.c1) : dp[nd.c1].first, k1 = nd.c1 < 0 ? leaf_count_d(nd.c1) : dp[nd.c1].count;
        D.first = f0 < f1 ? f0 : f1;
        D.count = k0 + k1;
        const float c_leaf = D.count <= ER_BVH_LEAF_MAX ? an * (float)D.count * ER_C_PRIM : INFINITY;
        float best = INFINITY;
        int bk = 1;
        for (int k = 1; k <= 7; k++) {
            float c = child_cost(dp, nd.c0, a0, k) + child_cost(dp, nd.c1, a1, 8 - k);
            if (c < best) { best = c; bk = k; }
        }
        const float c_int = an * 1.0f + best;
        D.C[0] = 0; D.dec[0] = 0;
        if (c_leaf <= c_int) { D.C[1] = c_leaf; D.dec[1] = 0; } else { D.C[1] = c_int; D.dec[1] = (unsigned char)bk; }
        for (int q = 2; q <= 7; q++) {
            float bd = INFINITY;
            int kk = 0;
            for (int k = 1; k < q; k++) {
                float c = child_cost(dp, nd.c0, a0, k) + child_cost(dp, nd.c1, a1, q - k);
                if (c < bd) { bd = c; kk = k; }
            }
            if (bd < D.C[q - 1]) { D.C[q] = bd; D.dec[q] = (unsigned char)kk; } else { D.C[q] = D.C[q - 1]; D.dec[q] = 0; }
        }
        dp[i] = D;
        if (i == 0) return;
        __threadfence();
        const int p = parent_inner[i] >> 1;
        const unsigned need = (nodes[p].c0 >= 0 ? 1u : 0u) + (nodes[p].c1 >= 0 ? 1u : 0u);
        if (atomicAdd(&arrived[p], 1u) + 1u < need) return;
        __threadfence();
        i = (uint32_t)p;
    }
}

struct ChildD { int ref; bool as_leaf; float lo[3], hi[3]; };

// the (at most eight) children of the wide node rooted at binary node n2, in er_collapse_bvh8's order
__device__ int collect_children(const ErNode* __restrict__ nodes, const DpD* __restrict__ dp, int n2, ChildD* ch) {
    struct Item { int ref; int i; float lo[3], hi[3]; };
    Item st[9];
    int sp = 0, nc = 0;
    const ErNode nd = nodes[n2];
    const DpD& D0 = dp[n2];
    const int k0 = D0.dec[1] != 0 ? D0.dec[1] : 4;
    // (children are collected first under child 0, then under child 1: push child 1 first)
    st[sp].ref = nd.c1; st[sp].i = 8 - k0;
    for (int a = 0; a < 3; a++) { st[sp].lo[a] = nd.lo1[a]; st[sp].hi[a] = nd.hi1[a]; }
    sp++;
    st[sp].ref = nd.c0; st[sp].i = k0;
    for (int a = 0; a < 3; a++) { st[sp].lo[a] = nd.lo0[a]; st[sp].hi[a] = nd.hi0[a]; }
    sp++;
    while (sp > 0) {
        const Item it = st[--sp];
        ChildD c;
        c.ref = it.ref; c.as_leaf = false;
        for (int a = 0; a < 3; a++) { c.lo[a] = it.lo[a]; c.hi[a] = it.hi[a]; }
        if (it.ref < 0) { c.as_leaf = true; ch[nc++] = c; continue; }
        const DpD& D = dp[it.ref];
        int i = it.i;
        while (i >= 2 && D.dec[i] == 0) i--;
        if (i == 1) { c.as_leaf = D.dec[1] == 0; ch[nc++] = c; continue; }
        const ErNode m = nodes[it.ref];
        const int k = D.dec[i];
        st[sp].ref = m.c1; st[sp].i = i - k;
        for (int a = 0; a < 3; a++) { st[sp].lo[a] = m.lo1[a]; st[sp].hi[a] = m.hi1[a]; }
        sp++;
        st[sp].ref = m.c0; st[sp].i = k;
        for (int a = 0; a < 3; a++) { st[sp].lo[a] = m.lo0[a]; st[sp].hi[a] = m.hi0[a]; }
        sp++;
    }
    return nc;
}

struct WorkD { uint32_t n8; int n2; };

__global__ __launch_bounds__(128) void k_wide_count(const ErNode* __restrict__ nodes, const DpD* __restrict__ dp, const WorkD* __restrict__ work, uint32_t m,
                                                     uint32_t* __restrict__ n_inner_out, uint32_t* __restrict__ n_tri_out) {
    const uint32_t w = blockIdx.x * 128 + threadIdx.x;
    if (w >= m) return;
    ChildD ch[8];
    const int nc = collect_children(nodes, dp, work[w].n2, ch);
    uint32_t ci = 0, ti = 0;
    for (int k = 0; k < nc; k++) {
        if (!ch[k].as_leaf) ci++;
        else ti += ch[k].ref < 0 ? leaf_count_d(ch[k].ref) : dp[ch[k].ref].count;
    }
    n_inner_out[w] = ci;
    n_tri_out[w] = ti;
}

__global__ __launch_bounds__(128) void k_wide_emit(const ErNode* __restrict__ nodes, const DpD* __restrict__ dp, const WorkD* __restrict__ work, uint32_t m,
                                                    const uint32_t* __restrict__ inner_off, const uint32_t* __restrict__ tri_off_in, uint32_t node_base,
                                                    uint32_t tri_base0, WorkD* __restrict__ next, ErNode8* __restrict__ nodes8, uint32_t* __restrict__ new_order) {
    const uint32_t w = blockIdx.x * 128 + threadIdx.x;
    if (w >= m) return;
    ChildD ch[8];
    const int nc = collect_children(nodes, dp, work[w].n2, ch);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < nc; k++)
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], ch[k].lo[a]); hi[a] = fmaxf(hi[a], ch[k].hi[a]); }
    // slot assignment: greedy on score = (centroid - centre) . (+-1,+-1,+-1), best score first (stable)
    float score[64];
    unsigned char order[64];
    int ns = 0;
    for (int k = 0; k < nc; k++)
        for (int s8 = 0; s8 < 8; s8++) {
            float v = 0;
            for (int a = 0; a < 3; a++) {
                float rel = 0.5f * (ch[k].lo[a] + ch[k].hi[a]) - 0.5f * (lo[a] + hi[a]);
                v += ((s8 >> a) & 1) ? rel : -rel;
            }
            // stable insertion by descending score
            int pos = ns;
            while (pos > 0 && score[pos - 1] < v) { score[pos] = score[pos - 1]; order[pos] = order[pos - 1]; pos--; }
            score[pos] = v;
            order[pos] = (unsigned char)(k * 8 + s8);
            ns++;
        }
    int slot_of[8], child_in[8];
    for (int k = 0; k < 8; k++) { slot_of[k] = -1; child_in[k] = -1; }
    for (int q = 0; q < ns; q++) {
        const int c = order[q] >> 3, s8 = order[q] & 7;
        if (slot_of[c] < 0 && child_in[s8] < 0) { slot_of[c] = s8; child_in[s8] = c; }
    }
    ErNode8 nd;
    memset(&nd, 0, sizeof(nd));
    float scale[3];
    for (int a = 0; a < 3; a++) {
        nd.p[a] = lo[a];
        const float ext = hi[a] - lo[a];
        int e = 0;
        if (ext > 0) (void)frexpf(ext / 255.0f, &e); else e = -126;
        if (e < -126) e = -126;
        while (lo[a] + 255.0f * ldexpf(1.0f, e) < hi[a]) e++;
        nd.e[a] = (uint8_t)(e + 127);
        scale[a] = ldexpf(1.0f, e);
    }
    const uint32_t child_base = node_base + inner_off[w];
    uint32_t tri_base = tri_base0 + tri_off_in[w];
    nd.child_base = child_base;
    nd.tri_base = tri_base;
    uint32_t n_in = 0;
    for (int s8 = 0; s8 < 8; s8++) {
        const int ci = child_in[s8];
        if (ci < 0) continue;
        const ChildD& c = ch[ci];
        for (int a = 0; a < 3; a++) {
            float fl = floorf((c.lo[a] - nd.p[a]) / scale[a]);
            float fh = ceilf((c.hi[a] - nd.p[a]) / scale[a]);
            int ql = (int)fminf(255.0f, fmaxf(0.0f, fl));
            int qh = (int)fminf(255.0f, fmaxf(0.0f, fh));
            while (ql > 0 && nd.p[a] + (float)ql * scale[a] > c.lo[a]) ql--;
            while (qh < 255 && nd.p[a] + (float)qh * scale[a] < c.hi[a]) qh++;
            nd.qlo[a][s8] = (uint8_t)ql;
            nd.qhi[a][s8] = (uint8_t)qh;
        }
        if (!c.as_leaf) {
            nd.imask |= (uint8_t)(1u << s8);
            next[inner_off[w] + n_in] = WorkD{child_base + n_in, c.ref};
            n_in++;
        } else {
            const uint32_t first = c.ref < 0 ? leaf_first_d(c.ref) : dp[c.ref].first;
            const uint32_t count = c.ref < 0 ? leaf_count_d(c.ref) : dp[c.ref].count;
            nd.tri_present |= ((1u << count) - 1u) << (2 * s8);
            for (uint32_t i = 0; i < count; i++) new_order[tri_base + i] = first + i;
            tri_base += count;
        }
    }
    nodes8[work[w].n8] = nd;
}

// old slot -> new slot, then every leaf reference of the binary tree follows its triangles
__global__ __launch_bounds__(256) void k_inverse(const uint32_t* __restrict__ new_order, uint32_t n, uint32_t* __restrict__ inv) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) inv[new_order[k]] = k;
}
__global__ __launch_bounds__(256) void k_fix_leaves(ErNode* nodes, uint32_t n_inner, const uint32_t* __restrict__ inv) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_inner) return;
    int c[2] = {nodes[i].c0, nodes[i].c1};
    for (int k = 0; k < 2; k++)
        if (c[k] < 0) c[k] = ~(int)((inv[leaf_first_d(c[k])] << 3) | (leaf_count_d(c[k]) - 1u));
    nodes[i].c0 = c[0];
    nodes[i].c1 = c[1];
}

// the records the kernels read, written in the final slot order straight from the scene arrays
__global__ __launch_bounds__(256) void k_records(const uint32_t* __restrict__ new_order, const uint32_t* __restrict__ sorted_ids, uint32_t n,
                                                  const float* __restrict__ v, const float* __restrict__ nrm, const float* __restrict__ tan,
                                                  const float* __restrict__ uv, const float* __restrict__ sign, const int* __restrict__ mat,
                                                  const float* __restrict__ lift, ErTriIsect* __restrict__ isect, ErTriAttr* __restrict__ attr,
                                                  uint32_t* __restrict__ slot_to_tri) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k > n) return;
    ErTriIsect r;
    if (k == n) {           // one zero record past the end: the wide traversal fetches triangles in pairs
        memset(&r, 0, sizeof(r));
        isect[k] = r;
        return;
    }
    const uint32_t id = sorted_ids[new_order[k]];
    const float* p = v + (size_t)id * 9;
    for (int a = 0; a < 3; a++) { r.v0[a] = p[a]; r.v1[a] = p[3 + a]; r.v2[a] = p[6 + a]; }
    r.tri_id = (int32_t)id;
    r.lift = lift[id];
    r.sign = sign[id];
    isect[k] = r;
    ErTriAttr t;
    for (int j = 0; j < 3; j++)
        for (int a = 0; a < 3; a++) { t.n[j][a] = nrm[(size_t)id * 9 + 3 * j + a]; t.t[j][a] = tan[(size_t)id * 9 + 3 * j + a]; }
    for (int j = 0; j < 3; j++) { t.uv[j][0] = uv[(size_t)id * 6 + 2 * j]; t.uv[j][1] = uv[(size_t)id * 6 + 2 * j + 1]; }
    t.material = mat[id];
    for (unsigned j = 0; j < sizeof(t.pad) / sizeof(t.pad[0]); j++) t.pad[j] = 0;
    attr[k] = t;
    slot_to_tri[k] = id;
}

template <class T>
struct Dev {
    T* p = nullptr;
    Dev() = default;
    Dev(const Dev&) = delete;
    Dev& operator=(const Dev&) = delete;
    ~Dev() { if (p) (void)hipFree(p); }
    T* take() { T* q = p; p = nullptr; return q; }
};

#define GB_OK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? -2 : -1; } \
    } while (0)

// everything the build keeps on the device between its stages
struct GpuBuild {
    uint32_t n = 0, n_inner = 0;
    hipStream_t st = nullptr;
    Dev<float> d_v, d_n, d_lift;
    Dev<Box3> d_box;
    Dev<uint32_t> d_ids2;       // slot -> triangle, the binary tree's order
    Dev<int> d_pi;              // parent link of every inner node: parent * 2 + side
    Dev<ErNode> d_nodes;
    Dev<unsigned> d_g;
    unsigned g[12] = {0};
    ~GpuBuild() { if (st) (void)hipStreamDestroy(st); }

    // stages 1-2 of the header comment; returns 0, > 0 (declined) or -1
    int binary(const float* vertices, const float* normals, uint32_t n_, int device, std::string& err) {
        n = n_;
        n_inner = n - 1;
        const uint32_t blocks = (n + 255) / 256;
        GB_OK(hipSetDevice(device));
        GB_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        GB_OK(hipMalloc(&d_v.p, (size_t)n * 36));
        GB_OK(hipMalloc(&d_n.p, (size_t)n * 36));
        GB_OK(hipMalloc(&d_lift.p, (size_t)n * 4));
        GB_OK(hipMalloc(&d_box.p, (size_t)n * sizeof(Box3)));
        GB_OK(hipMalloc(&d_ids2.p, (size_t)n * 4));
        GB_OK(hipMalloc(&d_pi.p, (size_t)n * 4));
        GB_OK(hipMalloc(&d_nodes.p, (size_t)n_inner * sizeof(ErNode)));
        GB_OK(hipMalloc(&d_g.p, 12 * 4));
        GB_OK(hipMemcpyAsync(d_v.p, vertices, (size_t)n * 36, hipMemcpyHostToDevice, st));
        GB_OK(hipMemcpyAsync(d_n.p, normals, (size_t)n * 36, hipMemcpyHostToDevice, st));
        // [0] vmax, [7] lift max; [8] the deepest node's depth, [9] leaf count (set on the host by sah())
        for (int k = 0; k < 12; k++) g[k] = 0;
        GB_OK(hipMemcpyAsync(d_g.p, g, sizeof(g), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_scene_bounds, dim3(blocks), dim3(256), 0, st, d_v.p, n, d_g.p);
        hipLaunchKernelGGL(k_prims, dim3(blocks), dim3(256), 0, st, d_v.p, d_n.p, n, d_g.p, d_box.p, d_lift.p, d_g.p + 7);
        int rc = sah(err);
        if (rc != 0) return rc;
        if (g[8] + 1 > ER_BVH_MAX_DEPTH - 1) {
            err = "binary tree of depth " + std::to_string(g[8] + 1) + " exceeds the traversal stack bound";
            return 2;
        }
        return 0;
    }
    // top-down binned SAH, one tree level per round (comment above k_sah_cen); leaves d_ids2 = slot -> triangle, d_nodes, d_pi, g[8], g[9]
    int sah(std::string& err) {
        const uint32_t blocks = (n + 255) / 256;
        const uint32_t act_cap = n / (ER_BVH_LEAF_MAX + 1) + 2;                       // an active node holds more than ER_BVH_LEAF_MAX triangles
        // bins for this many nodes at a time; a fuller round runs in chunks (each chunk's bin kernel scans the triangles once more: at 10 M
        // triangles the two or three widest levels take 13 scans instead of 4, a few ms of 330).  2^18 nodes x 1 344 B = 352 MB; 2^20 (1.4 GB)
        // until round 6, which on a GPU that holds other scenes or ranks turned into the host fallback for want of memory
        const uint32_t bin_cap = std::min<uint32_t>(act_cap, 1u << 18);
        Dev<float> d_cen;
        Dev<uint32_t> d_idx[2], d_flag, d_scan, d_kids, d_koff;
        Dev<int> d_nof[2];
        Dev<SahAct> d_act[2];
        Dev<SahSplit> d_split;
        Dev<int2> d_cact;
        Dev<unsigned> d_cbv, d_bins, d_tot;
        Dev<char> d_st;
        GB_OK(hipMalloc(&d_cen.p, (size_t)n * 12));
        for (int k = 0; k < 2; k++) {
            GB_OK(hipMalloc(&d_idx[k].p, (size_t)n * 4));
            GB_OK(hipMalloc(&d_nof[k].p, (size_t)n * 4));
            GB_OK(hipMalloc(&d_act[k].p, (size_t)act_cap * sizeof(SahAct)));
        }
        GB_OK(hipMalloc(&d_flag.p, (size_t)n * 4));
        GB_OK(hipMalloc(&d_scan.p, (size_t)n * 4));
        GB_OK(hipMalloc(&d_kids.p, (size_t)act_cap * 4));
        GB_OK(hipMalloc(&d_koff.p, (size_t)act_cap * 4));
        GB_OK(hipMalloc(&d_split.p, (size_t)act_cap * sizeof(SahSplit)));
        GB_OK(hipMalloc(&d_cact.p, (size_t)act_cap * sizeof(int2)));
        GB_OK(hipMalloc(&d_cbv.p, (size_t)act_cap * 6 * 4));
        GB_OK(hipMalloc(&d_bins.p, (size_t)bin_cap * SAH_NODE_WORDS * 4));
        GB_OK(hipMalloc(&d_tot.p, 8));
        size_t sb1 = 0, sb2 = 0;
        GB_OK(rocprim::exclusive_scan(nullptr, sb1, d_flag.p, d_scan.p, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
        GB_OK(rocprim::exclusive_scan(nullptr, sb2, d_kids.p, d_koff.p, 0u, (size_t)act_cap, rocprim::plus<uint32_t>(), st));
        const size_t scan_bytes = std::max(sb1, sb2);
        GB_OK(hipMalloc(&d_st.p, scan_bytes ? scan_bytes : 16));
        hipLaunchKernelGGL(k_sah_cen, dim3(blocks), dim3(256), 0, st, d_v.p, n, d_cen.p);
        hipLaunchKernelGGL(k_sah_init, dim3(blocks), dim3(256), 0, st, n, d_idx[0].p, d_nof[0].p);
        GB_OK(hipMemcpyAsync(g, d_g.p, sizeof(g), hipMemcpyDeviceToHost, st));
        GB_OK(hipStreamSynchronize(st));
        SahAct root;
        root.node = 0; root.lo = 0; root.hi = n; root.depth = 0;
        {   // (the root's box is only read if every centroid of the scene coincides: any box that holds the scene will do)
            float vmax; memcpy(&vmax, &g[0], 4);
            const float r = vmax * 1.00001f + 1e-30f;
            for (int m = 0; m < 3; m++) { root.blo[m] = -r; root.bhi[m] = r; }
        }
        GB_OK(hipMemcpyAsync(d_act[0].p, &root, sizeof(root), hipMemcpyHostToDevice, st));
        uint32_t n_act = 1, created = 1, leaves = 0, depth = 0;
        int cur = 0;
        while (n_act > 0) {
            depth++;
            if (depth > ER_BVH_MAX_DEPTH) { err = "binned-SAH build deeper than its own guard allows"; return -1; }
            const uint32_t ab = (n_act + 255) / 256;
            const size_t clear_items = std::max((size_t)n_act * 6, (size_t)std::min(n_act, bin_cap) * SAH_NODE_WORDS);
            hipLaunchKernelGGL(k_sah_clear, dim3((unsigned)((clear_items + 255) / 256)), dim3(256), 0, st, d_cbv.p, n_act, d_bins.p, std::min(n_act, bin_cap));
            hipLaunchKernelGGL(k_sah_cb, dim3(blocks), dim3(256), 0, st, n, d_idx[cur].p, d_nof[cur].p, d_cen.p, d_cbv.p);
            for (uint32_t a0 = 0; a0 < n_act; a0 += bin_cap) {
                const uint32_t a1 = std::min(n_act, a0 + bin_cap);
                if (a0 > 0) hipLaunchKernelGGL(k_sah_clear, dim3((unsigned)(((size_t)(a1 - a0) * SAH_NODE_WORDS + 255) / 256)), dim3(256), 0, st, d_cbv.p, 0u, d_bins.p, a1 - a0);
                hipLaunchKernelGGL(k_sah_bin, dim3(blocks), dim3(256), 0, st, n, d_idx[cur].p, d_nof[cur].p, d_cen.p, d_box.p, d_cbv.p, d_bins.p, (int)a0, (int)a1);
                hipLaunchKernelGGL(k_sah_split, dim3((a1 - a0 + 63) / 64), dim3(64), 0, st, d_act[cur].p, d_cbv.p, d_bins.p, (int)a0, (int)a1, d_split.p, d_nodes.p);
            }
            hipLaunchKernelGGL(k_sah_count, dim3(ab), dim3(256), 0, st, d_split.p, n_act, d_kids.p);
            size_t sb = scan_bytes;
            GB_OK(rocprim::exclusive_scan(d_st.p, sb, d_kids.p, d_koff.p, 0u, (size_t)n_act, rocprim::plus<uint32_t>(), st));
            GB_OK(hipMemsetAsync(d_tot.p, 0, 8, st));
            hipLaunchKernelGGL(k_sah_children, dim3(ab), dim3(256), 0, st, d_act[cur].p, d_split.p, d_koff.p, n_act, created, d_nodes.p, d_pi.p, d_act[cur ^ 1].p, d_cact.p, d_tot.p);
            hipLaunchKernelGGL(k_sah_flags, dim3(blocks), dim3(256), 0, st, n, d_idx[cur].p, d_nof[cur].p, d_cen.p, d_act[cur].p, d_split.p, d_cbv.p, d_flag.p);
            sb = scan_bytes;
            GB_OK(rocprim::exclusive_scan(d_st.p, sb, d_flag.p, d_scan.p, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
            hipLaunchKernelGGL(k_sah_scatter, dim3(blocks), dim3(256), 0, st, n, d_idx[cur].p, d_nof[cur].p, d_flag.p, d_scan.p, d_act[cur].p, d_split.p, d_cact.p,
                               d_idx[cur ^ 1].p, d_nof[cur ^ 1].p);
            unsigned tot[2] = {0, 0};
            GB_OK(hipMemcpyAsync(tot, d_tot.p, 8, hipMemcpyDeviceToHost, st));
            GB_OK(hipStreamSynchronize(st));
            created += tot[0];
            leaves += tot[1];
            if (created > n_inner || tot[0] > act_cap) { err = "binned-SAH build made more nodes than triangles"; return -1; }
            n_act = tot[0];
            cur ^= 1;
        }
        GB_OK(hipGetLastError());
        // slot -> triangle: the final order of the triangles (the rest of the build reads it from d_ids2)
        GB_OK(hipMemcpyAsync(d_ids2.p, d_idx[cur].p, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        GB_OK(hipStreamSynchronize(st));
        n_inner = created;                 // (two-triangle leaves come out directly: fewer than n - 1 nodes)
        g[8] = depth - 1;                  // the deepest node's depth; the check in binary() adds the leaf level
        g[9] = leaves;
        if (getenv("ER_GPU_BUILD_VERBOSE")) fprintf(stderr, "[er_gpu_build] binned SAH on the device: %u triangles, %u levels, %u nodes, %u leaves\n", n, depth, created, leaves);
        return 0;
    }
    float lift_bound() const { float lm; memcpy(&lm, &g[7], 4); return lm; }
};

}  // namespace

hipError_t er_probe_gpu_build(const char** which) {   // see er_kernels.h
    hipFuncAttributes at;
    *which = "k_scene_bounds (er_gpu_build.hip)";
    return hipFuncGetAttributes(&at, (const void*)k_scene_bounds);
}

int er_gpu_build_device(const ErGpuSceneArrays& a, uint32_t n, int device, ErGpuBvhDevice* out, std::string& err) {
    auto t0 = std::chrono::steady_clock::now();
    if (n <= ER_BVH_LEAF_MAX) { err = "too few triangles for the device builder"; return 1; }
    if (const int dbg = er_debug_gpu_build_failure.load()) {      // (test hook er_debug_set_gpu_build_failure: the caller's handling of a failed build)
        err = dbg == 2 ? "simulated failure: out of device memory" : "simulated failure: a fault inside the builder";
        return dbg == 2 ? -2 : -1;
    }
    GpuBuild B;
    int rc = B.binary(a.vertices, a.normals, n, device, err);
    if (rc != 0) return rc;
    const uint32_t n_inner = B.n_inner;
    hipStream_t st = B.st;
    // ---- dynamic programme ----
    Dev<DpD> d_dp;
    Dev<unsigned> d_arrived;
    GB_OK(hipMalloc(&d_dp.p, (size_t)n_inner * sizeof(DpD)));
    GB_OK(hipMalloc(&d_arrived.p, (size_t)n_inner * 4));
    GB_OK(hipMemsetAsync(d_arrived.p, 0, (size_t)n_inner * 4, st));
    hipLaunchKernelGGL(k_dp, dim3((n_inner + 255) / 256), dim3(256), 0, st, B.d_nodes.p, n_inner, B.d_pi.p, d_dp.p, d_arrived.p);
    // ---- breadth-first emission of the wide nodes, one level per iteration ----
    Dev<WorkD> d_work[2];
    Dev<uint32_t> d_ci, d_ti, d_co, d_to, d_new_order, d_inv, d_s2t;
    Dev<ErNode8> d_n8;
    Dev<char> d_scan_tmp;
    GB_OK(hipMalloc(&d_work[0].p, (size_t)n_inner * sizeof(WorkD)));
    GB_OK(hipMalloc(&d_work[1].p, (size_t)n_inner * sizeof(WorkD)));
    GB_OK(hipMalloc(&d_ci.p, (size_t)n_inner * 4));
    GB_OK(hipMalloc(&d_ti.p, (size_t)n_inner * 4));
    GB_OK(hipMalloc(&d_co.p, (size_t)n_inner * 4));
    GB_OK(hipMalloc(&d_to.p, (size_t)n_inner * 4));
    GB_OK(hipMalloc(&d_new_order.p, (size_t)n * 4));
    GB_OK(hipMalloc(&d_inv.p, (size_t)n * 4));
    GB_OK(hipMalloc(&d_s2t.p, (size_t)n * 4));
    GB_OK(hipMalloc(&d_n8.p, (size_t)n_inner * sizeof(ErNode8)));
    size_t scan_bytes = 0;
    GB_OK(rocprim::exclusive_scan(nullptr, scan_bytes, d_ci.p, d_co.p, 0u, (size_t)n_inner, rocprim::plus<uint32_t>(), st));
    GB_OK(hipMalloc(&d_scan_tmp.p, scan_bytes ? scan_bytes : 16));
    const WorkD root{0u, 0};
    GB_OK(hipMemcpyAsync(d_work[0].p, &root, sizeof(root), hipMemcpyHostToDevice, st));
    uint32_t m = 1, nodes8_count = 1, tris_done = 0, depth8 = 0;
    int cur = 0;
    while (m > 0) {
        depth8++;
        const uint32_t wb = (m + 127) / 128;
        hipLaunchKernelGGL(k_wide_count, dim3(wb), dim3(128), 0, st, B.d_nodes.p, d_dp.p, d_work[cur].p, m, d_ci.p, d_ti.p);
        size_t sb = scan_bytes;
        GB_OK(rocprim::exclusive_scan(d_scan_tmp.p, sb, d_ci.p, d_co.p, 0u, (size_t)m, rocprim::plus<uint32_t>(), st));
        sb = scan_bytes;
        GB_OK(rocprim::exclusive_scan(d_scan_tmp.p, sb, d_ti.p, d_to.p, 0u, (size_t)m, rocprim::plus<uint32_t>(), st));
        hipLaunchKernelGGL(k_wide_emit, dim3(wb), dim3(128), 0, st, B.d_nodes.p, d_dp.p, d_work[cur].p, m, d_co.p, d_to.p, nodes8_count, tris_done,
                           d_work[cur ^ 1].p, d_n8.p, d_new_order.p);
        uint32_t last[4];
        GB_OK(hipMemcpyAsync(&last[0], d_co.p + (m - 1), 4, hipMemcpyDeviceToHost, st));
        GB_OK(hipMemcpyAsync(&last[1], d_ci.p + (m - 1), 4, hipMemcpyDeviceToHost, st));
        GB_OK(hipMemcpyAsync(&last[2], d_to.p + (m - 1), 4, hipMemcpyDeviceToHost, st));
        GB_OK(hipMemcpyAsync(&last[3], d_ti.p + (m - 1), 4, hipMemcpyDeviceToHost, st));
        GB_OK(hipStreamSynchronize(st));
        const uint32_t new_nodes = last[0] + last[1];
        tris_done += last[2] + last[3];
        nodes8_count += new_nodes;
        m = new_nodes;
        cur ^= 1;
        if (nodes8_count > n_inner || depth8 > ER_BVH_MAX_DEPTH) { err = "wide-node emission out of bounds"; return -1; }
    }
    if (tris_done != n) { err = "wide-node emission lost triangles (" + std::to_string(tris_done) + " of " + std::to_string(n) + ")"; return -1; }
    // ---- binary tree follows the new slot order; final records ----
    hipLaunchKernelGGL(k_inverse, dim3((n + 255) / 256), dim3(256), 0, st, d_new_order.p, n, d_inv.p);
    hipLaunchKernelGGL(k_fix_leaves, dim3((n_inner + 255) / 256), dim3(256), 0, st, B.d_nodes.p, n_inner, d_inv.p);
    Dev<float> d_tan, d_uv, d_sign;
    Dev<int> d_mat;
    GB_OK(hipMalloc(&d_tan.p, (size_t)n * 36));
    GB_OK(hipMalloc(&d_uv.p, (size_t)n * 24));
    GB_OK(hipMalloc(&d_sign.p, (size_t)n * 4));
    GB_OK(hipMalloc(&d_mat.p, (size_t)n * 4));
    GB_OK(hipMemcpyAsync(d_tan.p, a.tangents, (size_t)n * 36, hipMemcpyHostToDevice, st));
    GB_OK(hipMemcpyAsync(d_uv.p, a.uvs, (size_t)n * 24, hipMemcpyHostToDevice, st));
    GB_OK(hipMemcpyAsync(d_sign.p, a.tangent_sign, (size_t)n * 4, hipMemcpyHostToDevice, st));
    GB_OK(hipMemcpyAsync(d_mat.p, a.material_id, (size_t)n * 4, hipMemcpyHostToDevice, st));
    const size_t n8_pieces = (size_t)nodes8_count * ER_NODE8_PIECES + 8;
    const size_t geom_f4 = n8_pieces + ((size_t)n + 1) * 3;
    Dev<float4> d_geom, d_attr;
    GB_OK(hipMalloc(&d_geom.p, geom_f4 * 16));
    GB_OK(hipMalloc(&d_attr.p, (size_t)n * ER_ATTR_PIECES * 16));
    GB_OK(hipMemsetAsync(d_geom.p, 0, n8_pieces * 16, st));
    GB_OK(hipMemcpy2DAsync(d_geom.p, (size_t)ER_NODE8_PIECES * 16, d_n8.p, sizeof(ErNode8), sizeof(ErNode8), nodes8_count, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_records, dim3((n + 256) / 256), dim3(256), 0, st, d_new_order.p, B.d_ids2.p, n, B.d_v.p, B.d_n.p, d_tan.p, d_uv.p, d_sign.p,
                       d_mat.p, B.d_lift.p, (ErTriIsect*)(d_geom.p + n8_pieces), (ErTriAttr*)d_attr.p, d_s2t.p);
    GB_OK(hipGetLastError());
    ErNode rootn;
    GB_OK(hipMemcpyAsync(&rootn, B.d_nodes.p, sizeof(ErNode), hipMemcpyDeviceToHost, st));
    GB_OK(hipStreamSynchronize(st));
    for (int k = 0; k < 3; k++) { out->lo[k] = std::fmin(rootn.lo0[k], rootn.lo1[k]); out->hi[k] = std::fmax(rootn.hi0[k], rootn.hi1[k]); }
    out->lift_bound = B.lift_bound();
    out->leaf_count = B.g[9];
    out->max_depth2 = B.g[8] + 1;
    out->max_depth8 = depth8;
    out->nodes8_count = nodes8_count;
    out->n8_pieces = n8_pieces;
    out->geom_f4 = geom_f4;
    out->nodes_f4 = (size_t)n_inner * 4;
    out->attr_f4 = (size_t)n * ER_ATTR_PIECES;
    out->nodes = (float4*)B.d_nodes.take();
    out->geom = d_geom.take();
    out->attr = d_attr.take();
    out->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}
