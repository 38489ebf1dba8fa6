// er_gpu_build.hip -- device-side builder of the binary BVH.
//
// The reference builds its tree on the host, one node at a time (reference src/BVH.cpp:132-415, 11.7 s for
// 1M triangles); the contract is only "same nearest hit", so the tree is free to differ.  The default builder of
// this library is the host binned-SAH build (er_bvh.cpp, 0.25 s + 0.2 s collapse at 1M on 16 threads); this file
// is the fast alternative (ER_FLAG_GPU_BUILD): the whole structure built on the GPU --
//
//   1. bounds      per-triangle padded box, centroid, lift bound (the same arithmetic as er_build_bvh); the
//                  centroid bounds of the scene by a wave reduction + one atomic pair per wave;
//   2. morton      63-bit Morton code of the centroid (21 bits per axis);
//   3. sort        rocPRIM radix sort of (code, triangle) pairs -- the only library call;
//   4. tree        Karras 2012: every inner node finds its key range and split by binary search on the length of
//                  the common prefix (ties broken by the position, so equal codes still make a finite tree);
//   5. refit       leaves walk up; the second child to arrive at a node (an atomic counter decides) unites the
//                  two boxes and goes on; heights ride along, so the depth bound of the traversal stacks can be
//                  checked (a tree deeper than ER_BVH_MAX_DEPTH - 1 makes the caller fall back to the host build);
//   6. pair        an inner node whose two children are single triangles becomes a two-triangle leaf of its
//                  parent (leaves of this library hold <= 2 triangles, neighbours in slot order).
//
//   7. collapse    the SAH-optimal collapse into 8-wide compressed nodes (the dynamic programme and the breadth-first
//                  layout of er_collapse_bvh8, er_bvh.cpp): the DP bottom-up with the same last-arrival rule as
//                  the refit, the emission one tree level per launch with prefix sums for the node and triangle
//                  positions, so the layout never depends on thread timing;
//   8. records     the binary tree's leaf references follow the new slot order; the 48-byte intersection and
//                  112-byte attribute records are written in that order straight from the scene arrays.
//
// Everything the kernels read is produced in place on the device; a few counters come back.  Measured on MI355X:
// 1M triangles 60 ms against 451 ms for the host build (er_bvh.cpp, 16 threads), 9.68M triangles 242 ms against
// 5.1 s.  A linear BVH is a slightly worse tree than the binned-SAH build -- 21.7 instead of 20.7 node visits per
// ray on the 1M soup, 950 instead of 961 Msamples/s -- which is why it is the option and not the default.  Every
// image is identical bit for bit whichever builder made the tree (tests/test_gpu_build.py).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
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

// pass 1: largest |coordinate| (absolute box pad) and the bounds of the centroids (Morton grid)
__global__ __launch_bounds__(256) void k_scene_bounds(const float* __restrict__ v, uint32_t n, unsigned* g /* [0] vmax bits, [1..3] lo, [4..6] hi */) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float vm = 0, c[3] = {INFINITY, INFINITY, INFINITY}, d[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < n) {
        const float* p = v + (size_t)i * 9;
        for (int k = 0; k < 9; k++) vm = fmaxf(vm, fabsf(p[k]));
        for (int a = 0; a < 3; a++) c[a] = d[a] = (p[a] + p[3 + a] + p[6 + a]) * (1.0f / 3.0f);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        vm = fmaxf(vm, __shfl_xor(vm, off, 64));
        for (int a = 0; a < 3; a++) { c[a] = fminf(c[a], __shfl_xor(c[a], off, 64)); d[a] = fmaxf(d[a], __shfl_xor(d[a], off, 64)); }
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&g[0], __float_as_uint(vm));     // vm >= 0: the bit pattern orders like the value
        for (int a = 0; a < 3; a++) { atomicMin(&g[1 + a], f2ord(c[a])); atomicMax(&g[4 + a], f2ord(d[a])); }
    }
}

__device__ __forceinline__ unsigned long long spread21(unsigned x) {   // 21 bits -> every third bit of 63
    unsigned long long v = x & 0x1fffffu;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

// pass 2: padded box, lift bound (er_build_bvh's arithmetic, reference src/Tri.h:106-112), Morton code
__global__ __launch_bounds__(256) void k_prims(const float* __restrict__ v, const float* __restrict__ nrm, uint32_t n, const unsigned* __restrict__ g,
                                                Box3* __restrict__ boxes, float* __restrict__ lift, unsigned long long* __restrict__ keys,
                                                uint32_t* __restrict__ ids, unsigned* __restrict__ lift_max) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    float my_lift = 0;
    if (i < n) {
        const float pad_abs = __uint_as_float(g[0]) * 1e-6f;
        const float* p = v + (size_t)i * 9;
        Box3 b;
        float c[3];
        for (int a = 0; a < 3; a++) {
            float lo = fminf(fminf(p[a], p[3 + a]), p[6 + a]), hi = fmaxf(fmaxf(p[a], p[3 + a]), p[6 + a]);
            float m = fmaxf(fabsf(lo), fabsf(hi));
            float pad = fmaxf(m * 4e-7f + 1e-37f, pad_abs);
            b.lo[a] = lo - pad;
            b.hi[a] = hi + pad;
            c[a] = (p[a] + p[3 + a] + p[6 + a]) * (1.0f / 3.0f);
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
        unsigned q[3];
        for (int a = 0; a < 3; a++) {
            float lo = ord2f(g[1 + a]), hi = ord2f(g[4 + a]);
            float t = hi > lo ? (c[a] - lo) / (hi - lo) : 0.0f;
            t = fminf(fmaxf(t, 0.0f), 1.0f);
            q[a] = (unsigned)fminf(t * 2097152.0f, 2097151.0f);
        }
        keys[i] = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
        ids[i] = i;
    }
    for (int off = 32; off >= 1; off >>= 1) my_lift = fmaxf(my_lift, __shfl_xor(my_lift, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(lift_max, __float_as_uint(my_lift));
}

// length of the common prefix of the keys at sorted positions i and j (position bits break ties); -1 outside
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, uint32_t n, int i, int j) {
    if (j < 0 || j >= (int)n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a != b) return __clzll((long long)(a ^ b));
    return 64 + __clz(i ^ j);
}

// Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees, and k-d trees", section 4
__global__ __launch_bounds__(256) void k_tree(const unsigned long long* __restrict__ keys, uint32_t n, int2* __restrict__ children,
                                               int* __restrict__ parent_inner, int* __restrict__ parent_leaf) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (int)n - 1) return;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2, prev = l; prev > 1; prev = t, t = (t + 1) / 2)
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    // child reference: >= 0 inner node; < 0 leaf ~((slot << 3) | 0)
    const int c0 = lo == gamma ? ~(gamma << 3) : gamma;
    const int c1 = hi == gamma + 1 ? ~((gamma + 1) << 3) : gamma + 1;
    children[i] = make_int2(c0, c1);
    if (c0 < 0) parent_leaf[gamma] = i * 2; else parent_inner[gamma] = i * 2;
    if (c1 < 0) parent_leaf[gamma + 1] = i * 2 + 1; else parent_inner[gamma + 1] = i * 2 + 1;
}

// refit: one thread per leaf walks up; the second arrival at a node owns it
__global__ __launch_bounds__(256) void k_refit(const Box3* __restrict__ boxes, const uint32_t* __restrict__ ids, uint32_t n, const int2* __restrict__ children,
                                                const int* __restrict__ parent_inner, const int* __restrict__ parent_leaf, ErNode* nodes,
                                                unsigned* visits, unsigned* heights, unsigned* max_height) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    Box3 b = boxes[ids[j]];
    int link = parent_leaf[j];
    unsigned h = 0;
    while (true) {
        const int p = link >> 1, side = link & 1;
        ErNode* nd = nodes + p;
        float* lo = side ? nd->lo1 : nd->lo0;
        float* hi = side ? nd->hi1 : nd->hi0;
        for (int a = 0; a < 3; a++) { lo[a] = b.lo[a]; hi[a] = b.hi[a]; }
        atomicMax(&heights[p], h + 1);
        __threadfence();
        if (atomicAdd(&visits[p], 1u) == 0) return;      // the sibling subtree is not finished: it will carry on
        __threadfence();
        const volatile float* l0 = nd->lo0; const volatile float* h0 = nd->hi0;
        const volatile float* l1 = nd->lo1; const volatile float* h1 = nd->hi1;
        for (int a = 0; a < 3; a++) { b.lo[a] = fminf(l0[a], l1[a]); b.hi[a] = fmaxf(h0[a], h1[a]); }
        h = atomicMax(&heights[p], 0u);                  // both children have reported: this is the node's height
        const int2 c = children[p];
        nd->c0 = c.x; nd->c1 = c.y; nd->pad[0] = 0; nd->pad[1] = 0;
        if (p == 0) { atomicMax(max_height, h); return; }
        link = parent_inner[p];
    }
}

// pair: a child that is an inner node over two single triangles becomes a two-triangle leaf
__global__ __launch_bounds__(256) void k_pair(ErNode* nodes, uint32_t n_inner, unsigned* leaf_count) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    unsigned leaves = 0;
    if (i < n_inner) {
        ErNode* nd = nodes + i;
        int c[2] = {nd->c0, nd->c1};
        // (a node over two single triangles is itself absorbed by its parent and must not count its leaves)
        const bool absorbed_self = i != 0 && c[0] < 0 && c[1] < 0 && ((~c[0]) & 7) == 0 && ((~c[1]) & 7) == 0;
        for (int k = 0; k < 2 && !absorbed_self; k++) {
            if (c[k] >= 0) {
                const int2 cc = make_int2(nodes[c[k]].c0, nodes[c[k]].c1);
                // (the child still holds its two single-triangle references: only parents rewrite, and only their own fields)
                if (cc.x < 0 && cc.y < 0 && ((~cc.x) & 7) == 0 && ((~cc.y) & 7) == 0) {
                    const int first = (~cc.x) >> 3;
                    c[k] = ~((first << 3) | 1);
                    leaves++;
                }
            } else {
                leaves++;
            }
        }
        nd->pad[0] = c[0];      // staged: written back by k_pair_commit so that no parent reads a rewritten child
        nd->pad[1] = c[1];
    }
    for (int off = 32; off >= 1; off >>= 1) leaves += __shfl_xor(leaves, off, 64);
    if ((threadIdx.x & 63) == 0 && leaves) atomicAdd(leaf_count, leaves);
}
__global__ __launch_bounds__(256) void k_pair_commit(ErNode* nodes, uint32_t n_inner, unsigned char* absorbed) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_inner) return;
    ErNode* nd = nodes + i;
    const int o0 = nd->c0, o1 = nd->c1;
    absorbed[i] = (i != 0 && o0 < 0 && o1 < 0 && ((~o0) & 7) == 0 && ((~o1) & 7) == 0) ? 1 : 0;
    nd->c0 = nd->pad[0];
    nd->c1 = nd->pad[1];
    nd->pad[0] = 0;
    nd->pad[1] = 0;
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
__global__ __launch_bounds__(256) void k_dp(const ErNode* __restrict__ nodes, uint32_t n_inner, const unsigned char* __restrict__ absorbed,
                                             const int* __restrict__ parent_inner, DpD* dp, unsigned* arrived) {
    uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_inner || absorbed[i]) return;
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
        if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return -1; } \
    } while (0)

// everything the build keeps on the device between its stages
struct GpuBuild {
    uint32_t n = 0, n_inner = 0;
    hipStream_t st = nullptr;
    Dev<float> d_v, d_n, d_lift;
    Dev<Box3> d_box;
    Dev<unsigned long long> d_keys, d_keys2;
    Dev<uint32_t> d_ids, d_ids2;
    Dev<int2> d_children;
    Dev<int> d_pi, d_pl;
    Dev<ErNode> d_nodes;
    Dev<unsigned> d_g, d_visits, d_heights;
    Dev<char> d_tmp;
    Dev<unsigned char> d_absorbed;
    unsigned g[12] = {0};
    ~GpuBuild() { if (st) (void)hipStreamDestroy(st); }

    // stages 1-6 of the header comment; returns 0, > 0 (declined) or -1
    int binary(const float* vertices, const float* normals, uint32_t n_, int device, std::string& err) {
        n = n_;
        n_inner = n - 1;
        const uint32_t blocks = (n + 255) / 256;
        size_t tmp_bytes = 0;
        GB_OK(hipSetDevice(device));
        GB_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        GB_OK(hipMalloc(&d_v.p, (size_t)n * 36));
        GB_OK(hipMalloc(&d_n.p, (size_t)n * 36));
        GB_OK(hipMalloc(&d_lift.p, (size_t)n * 4));
        GB_OK(hipMalloc(&d_box.p, (size_t)n * sizeof(Box3)));
        GB_OK(hipMalloc(&d_keys.p, (size_t)n * 8));
        GB_OK(hipMalloc(&d_keys2.p, (size_t)n * 8));
        GB_OK(hipMalloc(&d_ids.p, (size_t)n * 4));
        GB_OK(hipMalloc(&d_ids2.p, (size_t)n * 4));
        GB_OK(hipMalloc(&d_children.p, (size_t)n_inner * sizeof(int2)));
        GB_OK(hipMalloc(&d_pi.p, (size_t)n * 4));
        GB_OK(hipMalloc(&d_pl.p, (size_t)n * 4));
        GB_OK(hipMalloc(&d_nodes.p, (size_t)n_inner * sizeof(ErNode)));
        GB_OK(hipMalloc(&d_g.p, 12 * 4));
        GB_OK(hipMalloc(&d_visits.p, (size_t)n_inner * 4));
        GB_OK(hipMalloc(&d_heights.p, (size_t)n_inner * 4));
        GB_OK(hipMalloc(&d_absorbed.p, (size_t)n_inner));
        GB_OK(hipMemcpyAsync(d_v.p, vertices, (size_t)n * 36, hipMemcpyHostToDevice, st));
        GB_OK(hipMemcpyAsync(d_n.p, normals, (size_t)n * 36, hipMemcpyHostToDevice, st));
        // [0] vmax, [1..3] centroid lo (ordered), [4..6] centroid hi (ordered), [7] lift max, [8] max height, [9] leaf count
        for (int k = 0; k < 12; k++) g[k] = 0;
        g[1] = g[2] = g[3] = 0xffffffffu;
        GB_OK(hipMemcpyAsync(d_g.p, g, sizeof(g), hipMemcpyHostToDevice, st));
        GB_OK(hipMemsetAsync(d_visits.p, 0, (size_t)n_inner * 4, st));
        GB_OK(hipMemsetAsync(d_heights.p, 0, (size_t)n_inner * 4, st));
        hipLaunchKernelGGL(k_scene_bounds, dim3(blocks), dim3(256), 0, st, d_v.p, n, d_g.p);
        hipLaunchKernelGGL(k_prims, dim3(blocks), dim3(256), 0, st, d_v.p, d_n.p, n, d_g.p, d_box.p, d_lift.p, d_keys.p, d_ids.p, d_g.p + 7);
        GB_OK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys.p, d_keys2.p, d_ids.p, d_ids2.p, n, 0, 63, st));
        GB_OK(hipMalloc(&d_tmp.p, tmp_bytes ? tmp_bytes : 16));
        GB_OK(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_keys.p, d_keys2.p, d_ids.p, d_ids2.p, n, 0, 63, st));
        hipLaunchKernelGGL(k_tree, dim3((n_inner + 255) / 256), dim3(256), 0, st, d_keys2.p, n, d_children.p, d_pi.p, d_pl.p);
        hipLaunchKernelGGL(k_refit, dim3(blocks), dim3(256), 0, st, d_box.p, d_ids2.p, n, d_children.p, d_pi.p, d_pl.p, d_nodes.p, d_visits.p,
                           d_heights.p, d_g.p + 8);
        hipLaunchKernelGGL(k_pair, dim3((n_inner + 255) / 256), dim3(256), 0, st, d_nodes.p, n_inner, d_g.p + 9);
        hipLaunchKernelGGL(k_pair_commit, dim3((n_inner + 255) / 256), dim3(256), 0, st, d_nodes.p, n_inner, d_absorbed.p);
        GB_OK(hipGetLastError());
        GB_OK(hipMemcpyAsync(g, d_g.p, sizeof(g), hipMemcpyDeviceToHost, st));
        GB_OK(hipStreamSynchronize(st));
        if (g[8] + 1 > ER_BVH_MAX_DEPTH - 1) {
            err = "linear BVH of depth " + std::to_string(g[8] + 1) + " exceeds the traversal stack bound";
            return 2;
        }
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
    hipLaunchKernelGGL(k_dp, dim3((n_inner + 255) / 256), dim3(256), 0, st, B.d_nodes.p, n_inner, B.d_absorbed.p, B.d_pi.p, d_dp.p, d_arrived.p);
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
