// er_bvh.cpp -- host builder for the MI355X build's own BVH (see er_bvh.h).
// Replaces the role of Scene::buildBVH / BVH::build (reference src/Scene.cpp:122-143,
// src/BVH.cpp:132-415) with a different structure; the result contract is "same nearest hit".
#include "er_bvh.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>

namespace {

struct Prim {
    float lo[3], hi[3], c[3];
    uint32_t id;
};
struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; } }
    void grow(const float* l, const float* h) {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
    }
    void growp(const float* p) {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    float area() const {
        float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        if (!(x >= 0) || !(y >= 0) || !(z >= 0)) return 0;
        return 2 * (x * y + x * z + y * z);
    }
};

const int NBINS = 16;

struct Builder {
    std::vector<Prim> prims;
    std::vector<ErNode> nodes;          // scattered allocation, re-laid out at the end
    std::vector<uint8_t> node_depth;
    std::atomic<uint32_t> next_node{0};
    std::atomic<uint32_t> leaf_count{0};
    std::atomic<uint32_t> max_depth{0};

    static int ceil_log2(uint32_t v) { int r = 0; while ((1u << r) < v) r++; return r; }
    static int levels_needed(uint32_t n) { return ceil_log2((n + ER_BVH_LEAF_MAX - 1) / ER_BVH_LEAF_MAX); }

    int32_t make_leaf(uint32_t lo, uint32_t hi) {
        leaf_count.fetch_add(1, std::memory_order_relaxed);
        return ~(int32_t)((lo << 3) | (hi - lo - 1));
    }

    // returns the split position; fills child boxes
    uint32_t split(uint32_t lo, uint32_t hi, int depth, Box& bl, Box& br) {
        uint32_t n = hi - lo;
        Box cb;
        cb.reset();
        for (uint32_t i = lo; i < hi; i++) cb.growp(prims[i].c);
        bool force_median = depth + 1 + levels_needed(n) > ER_BVH_MAX_DEPTH - 1;
        int best_axis = -1, best_bin = -1;
        float best_cost = INFINITY;
        if (!force_median) {
            for (int axis = 0; axis < 3; axis++) {
                float ext = cb.hi[axis] - cb.lo[axis];
                if (!(ext > 0)) continue;
                Box bb[NBINS];
                uint32_t cnt[NBINS];
                for (int b = 0; b < NBINS; b++) { bb[b].reset(); cnt[b] = 0; }
                float scale = NBINS / ext;
                for (uint32_t i = lo; i < hi; i++) {
                    int b = (int)((prims[i].c[axis] - cb.lo[axis]) * scale);
                    b = b < 0 ? 0 : (b >= NBINS ? NBINS - 1 : b);
                    cnt[b]++;
                    bb[b].grow(prims[i].lo, prims[i].hi);
                }
                float ra[NBINS];
                uint32_t rc[NBINS];
                Box acc;
                acc.reset();
                uint32_t c = 0;
                for (int b = NBINS - 1; b >= 1; b--) {
                    if (cnt[b]) acc.grow(bb[b].lo, bb[b].hi);
                    c += cnt[b];
                    ra[b] = acc.area();
                    rc[b] = c;
                }
                acc.reset();
                c = 0;
                for (int b = 0; b < NBINS - 1; b++) {
                    if (cnt[b]) acc.grow(bb[b].lo, bb[b].hi);
                    c += cnt[b];
                    if (c == 0 || rc[b + 1] == 0) continue;
                    float cost = acc.area() * (float)c + ra[b + 1] * (float)rc[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
        }
        uint32_t mid;
        if (best_axis >= 0) {
            float ext = cb.hi[best_axis] - cb.lo[best_axis];
            float scale = NBINS / ext, base = cb.lo[best_axis];
            int ax = best_axis, bin = best_bin;
            Prim* p = std::partition(prims.data() + lo, prims.data() + hi, [=](const Prim& q) {
                int b = (int)((q.c[ax] - base) * scale);
                b = b < 0 ? 0 : (b >= NBINS ? NBINS - 1 : b);
                return b <= bin;
            });
            mid = (uint32_t)(p - prims.data());
        } else {
            // all centroids coincide (or depth guard): median split on the widest centroid axis
            int ax = 0;
            float e0 = cb.hi[0] - cb.lo[0], e1 = cb.hi[1] - cb.lo[1], e2 = cb.hi[2] - cb.lo[2];
            if (e1 > e0 && e1 >= e2) ax = 1; else if (e2 > e0 && e2 > e1) ax = 2;
            mid = lo + n / 2;
            std::nth_element(prims.data() + lo, prims.data() + mid, prims.data() + hi,
                             [=](const Prim& a, const Prim& b) { return a.c[ax] < b.c[ax] || (a.c[ax] == b.c[ax] && a.id < b.id); });
        }
        if (mid == lo || mid == hi) mid = lo + n / 2;   // cannot happen with the guards above; keep the tree finite
        bl.reset();
        br.reset();
        for (uint32_t i = lo; i < mid; i++) bl.grow(prims[i].lo, prims[i].hi);
        for (uint32_t i = mid; i < hi; i++) br.grow(prims[i].lo, prims[i].hi);
        return mid;
    }

    // builds the subtree over [lo,hi) (n > LEAF_MAX) and returns the inner node index
    int32_t build(uint32_t lo, uint32_t hi, int depth, int threads) {
        uint32_t me = next_node.fetch_add(1, std::memory_order_relaxed);
        node_depth[me] = (uint8_t)depth;
        uint32_t d = (uint32_t)depth + 1, cur = max_depth.load(std::memory_order_relaxed);
        while (d > cur && !max_depth.compare_exchange_weak(cur, d, std::memory_order_relaxed)) {}
        Box bl, br;
        uint32_t mid = split(lo, hi, depth, bl, br);
        int32_t c0, c1;
        auto child = [&](uint32_t a, uint32_t b, int th) -> int32_t {
            return (b - a <= ER_BVH_LEAF_MAX) ? make_leaf(a, b) : build(a, b, depth + 1, th);
        };
        if (threads > 1 && hi - lo > 8192) {
            int tl = threads / 2, tr = threads - tl;
            int32_t r0 = 0;
            std::thread t([&]() { r0 = child(lo, mid, tl); });
            c1 = child(mid, hi, tr);
            t.join();
            c0 = r0;
        } else {
            c0 = child(lo, mid, 1);
            c1 = child(mid, hi, 1);
        }
        ErNode& nd = nodes[me];
        for (int a = 0; a < 3; a++) { nd.lo0[a] = bl.lo[a]; nd.hi0[a] = bl.hi[a]; nd.lo1[a] = br.lo[a]; nd.hi1[a] = br.hi[a]; }
        nd.c0 = c0; nd.c1 = c1; nd.pad[0] = nd.pad[1] = 0;
        return (int32_t)me;
    }
};

}  // namespace

void er_build_bvh(const float* vertices, const float* normals, uint32_t tri_count, int threads, ErBvhBuild* out) {
    auto t0 = std::chrono::steady_clock::now();
    if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
    out->nodes.clear();
    out->slot_to_tri.clear();
    out->leaf_count = 0;
    out->max_depth = 0;
    out->lift_bound = 0;
    Builder B;
    B.prims.resize(tri_count);
    Box sb;
    sb.reset();
    double lift = 0;
    for (uint32_t i = 0; i < tri_count; i++) {
        Prim& p = B.prims[i];
        const float* v = vertices + (size_t)i * 9;
        Box b;
        b.reset();
        for (int k = 0; k < 3; k++) b.growp(v + 3 * k);
        for (int a = 0; a < 3; a++) {
            // conservative padding: Moller-Trumbore (reference src/Tri.h:41-77) accepts hits whose
            // computed position can sit a few ulp outside the exact vertex bounds
            float m = std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a]));
            float pad = m * 4e-7f + 1e-37f;
            p.lo[a] = b.lo[a] - pad;
            p.hi[a] = b.hi[a] + pad;
            p.c[a] = (v[a] + v[3 + a] + v[6 + a]) * (1.0f / 3.0f);
        }
        p.id = i;
        sb.grow(p.lo, p.hi);
        // bound on |shadingPosition - geomPosition| (src/Tri.h:106-112): the hit point is a convex
        // combination of the vertices and each p_j moves it by |dot(P - v_j, n_j)| * |n_j|
        const float* nn = normals + (size_t)i * 9;
        for (int j = 0; j < 3; j++) {
            double nx = nn[3 * j], ny = nn[3 * j + 1], nz = nn[3 * j + 2];
            double nl = std::sqrt(nx * nx + ny * ny + nz * nz);
            for (int k = 0; k < 3; k++) {
                if (k == j) continue;
                double dx = (double)v[3 * k] - v[3 * j], dy = (double)v[3 * k + 1] - v[3 * j + 1], dz = (double)v[3 * k + 2] - v[3 * j + 2];
                double l = std::fabs(dx * nx + dy * ny + dz * nz) * nl;
                if (l > lift) lift = l;
            }
        }
    }
    for (int a = 0; a < 3; a++) { out->lo[a] = tri_count ? sb.lo[a] : 0; out->hi[a] = tri_count ? sb.hi[a] : 0; }
    out->lift_bound = (float)(lift * 1.01);

    if (tri_count > 0) {
        B.nodes.resize(std::max<uint32_t>(tri_count, 1));
        B.node_depth.assign(B.nodes.size(), 0);
        int32_t root;
        if (tri_count <= ER_BVH_LEAF_MAX) {
            // a single leaf: wrap it in one node whose second child is empty
            uint32_t me = B.next_node.fetch_add(1);
            ErNode& nd = B.nodes[me];
            Box b;
            b.reset();
            for (uint32_t i = 0; i < tri_count; i++) b.grow(B.prims[i].lo, B.prims[i].hi);
            for (int a = 0; a < 3; a++) { nd.lo0[a] = b.lo[a]; nd.hi0[a] = b.hi[a]; nd.lo1[a] = 0; nd.hi1[a] = 0; }
            nd.c0 = B.make_leaf(0, tri_count);
            nd.c1 = ER_BVH_NO_CHILD;
            nd.pad[0] = nd.pad[1] = 0;
            B.max_depth = 1;
            root = (int32_t)me;
        } else {
            root = B.build(0, tri_count, 0, threads);
        }
        // re-layout: breadth-first for the top levels (shared by every ray, cache resident),
        // depth-first below (a subtree's nodes stay close together)
        uint32_t n_nodes = B.next_node.load();
        std::vector<ErNode>& src = B.nodes;
        std::vector<int32_t> remap(n_nodes, -1);
        std::vector<int32_t> order;
        order.reserve(n_nodes);
        const int TOP_LEVELS = 10;
        std::vector<int32_t> frontier{root}, next;
        for (int lvl = 0; lvl < TOP_LEVELS && !frontier.empty(); lvl++) {
            next.clear();
            for (int32_t ni : frontier) {
                remap[ni] = (int32_t)order.size();
                order.push_back(ni);
                if (src[ni].c0 >= 0 && src[ni].c0 != ER_BVH_NO_CHILD) next.push_back(src[ni].c0);
                if (src[ni].c1 >= 0 && src[ni].c1 != ER_BVH_NO_CHILD) next.push_back(src[ni].c1);
            }
            frontier.swap(next);
        }
        std::vector<int32_t> stack;
        for (auto it = frontier.rbegin(); it != frontier.rend(); ++it) stack.push_back(*it);
        while (!stack.empty()) {
            int32_t ni = stack.back();
            stack.pop_back();
            remap[ni] = (int32_t)order.size();
            order.push_back(ni);
            if (src[ni].c1 >= 0 && src[ni].c1 != ER_BVH_NO_CHILD) stack.push_back(src[ni].c1);
            if (src[ni].c0 >= 0 && src[ni].c0 != ER_BVH_NO_CHILD) stack.push_back(src[ni].c0);
        }
        out->nodes.resize(n_nodes);
        for (uint32_t i = 0; i < n_nodes; i++) {
            ErNode nd = src[order[i]];
            if (nd.c0 >= 0 && nd.c0 != ER_BVH_NO_CHILD) nd.c0 = remap[nd.c0];
            if (nd.c1 >= 0 && nd.c1 != ER_BVH_NO_CHILD) nd.c1 = remap[nd.c1];
            out->nodes[i] = nd;
        }
        out->leaf_count = B.leaf_count.load();
        out->max_depth = B.max_depth.load();
    }
    out->slot_to_tri.resize(tri_count);
    for (uint32_t i = 0; i < tri_count; i++) out->slot_to_tri[i] = B.prims[i].id;
    out->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
