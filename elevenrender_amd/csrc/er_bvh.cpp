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

#ifndef ER_SAH_BINS
#define ER_SAH_BINS 16
#endif
const int NBINS = ER_SAH_BINS;

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

// ---- collapse the binary tree into the 8-wide compressed tree (see ErNode8) ----
// SAH-optimal collapse by dynamic programming (Ylitie, Karras, Laine 2017, section 4.1): for every
// binary node n and budget i in 1..7, C(n,i) = cheapest way to represent n's subtree as at most i
// children of a wide node; a child is a leaf (whole subtree, <= ER_BVH_LEAF_MAX triangles) or a wide node.
namespace {
#ifndef ER_C_PRIM
#define ER_C_PRIM 0.3f
#endif
const float C_NODE = 1.0f, C_PRIM = ER_C_PRIM;
struct Dp {
    float C[8];          // C[1..7]
    uint8_t dec[8];      // dec[1]: 0 = leaf, k>0 = wide node giving k slots to child 0; dec[i>=2]: 0 = same as i-1, k>0 = split k / i-k
    uint32_t first, count;
};
inline float area3(const float* lo, const float* hi) {
    float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
    return 2.0f * (x * y + x * z + y * z);
}
}  // namespace

void er_collapse_bvh8(ErBvhBuild* out) {
    out->nodes8.clear();
    out->max_depth8 = 0;
    if (out->nodes.empty()) return;
    std::vector<ErNode>& N2 = out->nodes;
    const uint32_t n2 = (uint32_t)N2.size();
    const uint32_t tri_count = (uint32_t)out->slot_to_tri.size();

    // ---- bottom-up DP over the binary tree (explicit post-order) ----
    std::vector<Dp> dp(n2);
    auto leaf_first = [](int32_t ref) { return ((uint32_t)~ref) >> 3; };
    auto leaf_count = [](int32_t ref) { return (((uint32_t)~ref) & 7u) + 1u; };
    auto childC = [&](int32_t ref, float area, int i) -> float {
        if (ref == ER_BVH_NO_CHILD) return 0.0f;
        if (ref < 0) return area * (float)leaf_count(ref) * C_PRIM;
        return dp[ref].C[i];
    };
    {
        std::vector<std::pair<int32_t, int>> st;
        st.push_back({0, 0});
        while (!st.empty()) {
            int32_t n = st.back().first;
            int phase = st.back().second;
            const ErNode& nd = N2[n];
            if (phase == 0) {
                st.back().second = 1;
                if (nd.c0 >= 0 && nd.c0 != ER_BVH_NO_CHILD) st.push_back({nd.c0, 0});
                if (nd.c1 >= 0 && nd.c1 != ER_BVH_NO_CHILD) st.push_back({nd.c1, 0});
                continue;
            }
            st.pop_back();
            Dp& D = dp[n];
            const bool has1 = nd.c1 != ER_BVH_NO_CHILD;
            float a0 = area3(nd.lo0, nd.hi0), a1 = has1 ? area3(nd.lo1, nd.hi1) : 0.0f;
            float lo[3], hi[3];
            for (int a = 0; a < 3; a++) {
                lo[a] = has1 ? std::min(nd.lo0[a], nd.lo1[a]) : nd.lo0[a];
                hi[a] = has1 ? std::max(nd.hi0[a], nd.hi1[a]) : nd.hi0[a];
            }
            float an = area3(lo, hi);
            uint32_t f0 = nd.c0 < 0 ? leaf_first(nd.c0) : dp[nd.c0].first, k0 = nd.c0 < 0 ? leaf_count(nd.c0) : dp[nd.c0].count;
            uint32_t f1 = !has1 ? f0 : (nd.c1 < 0 ? leaf_first(nd.c1) : dp[nd.c1].first), k1 = !has1 ? 0 : (nd.c1 < 0 ? leaf_count(nd.c1) : dp[nd.c1].count);
            D.first = std::min(f0, f1);
            D.count = k0 + k1;
            float c_leaf = D.count <= ER_BVH_LEAF_MAX ? an * (float)D.count * C_PRIM : INFINITY;
            float best = INFINITY;
            int bk = 1;
            if (has1) {
                for (int k = 1; k <= 7; k++) {
                    float c = childC(nd.c0, a0, k) + childC(nd.c1, a1, 8 - k);
                    if (c < best) { best = c; bk = k; }
                }
            } else {
                best = childC(nd.c0, a0, 7);
                bk = 7;
            }
            float c_int = an * C_NODE + best;
            if (c_leaf <= c_int) { D.C[1] = c_leaf; D.dec[1] = 0; } else { D.C[1] = c_int; D.dec[1] = (uint8_t)bk; }
            for (int i = 2; i <= 7; i++) {
                float bd = INFINITY;
                int kk = 0;
                if (has1)
                    for (int k = 1; k < i; k++) {
                        float c = childC(nd.c0, a0, k) + childC(nd.c1, a1, i - k);
                        if (c < bd) { bd = c; kk = k; }
                    }
                if (bd < D.C[i - 1]) { D.C[i] = bd; D.dec[i] = (uint8_t)kk; } else { D.C[i] = D.C[i - 1]; D.dec[i] = 0; }
            }
        }
    }

    // ---- top-down reconstruction, breadth-first so that a node's inner children are consecutive ----
    std::vector<uint32_t> new_order;          // new slot -> old slot
    new_order.reserve(tri_count);
    struct Child { int32_t ref; float lo[3], hi[3]; bool as_leaf; int32_t parent2; int which; };
    struct Work { uint32_t n8; int32_t n2; uint32_t depth; };
    std::vector<Work> queue;
    out->nodes8.emplace_back();
    queue.push_back(Work{0, 0, 1});
    // collect(ref, box, budget i) appends the children that represent subtree `ref` within i slots
    std::vector<Child> ch;
    struct Item { int32_t ref; float lo[3], hi[3]; int i; int32_t parent2; int which; };
    auto collect = [&](Item root_item) {
        std::vector<Item> st{root_item};
        while (!st.empty()) {
            Item it = st.back();
            st.pop_back();
            Child c;
            c.ref = it.ref; memcpy(c.lo, it.lo, 12); memcpy(c.hi, it.hi, 12); c.parent2 = it.parent2; c.which = it.which; c.as_leaf = false;
            if (it.ref < 0) { c.as_leaf = true; ch.push_back(c); continue; }
            const Dp& D = dp[it.ref];
            int i = it.i;
            while (i >= 2 && D.dec[i] == 0) i--;
            if (i == 1) { c.as_leaf = D.dec[1] == 0; ch.push_back(c); continue; }
            const ErNode& nd = N2[it.ref];
            int k = D.dec[i];
            Item a, b;
            a.ref = nd.c0; memcpy(a.lo, nd.lo0, 12); memcpy(a.hi, nd.hi0, 12); a.i = k; a.parent2 = it.ref; a.which = 0;
            b.ref = nd.c1; memcpy(b.lo, nd.lo1, 12); memcpy(b.hi, nd.hi1, 12); b.i = i - k; b.parent2 = it.ref; b.which = 1;
            st.push_back(b);
            st.push_back(a);
        }
    };
    // rewrites the binary tree's leaf references inside subtree `ref` after its triangles moved by `delta` slots
    auto shift_leaves = [&](int32_t parent2, int which, int32_t ref, int64_t delta) {
        struct R { int32_t parent; int which; int32_t ref; };
        std::vector<R> st{R{parent2, which, ref}};
        while (!st.empty()) {
            R r = st.back();
            st.pop_back();
            if (r.ref == ER_BVH_NO_CHILD) continue;
            if (r.ref < 0) {
                uint32_t nf = (uint32_t)((int64_t)leaf_first(r.ref) + delta), cnt = leaf_count(r.ref);
                int32_t nref = ~(int32_t)((nf << 3) | (cnt - 1));
                if (r.which == 0) N2[r.parent].c0 = nref; else N2[r.parent].c1 = nref;
            } else {
                st.push_back(R{r.ref, 0, N2[r.ref].c0});
                st.push_back(R{r.ref, 1, N2[r.ref].c1});
            }
        }
    };
    for (size_t qi = 0; qi < queue.size(); qi++) {
        Work w = queue[qi];
        out->max_depth8 = std::max(out->max_depth8, w.depth);
        ch.clear();
        {
            const ErNode& nd = N2[w.n2];
            const Dp& D = dp[w.n2];
            const bool has1 = nd.c1 != ER_BVH_NO_CHILD;
            // the root may itself be a "leaf" by cost; it still becomes one wide node with leaf children
            int k = D.dec[1] != 0 ? D.dec[1] : (has1 ? 4 : 7);
            Item a;
            a.ref = nd.c0; memcpy(a.lo, nd.lo0, 12); memcpy(a.hi, nd.hi0, 12); a.i = k; a.parent2 = w.n2; a.which = 0;
            collect(a);
            if (has1) {
                Item b;
                b.ref = nd.c1; memcpy(b.lo, nd.lo1, 12); memcpy(b.hi, nd.hi1, 12); b.i = 8 - k; b.parent2 = w.n2; b.which = 1;
                collect(b);
            }
        }
        // node bounds
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (const Child& c : ch) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], c.lo[a]); hi[a] = std::max(hi[a], c.hi[a]); }
        // slot assignment: greedy on score = (centroid - centre) . (+-1,+-1,+-1)
        int slot_of[8], child_in[8];
        for (int i = 0; i < 8; i++) { slot_of[i] = -1; child_in[i] = -1; }
        struct Score { float v; int c, s; };
        std::vector<Score> sc;
        for (size_t i = 0; i < ch.size(); i++)
            for (int s8 = 0; s8 < 8; s8++) {
                float v = 0;
                for (int a = 0; a < 3; a++) {
                    float rel = 0.5f * (ch[i].lo[a] + ch[i].hi[a]) - 0.5f * (lo[a] + hi[a]);
                    v += ((s8 >> a) & 1) ? rel : -rel;
                }
                sc.push_back(Score{v, (int)i, s8});
            }
        std::stable_sort(sc.begin(), sc.end(), [](const Score& a, const Score& b) { return a.v > b.v; });
        for (const Score& x : sc)
            if (slot_of[x.c] < 0 && child_in[x.s] < 0) { slot_of[x.c] = x.s; child_in[x.s] = x.c; }
        // fill the node
        ErNode8 nd;
        memset(&nd, 0, sizeof(nd));
        float scale[3];
        for (int a = 0; a < 3; a++) {
            nd.p[a] = lo[a];
            float ext = hi[a] - lo[a];
            int e = 0;
            if (ext > 0) { (void)std::frexp(ext / 255.0f, &e); }   // 2^e > ext/255
            else e = -126;
            if (e < -126) e = -126;
            while (lo[a] + 255.0f * std::ldexp(1.0f, e) < hi[a]) e++;   // 255 steps must reach hi in float arithmetic
            nd.e[a] = (uint8_t)(e + 127);
            scale[a] = std::ldexp(1.0f, e);
        }
        nd.child_base = (uint32_t)out->nodes8.size();
        nd.tri_base = (uint32_t)new_order.size();
        for (int s8 = 0; s8 < 8; s8++) {
            int ci = child_in[s8];
            if (ci < 0) continue;
            const Child& c = ch[ci];
            for (int a = 0; a < 3; a++) {
                float fl = std::floor((c.lo[a] - nd.p[a]) / scale[a]);
                float fh = std::ceil((c.hi[a] - nd.p[a]) / scale[a]);
                int ql = (int)std::min(255.0f, std::max(0.0f, fl));
                int qh = (int)std::min(255.0f, std::max(0.0f, fh));
                // conservative in exactly the arithmetic the kernel decodes with: p + q * scale (one rounding)
                while (ql > 0 && nd.p[a] + (float)ql * scale[a] > c.lo[a]) ql--;
                while (qh < 255 && nd.p[a] + (float)qh * scale[a] < c.hi[a]) qh++;
                nd.qlo[a][s8] = (uint8_t)ql;
                nd.qhi[a][s8] = (uint8_t)qh;
            }
            if (!c.as_leaf) {
                nd.imask |= (uint8_t)(1u << s8);
                uint32_t idx = (uint32_t)out->nodes8.size();
                out->nodes8.emplace_back();
                queue.push_back(Work{idx, c.ref, w.depth + 1});
            } else {
                uint32_t first = c.ref < 0 ? leaf_first(c.ref) : dp[c.ref].first;
                uint32_t count = c.ref < 0 ? leaf_count(c.ref) : dp[c.ref].count;
                nd.tri_present |= ((1u << count) - 1u) << (2 * s8);
                uint32_t new_first = (uint32_t)new_order.size();
                for (uint32_t i = 0; i < count; i++) new_order.push_back(first + i);
                shift_leaves(c.parent2, c.which, c.ref, (int64_t)new_first - (int64_t)first);   // keep the binary tree valid
            }
        }
        out->nodes8[w.n8] = nd;
    }
    std::vector<uint32_t> s2t(tri_count);
    for (uint32_t i = 0; i < tri_count; i++) s2t[i] = out->slot_to_tri[new_order[i]];
    out->slot_to_tri.swap(s2t);
}

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
    out->tri_lift.assign(tri_count, 0.0f);
    Box sb;
    sb.reset();
    double lift = 0;
    // absolute part of the box padding: the wide traversal evaluates slab distances as one fused
    // multiply-add per plane (er_wavefront.hip), whose rounding error is bounded by ~3e-7 x the largest
    // coordinate in play; 1e-6 x scene scale keeps every box conservative under that arithmetic
    float vmax = 0;
    for (size_t i = 0; i < (size_t)tri_count * 9; i++) vmax = std::max(vmax, std::fabs(vertices[i]));
    const float pad_abs = vmax * 1e-6f;
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
            float pad = std::max(m * 4e-7f + 1e-37f, pad_abs);
            p.lo[a] = b.lo[a] - pad;
            p.hi[a] = b.hi[a] + pad;
            p.c[a] = (v[a] + v[3 + a] + v[6 + a]) * (1.0f / 3.0f);
        }
        p.id = i;
        sb.grow(p.lo, p.hi);
        // bound on |shadingPosition - geomPosition| (src/Tri.h:106-112): the hit point is a convex
        // combination of the vertices and each p_j moves it by |dot(P - v_j, n_j)| * |n_j|
        const float* nn = normals + (size_t)i * 9;
        double tl = 0;
        for (int j = 0; j < 3; j++) {
            double nx = nn[3 * j], ny = nn[3 * j + 1], nz = nn[3 * j + 2];
            double nl = std::sqrt(nx * nx + ny * ny + nz * nz);
            for (int k = 0; k < 3; k++) {
                if (k == j) continue;
                double dx = (double)v[3 * k] - v[3 * j], dy = (double)v[3 * k + 1] - v[3 * j + 1], dz = (double)v[3 * k + 2] - v[3 * j + 2];
                double l = std::fabs(dx * nx + dy * ny + dz * nz) * nl;
                if (l > tl) tl = l;
            }
        }
        if (tl > lift) lift = tl;
        out->tri_lift[i] = (float)(tl * 1.01) + 1e-30f;
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
    er_collapse_bvh8(out);
    out->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
