// er_wavefront.hip -- wavefront formulation of the per-sample path for gfx950.
//
// Same arithmetic per pixel as er_render_kernel (er_kernels.hip) and therefore as
// renderingKernel (reference src/kernel.cpp:477-646); what changes is the schedule:
//
//   * every owned pixel is a SLOT whose path state lives in HBM (SoA, 16-byte records);
//   * one ITERATION = [er_wf_trace: all pending rays] -> [er_wf_shade: one bounce-loop step per
//     active slot].  Rays are compacted into queues with wave-aggregated atomics (ballot +
//     popcount, one atomic per wave), closest-hit and shadow rays are kept in separate,
//     type-uniform waves, so every traversal wave starts full;
//   * the trace kernel is persistent: a fixed grid of waves pulls 64-ray tickets until the
//     queues are drained.  It holds only ray + traversal state (low VGPR count -> many waves per
//     SIMD to hide the dependent node/triangle fetches) and a per-lane stack in LDS;
//   * a path that ends is finalised and its pixel's next sample starts in the same shade step
//     (a pixel's samples form one RNG stream, so they run back to back in its slot).
//
// The shadow query's outcome only selects which of two precomputed contributions is added to
// the path's radiance (c_vis / c_occ), so shading never waits for it: the next shade step of
// the slot resolves it first.  A path that ends while its shadow ray is in flight is re-queued
// once with ER_WF_FINALIZE_ONLY.
#include "er_device.h"
#include "er_kernels.h"
#include "er_wavefront.h"

using namespace erd;

__device__ __forceinline__ unsigned wave_sum_u(unsigned v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// wave-aggregated queue append: returns this lane's position (valid only if `want`)
__device__ __forceinline__ unsigned queue_push(uint32_t* counter, bool want) {
    unsigned long long mask = __ballot(want);
    if (mask == 0) return 0;
    unsigned n = __popcll(mask);
    unsigned lane = threadIdx.x & 63;
    unsigned leader = __ffsll((long long)mask) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(counter, n);
    base = __shfl(base, leader, 64);
    unsigned rank = __popcll(mask & ((1ull << lane) - 1ull));
    return base + rank;
}

__device__ __forceinline__ void slot_pixel(const DevScene& S, uint32_t slot, uint32_t& px, uint32_t& py) {
    uint32_t tile = S.owned_tiles[slot >> 6];
    uint32_t lane = slot & 63;
    uint32_t tx = tile % S.tiles_x, ty = tile / S.tiles_x;
    px = tx * ER_TILE + (lane & 7);
    py = ty * ER_TILE + (lane >> 3);
}

// exact reference metric |Hit.position - origin| of triangle `tslot` for `ray` (inf if the ray misses it)
__device__ __forceinline__ float exact_distance(const DevScene& S, uint32_t tslot, const Ray& ray) {
    F3 v0, v1, v2;
    float4 qa, qb, qc;
    load_verts(S, tslot, v0, v1, v2, qa, qb, qc);
    float u, v, t;
    if (!tri_mt(v0, v1, v2, ray, u, v, t)) return __builtin_inff();
    return candidate_distance(S, tslot, v0, v1, v2, ray, u, v, t);
}

// packed per-slot flags in reduc.w: bounce (bits 0-15) | pending shadow (bit 16)
#define WF_PENDING 0x10000u

// ---- begin: first camera ray of every slot, queue 0 = all valid slots ----
__global__ __launch_bounds__(64) void er_wf_begin(DevScene S, WfState W, uint32_t n_samples) {
    uint32_t slot = blockIdx.x * 64 + threadIdx.x;
    uint32_t px, py;   // (all counters were zeroed by the memset that precedes this launch)
    slot_pixel(S, slot, px, py);
    bool valid = px < S.x_res && py < S.y_res && n_samples > 0;
    if (valid) {
        uint32_t idx = py * S.x_res + px;
        uint32_t rs = S.rng[idx];
        float c1 = rng_next(rs), c2 = rng_next(rs), c3 = rng_next(rs), c4 = rng_next(rs), c5 = rng_next(rs);
        Ray ray = camera_ray(S.cam, (int)px, (int)py, S.x_res, S.y_res, c1, c2, c3, c4, c5);
        W.ray_o[slot] = make_float4(ray.o.x, ray.o.y, ray.o.z, 0.0f);
        W.ray_d[slot] = make_float4(ray.d.x, ray.d.y, ray.d.z, 0.0f);
        W.light[slot] = make_float4(0.0f, 0.0f, 0.0f, __builtin_bit_cast(float, rs));
        W.reduc[slot] = make_float4(1.0f, 1.0f, 1.0f, __builtin_bit_cast(float, 0u));
        W.aov_n[slot] = make_float4(0, 0, 0, 0);
        W.aov_t[slot] = make_float4(0, 0, 0, 0);
        W.aov_b[slot] = make_float4(0, 0, 0, 0);
        W.left[slot] = n_samples;
    }
    unsigned pos = queue_push(&W.counts[WF_NC], valid);
    if (valid) W.q[0][pos] = slot;
}

// ---- trace: persistent waves, per-lane ray refill, 8-wide compressed BVH ----
//
// Every lane owns one ray at a time and advances it by ONE step per loop iteration: either a
// NODE step (fetch one 80-byte ErNode8, decode and box-test its eight children) or a TRIANGLE
// step (Moller-Trumbore on one or two adjacent 48-byte records).  Both read 96 bytes from one
// address, so the wave issues a single batch of six 16-byte loads per iteration whatever mix of
// states its lanes are in.  A lane whose ray is done writes the result and, once enough lanes
// are idle, the wave hands them new rays from its local chunk of the queue (one global atomic
// per chunk) -- lanes never wait for the slowest ray of a 64-ray batch.
//
// Traversal state per lane (after Ylitie et al. 2017): the current NODE GROUP (first-child
// index + mask of hit inner children, stored at bit `slot ^ octant` so the highest set bit is
// the nearest child) and the current TRIANGLE GROUP (first slot + mask).  Only node groups are
// pushed, at most one per level, so the 32-entry LDS stack is bounded by the tree depth.
//
// Nearest hit under the reference's metric m = |Hit.position - origin| (src/BVH.cpp:114) without
// fetching normals: for a triangle with lift bound l (er_bvh.h) a Moller-Trumbore hit at
// parameter t has m in [t - l - eps, t + l + eps].  The kernel keeps U = the smallest upper
// bound seen and the (at most two) candidates whose lower bound is <= U; almost always one
// survives and it is the reference's winner.  Two survivors -> the shade step compares their
// exact metrics; more -> the shade step re-traces that ray with the exact scalar routine.
#ifndef WF_REFILL_MIN
#define WF_REFILL_MIN 16
#endif
#define WF_LDS_STACK 8
#ifndef WF_COOP
#define WF_COOP 0   // cooperative LDS-DMA fetch: bit-exact, measured 12 % slower than the per-lane fetch
#endif

__device__ __forceinline__ float ubyte_f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }

template <bool COUNT>
__global__ __launch_bounds__(64) void er_wf_trace(DevScene S, WfState W, uint32_t parity) {
    // group stack: the first WF_LDS_STACK levels in LDS, deeper levels (never reached by SAH trees of the
    // benchmark scenes: 1M triangles -> depth 7) in a per-wave HBM spill area, so depth stays unbounded
    __shared__ uint2 s_stack[WF_LDS_STACK * 64];
#if WF_COOP
    // staging area of the cooperative fetch: pieces 0-3 of lane n at [4n + j], pieces 4-5 at [256 + 2n + j]
    __shared__ float4 s_stage[6 * 64];
    const uint32_t lds_stage = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)s_stage);
#endif
    const int lane = threadIdx.x;
    uint2* stack = s_stack + lane;
    uint2* spill = W.spill + (size_t)blockIdx.x * (ER_STACK * 64) + lane;
    const uint32_t nC = W.counts[WF_NC + WF_PAR(parity)], nS = W.counts[WF_NS + WF_PAR(parity)];
    if (blockIdx.x == 0 && lane == 0) {   // reset what the NEXT shade step appends to / pulls from
        W.counts[WF_NC + WF_PAR(parity ^ 1)] = 0;
        W.counts[WF_NS + WF_PAR(parity ^ 1)] = 0;
        W.counts[WF_TS] = 0;
    }
    const uint32_t total = nC + nS;
    const uint32_t* qc = W.q[parity];
    const uint32_t* qs = W.qs[parity];
    unsigned c_rays = 0, c_nodes = 0, c_tris = 0;

    // per-lane ray state
    bool busy = false, shadow = false;
    uint32_t entry = 0;
    F3 o = f3s(0), d = f3s(0), idir = f3s(0), noi = f3s(0);   // noi = -(o * idir)
    float U = 0, limit = 0;          // closest: smallest upper bound so far; shadow: exact distance of the self hit
    int s0 = -1, s1 = -1;            // surviving candidates
    float lo0 = 0, lo1 = 0;
    bool overflow = false;
    int skip = -1;
    uint32_t ng_base = 0, ng_bits = 0;   // node group: first child index; hit mask (bits 0-7, octant order) | imask << 8
    uint32_t tg_base = 0, tg_mask = 0;   // triangle group: first slot; mask of slots still to test
    uint32_t oct7 = 0;
    int sp = 0;
    bool exhausted = total == 0;
    // rays are handed out in chunks: big enough that the single queue-head word is not the bottleneck
    // (one word saturates near 90 atomics/us), small enough to keep every wave busy on short queues
    unsigned chunk = total / (gridDim.x * 4u);
    chunk = chunk < 64u ? 64u : (chunk > 512u ? 512u : chunk);
    chunk = (chunk + 63u) & ~63u;
    unsigned pool_cur = 0, pool_end = 0;
    // HW_REG_XCC_ID (hwreg 20, bits 3:0): which XCD this wave runs on
    const unsigned xcd = (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;
    unsigned steal = 0, cLo = 0, cLen = 0, sLo = 0;

    while (true) {
        // ---- refill idle lanes from the wave's local chunk; one global atomic per chunk ----
        unsigned long long idle = __ballot(!busy);
        unsigned n_idle = __popcll(idle);
        if (!exhausted && (n_idle >= WF_REFILL_MIN || n_idle == 64)) {
            if (pool_cur >= pool_end) {
                // XCD-aware hand-out: the queue is cut into 8 contiguous ranges, one per XCD (each XCD has its own
                // L2).  The queue is ordered by pixel tile, so a range is a band of the image and its rays touch a
                // slab of the scene; a wave drains its own XCD's range first and then steals from the others.
                // (Placement changes speed only: any wave may process any ray.)
                while (true) {
                    const unsigned r = (xcd + steal) & 7u;
                    // range r = its eighth of the closest-hit queue followed by its eighth of the shadow queue
                    cLo = (unsigned)(((unsigned long long)nC * r) >> 3);
                    cLen = (unsigned)(((unsigned long long)nC * (r + 1)) >> 3) - cLo;
                    sLo = (unsigned)(((unsigned long long)nS * r) >> 3);
                    const unsigned r_len = cLen + (unsigned)(((unsigned long long)nS * (r + 1)) >> 3) - sLo;
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(&W.counts[WF_TT + r * WF_LINE], chunk);
                    base = __shfl(base, 0, 64);
                    if (base < r_len) {
                        pool_cur = base;
                        pool_end = base + chunk < r_len ? base + chunk : r_len;
                        break;
                    }
                    steal++;
                    if (steal == 8) { exhausted = true; pool_end = pool_cur; break; }
                }
            }
            if (!busy) {
                unsigned item = pool_cur + __popcll(idle & ((1ull << lane) - 1ull));
                if (item < pool_end) {
                    shadow = item >= cLen;
                    entry = shadow ? qs[sLo + item - cLen] : qc[cLo + item];
                    if (!(entry & ER_WF_FINALIZE_ONLY)) {
                        float4 ro = shadow ? W.sh_o[entry] : W.ray_o[entry];
                        float4 rd = shadow ? W.sh_d[entry] : W.ray_d[entry];
                        o = f3(ro.x, ro.y, ro.z);
                        d = f3(rd.x, rd.y, rd.z);
                        // 1/d clamped to +-1e18: a zero (or denormal) component would make the fused plane
                        // distances inf - inf = NaN; with 1e18 the ray stays inside its slab for any finite t
                        idir = f3(clampf(1.0f / d.x, -1e18f, 1e18f), clampf(1.0f / d.y, -1e18f, 1e18f), clampf(1.0f / d.z, -1e18f, 1e18f));
                        noi = f3(-(o.x * idir.x), -(o.y * idir.y), -(o.z * idir.z));
                        skip = shadow ? __builtin_bit_cast(int, ro.w) : -1;
                        limit = shadow ? rd.w : __builtin_inff();
                        U = limit;
                        s0 = -1; s1 = -1; overflow = false;
                        sp = 0;
                        // a positive direction visits low-coordinate children first: they get the high bits
                        oct7 = (idir.x >= 0.0f ? 1u : 0u) | (idir.y >= 0.0f ? 2u : 0u) | (idir.z >= 0.0f ? 4u : 0u);
                        ng_base = 0;
                        ng_bits = (1u << oct7) | (1u << 8);      // the root: slot 0 of a virtual parent, an inner child
                        tg_base = 0; tg_mask = 0;
                        c_rays++;
                        busy = S.node_count != 0;
                        if (!busy) {   // empty scene: every ray misses
                            if (shadow) W.occluded[entry] = 0; else { W.hit[entry] = -1; W.hit2[entry] = -1; }
                        }
                    }
                }
            }
            pool_cur = pool_cur + n_idle < pool_end ? pool_cur + n_idle : pool_end;
        }
        if (__ballot(busy) == 0) {
            if (exhausted) break;
            continue;
        }
        // ---- phase 1 (per lane): pop if nothing is pending, then choose this iteration's step ----
        bool finished = false, do_step = false, tri_step = false, two = false;
        uint32_t tslot = 0, off = 0;      // off: what to fetch, in 16-byte pieces from S.nodes8 (nodes and triangles share one buffer)
        if (busy) {
            if (tg_mask == 0 && (ng_bits & 0xffu) == 0) {
                if (sp == 0) {
                    finished = true;
                    if (shadow) W.occluded[entry] = overflow ? 3 : (s0 >= 0 ? 2 : 0);
                } else {
                    sp--;
                    uint2 g = sp < WF_LDS_STACK ? stack[sp * 64] : spill[(sp - WF_LDS_STACK) * 64];
                    ng_base = g.x;
                    ng_bits = g.y;
                }
            }
            if (!finished) {
                do_step = true;
                tri_step = tg_mask != 0;
                if (tri_step) {
                    unsigned i = __ffs(tg_mask) - 1;
                    two = ((tg_mask >> i) & 2u) != 0;
                    tg_mask &= ~((two ? 3u : 1u) << i);
                    tslot = tg_base + i;
                    off = S.tri_base_pieces + tslot * 3u;
                } else {
                    uint32_t nmask = ng_bits & 0xffu, imask = (ng_bits >> 8) & 0xffu;
                    unsigned b = 31 - __clz(nmask);
                    nmask &= ~(1u << b);
                    unsigned s8 = b ^ oct7;
                    uint32_t child = ng_base + __popc(imask & ((1u << s8) - 1u));
                    if (nmask) {                       // siblings still to visit: one stack entry for the whole group
                        uint2 g = make_uint2(ng_base, nmask | (imask << 8));
                        if (sp < WF_LDS_STACK) stack[sp * 64] = g; else spill[(sp - WF_LDS_STACK) * 64] = g;
                        sp++;
                    }
                    ng_bits = 0;
                    off = child * 5u;
                }
            }
        }
        // ---- phase 2 (whole wave): fetch up to 96 bytes per stepping lane ----
        // The vector-memory pipeline (TA/TD/TCP) is the busiest unit of this kernel, and it pays per cache
        // access, not per byte: a lane gathering six 16-byte pieces from its own line costs six accesses.
        float4 a, b4, c, dd, e4, f4;
#if WF_COOP
        {
            // Cooperative fetch: lanes exchange their offsets (ds_bpermute) so that FOUR adjacent lanes read the
            // 64 contiguous bytes of one lane's record (pieces 0-3) and TWO adjacent lanes read its pieces 4-5;
            // adjacent lanes on one line coalesce into a single cache access (16 resp. 32 accesses per load
            // instead of 64).  The loads are LDS-DMA (global_load_lds_dwordx4: lane l writes 16 bytes at
            // M0 + 16 l), so load k of the first four lands the records of lanes 16k..16k+15 at s_stage[4n + j];
            // every lane then reads its own 96 bytes back with six ds_read_b128.
            const bool need34 = do_step && (!tri_step || two), need5 = do_step && tri_step && two;
            const uint32_t offp = (do_step ? off : 0u) | (need34 ? 0x80000000u : 0u) | (need5 ? 0x40000000u : 0u);
            const float4* g[6];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((16 * k + (lane >> 2)) * 4, (int)offp);
                g[k] = S.nodes8 + (v & 0x3fffffffu) + (lane & 3);
            }
#pragma unroll
            for (int k = 0; k < 2; k++) {
                uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((32 * k + (lane >> 1)) * 4, (int)offp);
                const bool needed = (lane & 1) ? (v & 0x40000000u) != 0 : (v & 0x80000000u) != 0;
                g[4 + k] = S.nodes8 + (needed ? (v & 0x3fffffffu) : 0u) + 4 + (lane & 1);   // unneeded -> one shared line
            }
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\t"
                         "s_mov_b32 m0, %7\n\t" "s_nop 0\n\t" "global_load_lds_dwordx4 %1, off\n\t"
                         "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t" "global_load_lds_dwordx4 %2, off\n\t"
                         "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t" "global_load_lds_dwordx4 %3, off\n\t"
                         "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t" "global_load_lds_dwordx4 %4, off\n\t"
                         "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t" "global_load_lds_dwordx4 %5, off\n\t"
                         "s_add_u32 m0, m0, 0x400\n\t" "s_nop 0\n\t" "global_load_lds_dwordx4 %6, off\n\t"
                         "s_mov_b32 m0, %0\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&s"(keep)
                         : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "s"(lds_stage)
                         : "memory", "scc");
            const float4* mine = s_stage + lane * 4;
            a = mine[0]; b4 = mine[1]; c = mine[2]; dd = mine[3];
            const float4* mine2 = s_stage + 256 + lane * 2;
            e4 = mine2[0]; f4 = mine2[1];
        }
#else
        {
            // Per-lane fetch: whole dwordx4 pieces in ONE asm statement with their wait (the compiler treats asm
            // outputs as ready when the statement ends); lanes that do not need a piece read one shared address.
            const float4* p = S.nodes8 + (do_step ? off : 0u);
            const float4* p34 = (do_step && (!tri_step || two)) ? p : S.nodes8;
            const float4* p5 = (do_step && tri_step && two) ? p : S.nodes8;
            asm volatile("global_load_dwordx4 %0, %6, off\n\t"
                         "global_load_dwordx4 %1, %6, off offset:16\n\t"
                         "global_load_dwordx4 %2, %6, off offset:32\n\t"
                         "global_load_dwordx4 %3, %7, off offset:48\n\t"
                         "global_load_dwordx4 %4, %7, off offset:64\n\t"
                         "global_load_dwordx4 %5, %8, off offset:80\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(a), "=&v"(b4), "=&v"(c), "=&v"(dd), "=&v"(e4), "=&v"(f4)
                         : "v"(p), "v"(p34), "v"(p5)
                         : "memory");
        }
#endif
        // ---- phase 3 (per lane): the step itself ----
        if (busy) {
            if (do_step) {
                const float eps_far = (S.scene_scale + (U < 3.0e38f ? U : 0.0f)) * 4e-6f;
                const float bound = U + S.max_lift + eps_far;
                if (tri_step) {
                    // Both records are tested with straight-line code (rejections folded into one predicate, exactly
                    // the comparisons of Tri::hit, src/Tri.h:56-77), then the interval bookkeeping runs once per record.
                    if (COUNT) c_tris += two ? 2u : 1u;
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const uint32_t slot = tslot + k;
                        const F3 v0 = k == 0 ? f3(a.x, a.y, a.z) : f3(dd.x, dd.y, dd.z);
                        const F3 v1 = k == 0 ? f3(b4.x, b4.y, b4.z) : f3(e4.x, e4.y, e4.z);
                        const F3 v2 = k == 0 ? f3(c.x, c.y, c.z) : f3(f4.x, f4.y, f4.z);
                        const float lift = k == 0 ? b4.w : e4.w;
                        const float EPSILON = 0.0000001f;
                        const F3 edge1 = v1 - v0, edge2 = v2 - v0;
                        const F3 pvec = cross(d, edge2);
                        const float det = dot(edge1, pvec);
                        const float inv_det = 1.0f / det;
                        const F3 tvec = o - v0;
                        const float u = dot(tvec, pvec) * inv_det;
                        const F3 qvec = cross(tvec, edge1);
                        const float v = dot(d, qvec) * inv_det;
                        const float t = dot(edge2, qvec) * inv_det;
                        const bool rejected = (det > -EPSILON && det < EPSILON) || (u < 0 || u > 1) || (v < 0 || (u + v) > 1) || (t < 0);
                        const bool valid = !rejected && (k == 0 || two) && (int)slot != skip && !finished;
                        const float eps = (S.scene_scale + t) * 4e-6f;
                        const float lo = t - lift - eps, hi = t + lift + eps;
                        // shadow query: certainly nearer than the self hit -> occluded; inside the interval -> ambiguous
                        const bool occl = valid && shadow && hi < limit;
                        const bool amb = valid && shadow && !(hi < limit) && lo < limit;
                        // closest query: survives if its lower bound does not exceed the smallest upper bound so far
                        const bool cand = valid && !shadow && !(lo > U);
                        U = (cand && hi < U) ? hi : U;
                        s0 = (cand && s0 >= 0 && lo0 > U) ? -1 : s0;
                        s1 = (cand && s1 >= 0 && lo1 > U) ? -1 : s1;
                        const bool want = cand || amb;
                        const bool ins0 = want && s0 < 0;
                        const bool ins1 = want && !ins0 && s1 < 0;
                        overflow = overflow || (want && !ins0 && !ins1);
                        s0 = ins0 ? (int)slot : s0;
                        lo0 = ins0 ? lo : lo0;
                        s1 = ins1 ? (int)slot : s1;
                        lo1 = ins1 ? lo : lo1;
                        if (occl) { W.occluded[entry] = 1; finished = true; }
                    }
                } else {
                    if (COUNT) c_nodes++;
                    const uint32_t ebits = __builtin_bit_cast(uint32_t, a.w);
                    const float sx = __builtin_bit_cast(float, (ebits & 0xffu) << 23);
                    const float sy = __builtin_bit_cast(float, ((ebits >> 8) & 0xffu) << 23);
                    const float sz = __builtin_bit_cast(float, ((ebits >> 16) & 0xffu) << 23);
                    const uint32_t imask = ebits >> 24;
                    const uint32_t meta_w[2] = {__builtin_bit_cast(uint32_t, b4.z), __builtin_bit_cast(uint32_t, b4.w)};
                    const uint32_t qlx[2] = {__builtin_bit_cast(uint32_t, c.x), __builtin_bit_cast(uint32_t, c.y)};
                    const uint32_t qly[2] = {__builtin_bit_cast(uint32_t, c.z), __builtin_bit_cast(uint32_t, c.w)};
                    const uint32_t qlz[2] = {__builtin_bit_cast(uint32_t, dd.x), __builtin_bit_cast(uint32_t, dd.y)};
                    const uint32_t qhx[2] = {__builtin_bit_cast(uint32_t, dd.z), __builtin_bit_cast(uint32_t, dd.w)};
                    const uint32_t qhy[2] = {__builtin_bit_cast(uint32_t, e4.x), __builtin_bit_cast(uint32_t, e4.y)};
                    const uint32_t qhz[2] = {__builtin_bit_cast(uint32_t, e4.z), __builtin_bit_cast(uint32_t, e4.w)};
                    // Slab test of the eight children.  Box tests only gate the traversal, so any conservative
                    // evaluation is allowed: the entry/exit planes per axis are picked by the ray's direction sign
                    // and each plane distance is ONE fused multiply-add, t = q * (2^e * idir) + (p * idir - o * idir);
                    // its rounding error is covered by the absolute box padding of the builder (er_bvh.cpp).
                    // (1/d is clamped at ray setup, so no plane distance is NaN for finite inputs.)
                    const float Ax = sx * idir.x, Ay = sy * idir.y, Az = sz * idir.z;
                    const float Bx = __builtin_fmaf(a.x, idir.x, noi.x), By = __builtin_fmaf(a.y, idir.y, noi.y), Bz = __builtin_fmaf(a.z, idir.z, noi.z);
                    const bool posx = (oct7 & 1u) != 0, posy = (oct7 & 2u) != 0, posz = (oct7 & 4u) != 0;
                    const uint32_t nx[2] = {posx ? qlx[0] : qhx[0], posx ? qlx[1] : qhx[1]}, fx[2] = {posx ? qhx[0] : qlx[0], posx ? qhx[1] : qlx[1]};
                    const uint32_t ny[2] = {posy ? qly[0] : qhy[0], posy ? qly[1] : qhy[1]}, fy[2] = {posy ? qhy[0] : qly[0], posy ? qhy[1] : qly[1]};
                    const uint32_t nz[2] = {posz ? qlz[0] : qhz[0], posz ? qlz[1] : qhz[1]}, fz[2] = {posz ? qhz[0] : qlz[0], posz ? qhz[1] : qlz[1]};
                    uint32_t hits = 0, tmask = 0;
#pragma unroll
                    for (int s8 = 0; s8 < 8; s8++) {
                        const int w = s8 >> 2, k = s8 & 3;
                        const uint32_t meta = (meta_w[w] >> (8 * k)) & 0xffu;
                        const float tnx = __builtin_fmaf(ubyte_f(nx[w], k), Ax, Bx), tfx = __builtin_fmaf(ubyte_f(fx[w], k), Ax, Bx);
                        const float tny = __builtin_fmaf(ubyte_f(ny[w], k), Ay, By), tfy = __builtin_fmaf(ubyte_f(fy[w], k), Ay, By);
                        const float tnz = __builtin_fmaf(ubyte_f(nz[w], k), Az, Bz), tfz = __builtin_fmaf(ubyte_f(fz[w], k), Az, Bz);
                        const float tmin = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);
                        const float tmax = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
                        const bool hit = (tmin <= tmax) && (tmax >= 0.0f) && (tmin <= bound);
                        // meta: empty 0 and inner 1 have a zero triangle count, so they add no triangle bits; empty
                        // slots are not in imask, so they add no node bit either -- no branch on the child kind
                        const uint32_t leafbits = ((1u << (meta >> 5)) - 1u) << (meta & 31u);
                        hits |= hit ? (1u << s8) : 0u;
                        tmask |= hit ? leafbits : 0u;
                    }
                    // inner hits, moved from bit `slot` to bit `slot ^ oct7` (three conditional swap stages)
                    uint32_t nmask = hits & imask;
                    nmask = (oct7 & 1u) ? (((nmask & 0xAAu) >> 1) | ((nmask & 0x55u) << 1)) : nmask;
                    nmask = (oct7 & 2u) ? (((nmask & 0xCCu) >> 2) | ((nmask & 0x33u) << 2)) : nmask;
                    nmask = (oct7 & 4u) ? (((nmask & 0xF0u) >> 4) | ((nmask & 0x0Fu) << 4)) : nmask;
                    ng_base = __builtin_bit_cast(uint32_t, b4.x);
                    ng_bits = nmask | (imask << 8);
                    tg_base = __builtin_bit_cast(uint32_t, b4.y);
                    tg_mask = tmask;
                }
            }
            if (finished) {
                if (shadow) {
                    W.occ_a[entry] = s0;
                    W.occ_b[entry] = s1;
                } else {
                    W.hit[entry] = s0 >= 0 ? s0 : s1;
                    W.hit2[entry] = overflow ? -2 : ((s0 >= 0 && s1 >= 0) ? s1 : -1);
                }
                busy = false;
            }
        }
    }
    unsigned t0 = wave_sum_u(c_rays);
    unsigned t1 = 0, t2 = 0;
    if (COUNT) { t1 = wave_sum_u(c_nodes); t2 = wave_sum_u(c_tris); }
    if (lane == 0 && t0) {
        atomicAdd(&S.counters->rays, (unsigned long long)t0);
        if (COUNT) {
            atomicAdd(&S.counters->node_visits, (unsigned long long)t1);
            atomicAdd(&S.counters->tri_tests, (unsigned long long)t2);
        }
    }
}

// ---- shade: one bounce-loop step per active slot (src/kernel.cpp:508-645) ----
#ifndef WF_SHADE_WAVES
#define WF_SHADE_WAVES 2
#endif
template <bool COUNT>
__global__ __launch_bounds__(64, WF_SHADE_WAVES) void er_wf_shade(DevScene S, WfState W, uint32_t parity) {
    __shared__ int s_stack[ER_STACK * 64];   // only for the rare exact re-trace of an overflowed ray
    const int lane = threadIdx.x;
    int* stack = s_stack + lane;
    unsigned c_nodes = 0, c_tris = 0;
    const uint32_t nC = W.counts[WF_NC + WF_PAR(parity)];
    if (blockIdx.x == 0 && lane < 8) W.counts[WF_TT + lane * WF_LINE] = 0;
    const uint32_t wC = (nC + 63) >> 6;
    const uint32_t* qc = W.q[parity];
    uint32_t* qn = W.q[parity ^ 1];
    uint32_t* qsn = W.qs[parity ^ 1];
    const size_t npx = (size_t)S.x_res * S.y_res;
    const int hw = S.hdri_tex.width, hh = S.hdri_tex.height;
    unsigned c_paths = 0, c_bounce = 0, c_shaded = 0, c_texels = 0, c_hdri = 0;
    // tickets (64 slots each) are taken several at a time: one atomic per chunk
    uint32_t tchunk = wC / (gridDim.x * 4u);
    tchunk = tchunk < 1u ? 1u : (tchunk > 8u ? 8u : tchunk);
    uint32_t t_cur = 0, t_end = 0;
    while (true) {
        if (t_cur >= t_end) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&W.counts[WF_TS], tchunk);
            base = __shfl(base, 0, 64);
            t_cur = base;
            t_end = base + tchunk < wC ? base + tchunk : wC;
            if (base >= wC) break;
        }
        uint32_t ticket = t_cur++;
        uint32_t item = ticket * 64 + lane;
        bool active = item < nC;
        bool push_closest = false, push_shadow = false;
        uint32_t next_entry = 0, slot = 0;
        if (active) {
            uint32_t e = qc[item];
            slot = e & ~ER_WF_FINALIZE_ONLY;
            bool fin_only = (e & ER_WF_FINALIZE_ONLY) != 0;
            uint32_t px, py;
            slot_pixel(S, slot, px, py);
            const uint32_t idx = py * S.x_res + px;
            float4 L4 = W.light[slot], R4 = W.reduc[slot];
            F3 light = f3(L4.x, L4.y, L4.z), reduction = f3(R4.x, R4.y, R4.z);
            uint32_t rs = __builtin_bit_cast(uint32_t, L4.w);
            uint32_t packed = __builtin_bit_cast(uint32_t, R4.w);
            uint32_t bounce = packed & 0xFFFFu;
            if (packed & WF_PENDING) {   // resolve the previous bounce's shadow query
                int occ = W.occluded[slot];
                if (occ >= 2) {          // the trace kernel could not decide from t-intervals: exact metric
                    float4 so = W.sh_o[slot], sd = W.sh_d[slot];
                    Ray sr;
                    sr.o = f3(so.x, so.y, so.z);
                    sr.d = f3(sd.x, sd.y, sd.z);
                    if (occ == 3) {
                        float dd;
                        occ = trace<COUNT, true>(S, stack, sr, __builtin_bit_cast(int, so.w), sd.w, dd, c_nodes, c_tris) >= 0 ? 1 : 0;
                    } else {
                        int ca = W.occ_a[slot], cb = W.occ_b[slot];
                        bool nearer = exact_distance(S, (uint32_t)ca, sr) < sd.w;
                        if (cb >= 0) nearer = nearer || (exact_distance(S, (uint32_t)cb, sr) < sd.w);
                        occ = nearer ? 1 : 0;
                    }
                }
                float4 c = occ ? W.c_occ[slot] : W.c_vis[slot];
                light = light + f3(c.x, c.y, c.z);
            }
            bool pending = false;
            bool done = fin_only;
            Ray ray;
            if (!fin_only) {
                float4 o = W.ray_o[slot], d = W.ray_d[slot];
                ray.o = f3(o.x, o.y, o.z);
                ray.d = f3(d.x, d.y, d.z);
                int hslot = W.hit[slot];
                int h2 = W.hit2[slot];
                if (h2 == -2) {          // more than two candidates inside one t-interval: exact scalar traversal
                    float dd;
                    hslot = trace<COUNT, false>(S, stack, ray, -1, __builtin_inff(), dd, c_nodes, c_tris);
                } else if (h2 >= 0) {    // two candidates: the reference's strict '<' on the exact metric
                    if (exact_distance(S, (uint32_t)h2, ray) < exact_distance(S, (uint32_t)hslot, ray)) hslot = h2;
                }
                c_bounce++;
                if (hslot < 0) {
                    float u, v;
                    spherical_mapping(-1 * ray.d, u, v);
                    light = light + reduction * tex_filtered(S, S.hdri_tex, u, v);
                    if (COUNT) c_texels++;
                    done = true;
                } else {
                    c_shaded++;
                    HitFull hit;
                    full_hit(S, (uint32_t)hslot, ray, hit);
                    const ErMaterial& mat = S.materials[hit.material];
                    HitData hd;
                    generate_hit_data<COUNT>(S, mat, hit, hd, c_texels);
                    int shader = mat.albedo_shader_id;
                    if (shader != -1) {
                        hd.albedo = f3s(0);
                        if (shader >= 0 && shader < 4) hd.albedo = f3(1, 1, 0);
                    }
                    if (rng_next(rs) <= hd.opacity) {
                        F3 wo = ray.d * -1.0f;
                        F3 N = hd.normal;
                        c_hdri++;
                        int count = er_cdf_search(S.hdri_cdf, hw * hh, S.hdri_guide, S.hdri_buckets, rng_next(rs));   // == HDRI::binarySearch
                        float tcx = (float)(count % hw), tcy = (float)(count / hw);
                        float d1 = rng_next(rs), d2 = rng_next(rs), d3 = rng_next(rs);
                        F3 wibrdf = DisneySample(hd, wo, N, d1, d2, d3);
                        float nu = tcx / (float)hw, nv = tcy / (float)hh;
                        float iu, iv;
                        inverse_transform_uv(S.hdri_tex, nu, nv, iu, iv);
                        F3 wihdri = normalized(reverse_spherical_mapping(iu, iv)) * -1.0f;
                        F3 hdriValue = tex_uv(S, S.hdri_tex, iu, iv);
                        if (COUNT) c_texels += 2;
                        F3 evalh = DisneyEval(hd, wo, N, wihdri);
                        float hdripdf = hdri_pdf(S, ermath::f2i(iu * hw), ermath::f2i(iv * hh));
                        float absdot = __builtin_fabsf(dot(wihdri, N));
                        F3 c_vis = reduction * (hd.emission + hdriValue * evalh * absdot / hdripdf);
                        if (evalh.x != 0.0f || evalh.y != 0.0f || evalh.z != 0.0f) {
                            // shadow query needed (see er_kernels.hip): occluded iff the closest hit is another triangle
                            F3 c_occ = reduction * (hd.emission + f3s(0) * evalh * absdot / hdripdf);
                            Ray sr = make_ray(hd.position + N * 0.001f, wihdri);
                            F3 v0, v1, v2;
                            float4 qa, qb, qc4;
                            load_verts(S, (uint32_t)hslot, v0, v1, v2, qa, qb, qc4);
                            float su, sv, st, d_self = __builtin_inff();
                            if (tri_mt(v0, v1, v2, sr, su, sv, st)) d_self = candidate_distance(S, (uint32_t)hslot, v0, v1, v2, sr, su, sv, st);
                            W.sh_o[slot] = make_float4(sr.o.x, sr.o.y, sr.o.z, __builtin_bit_cast(float, hslot));
                            W.sh_d[slot] = make_float4(sr.d.x, sr.d.y, sr.d.z, d_self);
                            W.c_vis[slot] = make_float4(c_vis.x, c_vis.y, c_vis.z, 0.0f);
                            W.c_occ[slot] = make_float4(c_occ.x, c_occ.y, c_occ.z, 0.0f);
                            pending = true;
                        } else {
                            light = light + c_vis;
                        }
                        float brdfpdf = DisneyPdf(hd, wo, N, wibrdf);
                        reduction = reduction * (DisneyEval(hd, wo, N, wibrdf) * __builtin_fabsf(dot(wibrdf, N)) / brdfpdf);
                        if (bounce == 0) {
                            W.aov_n[slot] = make_float4(hd.normal.x, hd.normal.y, hd.normal.z, 0.0f);
                            W.aov_t[slot] = make_float4(hd.tangent.x, hd.tangent.y, hd.tangent.z, 0.0f);
                            W.aov_b[slot] = make_float4(hd.bitangent.x, hd.bitangent.y, hd.bitangent.z, 0.0f);
                        }
                        ray = make_ray(hit.position + wibrdf * 0.001f, wibrdf);
                    } else {
                        ray = make_ray(hit.position + ray.d * 0.001f, ray.d);
                    }
                    bounce++;
                    if (bounce >= S.max_bounces) done = true;
                }
            }
            bool alive = true;
            if (done && pending) {
                // the path is over but its last shadow query is in flight: come back once, without a ray
                next_entry = slot | ER_WF_FINALIZE_ONLY;
                push_closest = true;
            } else if (done) {
                // src/kernel.cpp:597-645
                light = f3(clampf(light.x, 0, 10), clampf(light.y, 0, 10), clampf(light.z, 0, 10));
                uint32_t sa = S.samples[idx];
                if (!(light.x != light.x) && !(light.y != light.y) && !(light.z != light.z)) {
                    float k = ((float)sa) / ((float)(sa + 1));
                    float inv = (float)(sa + 1);
                    float4 an = W.aov_n[slot], at = W.aov_t[slot], ab = W.aov_b[slot];
                    const F3 vals[4] = {light, f3(an.x, an.y, an.z), f3(at.x, at.y, at.z), f3(ab.x, ab.y, ab.z)};
                    const int planes[4] = {ER_PASS_BEAUTY, ER_PASS_NORMAL, ER_PASS_TANGENT, ER_PASS_BITANGENT};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        float4* pp = S.passes + (size_t)planes[q] * npx + idx;
                        float4 p = *pp;
                        if (sa > 0) { p.x *= k; p.y *= k; p.z *= k; }
                        p.x += vals[q].x / inv; p.y += vals[q].y / inv; p.z += vals[q].z / inv;
                        *pp = p;
                    }
                    S.samples[idx] = sa + 1;
                }
                S.rng[idx] = rs;
                c_paths++;
                uint32_t left = W.left[slot] - 1;
                W.left[slot] = left;
                if (left > 0) {
                    float c1 = rng_next(rs), c2 = rng_next(rs), c3 = rng_next(rs), c4 = rng_next(rs), c5 = rng_next(rs);
                    ray = camera_ray(S.cam, (int)px, (int)py, S.x_res, S.y_res, c1, c2, c3, c4, c5);
                    light = f3s(0);
                    reduction = f3s(1);
                    bounce = 0;
                    W.aov_n[slot] = make_float4(0, 0, 0, 0);
                    W.aov_t[slot] = make_float4(0, 0, 0, 0);
                    W.aov_b[slot] = make_float4(0, 0, 0, 0);
                    next_entry = slot;
                    push_closest = true;
                } else {
                    alive = false;
                }
            } else {
                next_entry = slot;
                push_closest = true;
            }
            push_shadow = pending;
            if (alive) {
                if (!(next_entry & ER_WF_FINALIZE_ONLY)) {
                    W.ray_o[slot] = make_float4(ray.o.x, ray.o.y, ray.o.z, 0.0f);
                    W.ray_d[slot] = make_float4(ray.d.x, ray.d.y, ray.d.z, 0.0f);
                }
                W.light[slot] = make_float4(light.x, light.y, light.z, __builtin_bit_cast(float, rs));
                W.reduc[slot] = make_float4(reduction.x, reduction.y, reduction.z,
                                            __builtin_bit_cast(float, bounce | (pending ? WF_PENDING : 0u)));
            }
        }
        unsigned pc = queue_push(&W.counts[WF_NC + WF_PAR(parity ^ 1)], push_closest);
        if (push_closest) qn[pc] = next_entry;
        unsigned ps = queue_push(&W.counts[WF_NS + WF_PAR(parity ^ 1)], push_shadow);
        if (push_shadow) qsn[ps] = slot;
    }
    unsigned t0 = wave_sum_u(c_paths), t1 = wave_sum_u(c_bounce), t3 = wave_sum_u(c_shaded), t4 = wave_sum_u(c_hdri);
    unsigned t7 = COUNT ? wave_sum_u(c_texels) : 0;
    if (lane == 0 && t1 + t0) {
        atomicAdd(&S.counters->paths, (unsigned long long)t0);
        atomicAdd(&S.counters->bounce_samples, (unsigned long long)t1);
        atomicAdd(&S.counters->shaded_hits, (unsigned long long)t3);
        atomicAdd(&S.counters->hdri_samples, (unsigned long long)t4);
        if (COUNT) atomicAdd(&S.counters->texel_fetches, (unsigned long long)t7);
    }
}

void er_launch_wf_begin(const DevScene& S, const WfState& W, uint32_t n_samples, hipStream_t stream) {
    if (S.owned_tile_count == 0) return;
    (void)hipMemsetAsync(W.counts, 0, WF_COUNTS * sizeof(uint32_t), stream);
    hipLaunchKernelGGL(er_wf_begin, dim3(S.owned_tile_count), dim3(64), 0, stream, S, W, n_samples);
}
void er_launch_wf_trace(const DevScene& S, const WfState& W, uint32_t parity, bool count, uint32_t blocks, hipStream_t stream) {
    if (S.owned_tile_count == 0) return;
    if (count) hipLaunchKernelGGL(er_wf_trace<true>, dim3(blocks), dim3(64), 0, stream, S, W, parity);
    else hipLaunchKernelGGL(er_wf_trace<false>, dim3(blocks), dim3(64), 0, stream, S, W, parity);
}
void er_launch_wf_shade(const DevScene& S, const WfState& W, uint32_t parity, bool count, uint32_t blocks, hipStream_t stream) {
    if (S.owned_tile_count == 0) return;
    if (count) hipLaunchKernelGGL(er_wf_shade<true>, dim3(blocks), dim3(64), 0, stream, S, W, parity);
    else hipLaunchKernelGGL(er_wf_shade<false>, dim3(blocks), dim3(64), 0, stream, S, W, parity);
}
