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
#include <cstdlib>
#include "er_device.h"
#include "er_kernels.h"
#include "er_wavefront.h"
#include "er_trav.h"
#include "er_shade.h"

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

// append a wave's `n` staged entries (wave-uniform n; all lanes call): one atomic, coalesced copy
#define WF_STAGE 8
__device__ __forceinline__ void queue_flush(uint32_t* counter, uint32_t* queue, const uint32_t* staged, unsigned n) {
    if (n == 0) return;
    const unsigned lane = threadIdx.x & 63;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(counter, n);
    base = __shfl(base, 0, 64);
    for (unsigned i = lane; i < n; i += 64) queue[base + i] = staged[i];
}

__device__ __forceinline__ void slot_pixel(const DevScene& S, uint32_t slot, uint32_t& px, uint32_t& py) {
    uint32_t tile = S.owned_tiles[slot >> 6];
    uint32_t lane = slot & 63;
    uint32_t tx = tile % S.tiles_x, ty = tile / S.tiles_x;
    px = tx * ER_TILE + (lane & 7);
    py = ty * ER_TILE + (lane >> 3);
}

// packed per-slot flags in reduc.w: bounce (bits 0-15) | pending shadow (bit 16)
#define WF_PENDING 0x10000u
#define WF_LPENDING 0x20000u   // ... | pending point-light shadow (bit 17)

// ---- begin: first camera ray of every slot, queue 0 = all valid slots ----
__global__ __launch_bounds__(64) void er_wf_begin(DevScene S, WfState W, uint32_t n_samples) {
    uint32_t slot = (blockIdx.x * W.pools + W.pool) * 64 + threadIdx.x;   // this pool's share of the owned tiles
    uint32_t px, py;   // (all counters were zeroed by the memset that precedes this launch)
    slot_pixel(S, slot, px, py);
    bool valid = px < S.x_res && py < S.y_res && n_samples > 0;
    if (valid) {
        uint32_t idx = py * S.x_res + px;
        uint32_t rs = S.rng[idx];
        float c1 = rng_next(rs), c2 = rng_next(rs), c3 = rng_next(rs), c4 = rng_next(rs), c5 = rng_next(rs);
        Ray ray = camera_ray(S.cam, (int)px, (int)py, S.x_res, S.y_res, c1, c2, c3, c4, c5);
        W.ray_o[slot] = make_float4(ray.o.x, ray.o.y, ray.o.z, 0.0f);
        W.ray_d[slot] = make_float4(ray.d.x, ray.d.y, ray.d.z, -1.0f);   // w: brdfpdf of the last opaque bounce (ER_FLAG_MIS)
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
// Every lane owns one ray at a time and advances it by ONE step per loop iteration (er_trav.h): either a NODE
// step (one 80-byte ErNode8: decode and box-test its eight children) or a TRIANGLE step (Moller-Trumbore on one
// or two adjacent 48-byte records).  Both read up to 96 bytes from one address, so the wave issues a single
// batch of six 16-byte loads per iteration whatever mix of states its lanes are in.  A lane whose ray is done
// writes the result and, once enough lanes are idle, the wave hands them new rays from its local chunk of the
// queue (one global atomic per chunk) -- lanes never wait for the slowest ray of a 64-ray batch.
#define WF_REFILL_MIN 16   // default; ER_TRACE_REFILL_MIN overrides it at run time (tuning knob; flat from 4 to 32)

#ifndef WF_TRACE_WAVES
#define WF_TRACE_WAVES 4
#endif
template <bool COUNT>
__global__ __launch_bounds__(64, WF_TRACE_WAVES) void er_wf_trace(DevScene S, WfState W, uint32_t parity, uint32_t refill_min, uint32_t* ray_log) {
    __shared__ uint2 s_stack[WF_LDS_STACK * 64];
    const int lane = threadIdx.x;
    uint2* stack = s_stack + lane;
    uint2* spill = W.spill + (size_t)blockIdx.x * (ER_STACK8 * 64) + lane;
    const uint32_t nC = W.counts[WF_NC + WF_PAR(parity)], nS = W.counts[WF_NS + WF_PAR(parity)];
    if (blockIdx.x == 0 && lane == 0) {   // reset what the NEXT shade step appends to / pulls from
        W.counts[WF_NC + WF_PAR(parity ^ 1)] = 0;
        W.counts[WF_NS + WF_PAR(parity ^ 1)] = 0;
        W.counts[WF_TS] = 0;
        if (ray_log) *ray_log = nC + nS;
    }
    const uint32_t total = nC + nS;
    const uint32_t* qc = W.q[parity];
    const uint32_t* qs = W.qs[parity];
    unsigned c_rays = 0, c_nodes = 0, c_tris = 0;
    unsigned c_wsteps = 0, c_busy = 0, c_nl = 0, c_tl = 0;   // ER_FLAG_COUNTERS: lane occupancy of this loop (wave-uniform)

    Trav T;
    trav_begin(T, f3s(0), f3(0, 0, 1), false, -1, 0.0f);
    bool busy = false;
    uint32_t entry = 0;
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
        if (!exhausted && (n_idle >= refill_min || n_idle == 64)) {
            if (pool_cur >= pool_end) {
                // XCD-aware hand-out: the queues are cut into 8 contiguous ranges, one per XCD (each XCD has its own
                // L2).  The queues are ordered by pixel tile, so a range is a band of the image; a wave drains its
                // own XCD's range first and then steals from the others.  (Placement changes speed only: any wave
                // may process any ray.  Measured effect on the soup: none -- its rays cross the whole scene.)
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
                    const bool shadow = item >= cLen;
                    entry = shadow ? qs[sLo + item - cLen] : qc[cLo + item];
                    if (!(entry & ER_WF_FINALIZE_ONLY)) {
                        float4 ro = shadow ? W.sh_o[entry] : W.ray_o[entry];
                        float4 rd = shadow ? W.sh_d[entry] : W.ray_d[entry];
                        trav_begin(T, f3(ro.x, ro.y, ro.z), f3(rd.x, rd.y, rd.z), shadow, shadow ? __builtin_bit_cast(int, ro.w) : -1,
                                   shadow ? rd.w : __builtin_inff());
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
        bool finished = false, do_step = false;
        TravStep st;
        st.node = false; st.tri = false; st.two = false; st.tslot = 0; st.noff = 0; st.toff = 0;
        if (busy) { do_step = trav_choose(T, S, stack, spill, st); finished = !do_step; }
        if (COUNT) {
            c_wsteps++;
            c_busy += (unsigned)__popcll(__ballot(busy));
            c_nl += (unsigned)__popcll(__ballot(st.node));
            c_tl += (unsigned)__popcll(__ballot(st.tri));
        }
        TravData D;
        trav_fetch(S, st, D);
#if defined(ER_TRACE_PAD_VALU)   // diagnostic builds (tools/ab_pad.sh): what is a step bound by?  N extra dependent VALU operations ...
        {
            float pad = T.U;
#pragma unroll
            for (int k = 0; k < ER_TRACE_PAD_VALU; k++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(pad));
            asm volatile("" ::"v"(pad));
        }
#endif
#if defined(ER_TRACE_PAD_SLEEP)  // ... or N x 64 idle cycles of this wave per step (issue slots stay free for the other waves)
        __builtin_amdgcn_s_sleep(ER_TRACE_PAD_SLEEP);
#endif
        if (busy) {
            if (do_step) {
                if (trav_apply<COUNT>(T, S, st, D, c_nodes, c_tris)) {
                    W.occluded[entry] = 1;   // a certain occluder ends the shadow query
                    finished = true;
                }
            } else if (T.shadow) {
                W.occluded[entry] = T.overflow ? 3 : (T.s0 >= 0 ? 2 : 0);
            }
            if (finished) {
                if (T.shadow) {
                    W.occ_a[entry] = T.s0;
                    W.occ_b[entry] = T.s1;
                } else {
                    W.hit[entry] = T.s0 >= 0 ? T.s0 : T.s1;
                    W.hit2[entry] = T.overflow ? -2 : ((T.s0 >= 0 && T.s1 >= 0) ? T.s1 : -1);
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
            atomicAdd(&S.counters->trace_wave_steps, (unsigned long long)c_wsteps);
            atomicAdd(&S.counters->trace_busy_lanes, (unsigned long long)c_busy);
            atomicAdd(&S.counters->trace_node_lanes, (unsigned long long)c_nl);
            atomicAdd(&S.counters->trace_tri_lanes, (unsigned long long)c_tl);
        }
    }
}

// ---- shade: one bounce-loop step per active slot (src/kernel.cpp:508-645) ----
// Compiled for 4 waves per SIMD (128 VGPRs, 118 spilled) although only 5 shade waves per CU are launched: the three
// slot pools run trace and shade launches side by side, and a shade wave that holds 128 instead of 200 registers
// leaves room for one more trace wave on its SIMD.  Measured on C2 (trace 12 / shade 5 waves per CU): 1045 Msamples/s
// against 962 with the unspilled 200-register build; a 96-register trace kernel (5 per SIMD) loses 10 %.
#ifndef WF_SHADE_WAVES
#define WF_SHADE_WAVES 4
#endif
template <bool COUNT, bool EXT>
__global__ __launch_bounds__(64, WF_SHADE_WAVES) void er_wf_shade(DevScene S, WfState W, uint32_t parity) {
    // queue entries are staged per wave and appended WF_STAGE tickets at a time: the two queue-length words are
    // single addresses, and one address takes ~90 atomics/us whatever the number of waves
    __shared__ uint32_t s_qc[WF_STAGE * 64], s_qs[WF_STAGE * 64];
    unsigned n_qc = 0, n_qs = 0;             // staged entries (wave-uniform)
    const int lane = threadIdx.x;
    int* stack = W.shade_stack + (size_t)blockIdx.x * (ER_STACK * 64) + lane;      // only for the rare exact re-trace of an overflowed ray
    unsigned c_nodes = 0, c_tris = 0;
    const uint32_t nC = W.counts[WF_NC + WF_PAR(parity)];
    if (blockIdx.x == 0 && lane < 8) W.counts[WF_TT + lane * WF_LINE] = 0;
    const uint32_t wC = (nC + 63) >> 6;
    const uint32_t* qc = W.q[parity];
    uint32_t* qn = W.q[parity ^ 1];
    uint32_t* qsn = W.qs[parity ^ 1];
    const size_t npx = (size_t)S.x_res * S.y_res;
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
        if (n_qc > (WF_STAGE - 1) * 64 || n_qs > (WF_STAGE - (EXT ? 2 : 1)) * 64) {   // no room for another ticket: append
            __syncthreads();
            queue_flush(&W.counts[WF_NC + WF_PAR(parity ^ 1)], qn, s_qc, n_qc);
            queue_flush(&W.counts[WF_NS + WF_PAR(parity ^ 1)], qsn, s_qs, n_qs);
            __syncthreads();
            n_qc = 0; n_qs = 0;
        }
        uint32_t ticket = t_cur++;
        uint32_t item = ticket * 64 + lane;
        bool active = item < nC;
        bool push_closest = false, push_shadow = false, push_light = false;
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
            if (EXT && (packed & WF_LPENDING)) {   // ... then its point-light query (second half of the shadow records)
                const uint32_t q = slot + W.slots;
                int occ = W.occluded[q];
                if (occ >= 2) {
                    float4 so = W.sh_o[q], sd = W.sh_d[q];
                    Ray sr;
                    sr.o = f3(so.x, so.y, so.z);
                    sr.d = f3(sd.x, sd.y, sd.z);
                    occ = resolve_shadow<COUNT>(S, stack, sr, __builtin_bit_cast(int, so.w), sd.w, occ, W.occ_a[q], W.occ_b[q], c_nodes, c_tris) ? 1 : 0;
                }
                float4 c = occ ? W.c_occ[q] : W.c_vis[q];
                light = light + f3(c.x, c.y, c.z);
            }
            bool pending = false, lpending = false;
            bool done = fin_only;
            Ray ray;
            float prev_pdf = -1.0f;
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
                if (EXT) prev_pdf = d.w;
#define ER_BOUNCE_HDRI_QUERY(sr, self_slot, d_self, cv, co)                                                     \
    W.sh_o[slot] = make_float4((sr).o.x, (sr).o.y, (sr).o.z, __builtin_bit_cast(float, (int)(self_slot)));      \
    W.sh_d[slot] = make_float4((sr).d.x, (sr).d.y, (sr).d.z, (d_self));                                          \
    W.c_vis[slot] = make_float4((cv).x, (cv).y, (cv).z, 0.0f);                                                   \
    W.c_occ[slot] = make_float4((co).x, (co).y, (co).z, 0.0f)
#define ER_BOUNCE_LIGHT_QUERY(lr, limit, lv, lo)                                                                \
    {   /* the point-light query of a slot lives in the second half of the shadow records */                    \
        const uint32_t lq = slot + W.slots;                                                                      \
        const F3 lv_ = (lv), lo_ = (lo);                                                                         \
        W.sh_o[lq] = make_float4((lr).o.x, (lr).o.y, (lr).o.z, __builtin_bit_cast(float, -1));                   \
        W.sh_d[lq] = make_float4((lr).d.x, (lr).d.y, (lr).d.z, (limit));                                         \
        W.c_vis[lq] = make_float4(lv_.x, lv_.y, lv_.z, 0.0f);                                                    \
        W.c_occ[lq] = make_float4(lo_.x, lo_.y, lo_.z, 0.0f);                                                    \
    }
#define ER_BOUNCE_FIRST_HIT(n, t, b)                                                                            \
    W.aov_n[slot] = make_float4((n).x, (n).y, (n).z, 0.0f);                                                      \
    W.aov_t[slot] = make_float4((t).x, (t).y, (t).z, 0.0f);                                                      \
    W.aov_b[slot] = make_float4((b).x, (b).y, (b).z, 0.0f)
#include "er_bounce.inc"
#undef ER_BOUNCE_HDRI_QUERY
#undef ER_BOUNCE_LIGHT_QUERY
#undef ER_BOUNCE_FIRST_HIT
            }
            bool alive = true;
            if (done && (pending || lpending)) {
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
                        float4* pp = S.passes + er_pass_index(npx, planes[q], idx);
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
                    prev_pdf = -1.0f;
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
            push_light = EXT && lpending;
            if (alive) {
                if (!(next_entry & ER_WF_FINALIZE_ONLY)) {
                    W.ray_o[slot] = make_float4(ray.o.x, ray.o.y, ray.o.z, 0.0f);
                    W.ray_d[slot] = make_float4(ray.d.x, ray.d.y, ray.d.z, EXT ? prev_pdf : -1.0f);
                }
                W.light[slot] = make_float4(light.x, light.y, light.z, __builtin_bit_cast(float, rs));
                W.reduc[slot] = make_float4(reduction.x, reduction.y, reduction.z,
                                            __builtin_bit_cast(float, bounce | (pending ? WF_PENDING : 0u) | ((EXT && lpending) ? WF_LPENDING : 0u)));
            }
        }
        const unsigned long long mc = __ballot(push_closest), ms = __ballot(push_shadow);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (push_closest) s_qc[n_qc + __popcll(mc & below)] = next_entry;
        if (push_shadow) s_qs[n_qs + __popcll(ms & below)] = slot;
        n_qc += __popcll(mc);
        n_qs += __popcll(ms);
        if (EXT) {   // point-light queries (ER_FLAG_POINT_LIGHTS): entry = slot + slots, the second half of the shadow records
            const unsigned long long ml = __ballot(push_light);
            if (push_light) s_qs[n_qs + __popcll(ml & below)] = slot + W.slots;
            n_qs += __popcll(ml);
        }
    }
    __syncthreads();
    queue_flush(&W.counts[WF_NC + WF_PAR(parity ^ 1)], qn, s_qc, n_qc);
    queue_flush(&W.counts[WF_NS + WF_PAR(parity ^ 1)], qsn, s_qs, n_qs);
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


hipError_t er_probe_wavefront(const char** which) {
    hipFuncAttributes a;
    hipError_t e;
    *which = "er_wf_trace";
    if ((e = hipFuncGetAttributes(&a, (const void*)er_wf_trace<false>)) != hipSuccess) return e;
    *which = "er_wf_shade";
    return hipFuncGetAttributes(&a, (const void*)er_wf_shade<false, false>);
}

void er_launch_wf_begin(const DevScene& S, const WfState& W, uint32_t n_samples, hipStream_t stream) {
    if (S.owned_tile_count <= W.pool) return;
    (void)hipMemsetAsync(W.counts, 0, WF_COUNTS * sizeof(uint32_t), stream);
    const uint32_t tiles = (S.owned_tile_count - W.pool + W.pools - 1) / W.pools;
    hipLaunchKernelGGL(er_wf_begin, dim3(tiles), dim3(64), 0, stream, S, W, n_samples);
}
void er_launch_wf_trace(const DevScene& S, const WfState& W, uint32_t parity, bool count, uint32_t blocks, uint32_t* ray_log, hipStream_t stream) {
    static const uint32_t refill_min = [] {
        const char* e = getenv("ER_TRACE_REFILL_MIN");
        int v = e ? atoi(e) : WF_REFILL_MIN;
        return (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
    }();
    if (S.owned_tile_count == 0) return;
    if (count) hipLaunchKernelGGL(er_wf_trace<true>, dim3(blocks), dim3(64), 0, stream, S, W, parity, refill_min, ray_log);
    else hipLaunchKernelGGL(er_wf_trace<false>, dim3(blocks), dim3(64), 0, stream, S, W, parity, refill_min, ray_log);
}
void er_launch_wf_shade(const DevScene& S, const WfState& W, uint32_t parity, bool count, uint32_t blocks, hipStream_t stream) {
    if (S.owned_tile_count == 0) return;
    const bool ext = er_ext_active(S);
    auto k = count ? (ext ? er_wf_shade<true, true> : er_wf_shade<true, false>) : (ext ? er_wf_shade<false, true> : er_wf_shade<false, false>);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, stream, S, W, parity);
}
