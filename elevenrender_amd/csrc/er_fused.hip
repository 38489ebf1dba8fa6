// er_fused.hip -- lane-asynchronous fused schedule of the per-sample path for gfx950.
//
// Same arithmetic per pixel as renderingKernel (reference src/kernel.cpp:477-646) and as the other two
// schedules of this library (er_wavefront.hip, er_kernels.hip); what changes is, again, only the schedule:
//
//   * persistent waves; every WAVE owns a fixed share of the pixels (chunks of four, dealt round-robin) and keeps
//     the ones that are not in flight in a ring of (pixel, samples left) records; a LANE takes the pixel at the
//     head of the ring, runs ONE sample of it and puts it back at the tail.  A pixel's samples are one RNG stream,
//     so they must run one after the other -- but not on the same lane: with whole pixels as the unit a GPU that
//     owns 1.3 pixels per lane (an eighth of a 1080p frame) needs two rounds and idles through a third of them;
//     with single samples as the unit all of a wave's pixels advance side by side and finish together.  The ring
//     is private to its wave, so there is no atomic and no cross-wave ordering anywhere in this kernel;
//   * there is NO barrier between bounces, neither across the GPU nor inside the wave: in every loop
//     iteration the lanes that are tracing advance their ray by one traversal step (er_trav.h, one unified
//     96-byte fetch for the whole wave), and the lanes whose ray has finished wait in a small "needs shading"
//     state; the shading code runs for them as a batch as soon as enough lanes wait (or nobody is tracing);
//   * a shadow ray is traced by the same lane before the next bounce ray (which is parked in LDS together
//     with the two candidate contributions); its result only selects which contribution is added.
//
// The wavefront schedule needs one kernel pair per bounce and every launch lasts at least as long as its
// longest ray; with few pixels per GPU (tile-sharded multi-GPU runs, the tail of a call) those floors dominate.
// Here the critical path of a pixel is only its own rays and shading steps.
#include "er_device.h"
#include "er_kernels.h"
#include "er_trav.h"
#include "er_shade.h"

using namespace erd;

namespace {

enum { M_IDLE = 0, M_START = 1, M_TRACE = 2, M_SHADE = 3, M_RESOLVE = 4, M_FINALIZE = 5 };

__device__ __forceinline__ unsigned fwave_sum(unsigned v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

#ifndef FUSED_BATCH_MIN
#define FUSED_BATCH_MIN 32   // measured on C2: 12 -> 645, 20 -> 736, 32 -> 778, 48 -> 736 Msamples/s
#endif
#ifndef FUSED_BATCH_LOW
#define FUSED_BATCH_LOW 4
#endif
#ifndef FUSED_REFILL_MIN
#define FUSED_REFILL_MIN 8
#endif

}  // namespace

// waves per SIMD the kernel is compiled for.  Measured on C2 (full frame), waves per CU in brackets: 2 -> 863 [8];
// 3 -> 962 [12] (66 VGPRs spilled, but half as many again resident waves); 4 -> 872 [16] (180 spilled)
#ifndef FUSED_WAVES
#define FUSED_WAVES 3
#endif
// park layout (floats per lane, stride 64): 0-5 next bounce ray (o, d), 6-8 / 9-11 HDRI contribution if visible / if
// occluded, 12 brdfpdf of the last opaque bounce (ER_FLAG_MIS); LIGHTS only: 13-18 point-light shadow ray (o, d),
// 19 its limit, 20-22 / 23-25 its contribution if visible / if occluded
#define PK_NEXT 0
#define PK_CVIS 6
#define PK_COCC 9
#define PK_PDF 12
#define PK_LRAY 13
#define PK_LLIM 19
#define PK_LVIS 20
#define PK_LOCC 23
template <bool COUNT, bool LIGHTS>   // LIGHTS = the EXT of er_shade.h (point lights and / or MIS)
__global__ __launch_bounds__(64, FUSED_WAVES) void er_fused_kernel(DevScene S, uint2* ring_base, uint2* spill_base, uint32_t n_samples) {
    __shared__ uint2 s_stack[WF_LDS_STACK * 64];
    __shared__ float s_park[(LIGHTS ? 26 : 13) * 64];
    __shared__ float s_aov[9 * 64];        // first-bounce normal / tangent / bitangent of the current path
    __shared__ float s_job[8 * 64];        // shadow ray a lane offers to a helper: o, d, bits(slot it leaves), self-hit distance
    __shared__ int s_mail[64];             // helper -> owner: 0 nothing yet, 1 visible, 2 occluded
    __shared__ unsigned char s_list[64];   // k-th offering lane, for the k-th free lane
    const int lane = threadIdx.x;
    uint2* stack = s_stack + lane;
    uint2* spill = spill_base + (size_t)blockIdx.x * (ER_STACK * 64) + lane;
    // exact re-trace fallback (binary BVH, rare): its stack lives in HBM behind the wave's spill area, so that LDS
    // (9.3 KB per wave) does not cap the number of resident waves
    int* stack2 = (int*)(spill_base + (size_t)gridDim.x * (ER_STACK * 64)) + (size_t)blockIdx.x * (ER_STACK * 64) + lane;
    float* park = s_park + lane;
    float* aov = s_aov + lane;
    volatile __attribute__((address_space(3))) int* mail = (volatile __attribute__((address_space(3))) int*)s_mail;      // (explicitly LDS: a volatile access through a generic pointer is a flat_load)
    s_mail[lane] = 0;
    const uint32_t n_slots = S.owned_tile_count * 64u;
    unsigned c_paths = 0, c_bounce = 0, c_rays = 0, c_nodes = 0, c_tris = 0, c_shaded = 0, c_texels = 0, c_hdri = 0;

    Trav T;
    trav_begin(T, f3s(0), f3(0, 0, 1), false, -1, 0.0f);
    int mode = M_IDLE;
    uint32_t slot = 0, idx = 0, rs = 0, left = 0, bounce = 0;
    F3 light = f3s(0), reduction = f3s(1);
    bool terminal = false;   // the path is over but its last shadow ray is still being traced
    int occ_code = 0;
    bool lnext = false;      // LIGHTS: a point-light shadow ray is parked; it is traced after the HDRI one
    bool lkind = false;      // LIGHTS: the shadow ray this lane is tracing for itself is the point-light one
    // Shadow helpers.  A pixel's samples -- and the bounces of a sample -- are a sequential chain, and when a wave has
    // fewer pixels than lanes (a GPU that owns few pixels; the tail of a call) that chain is all that matters.  Half
    // of it is shadow rays, whose only effect is to select one of two parked contributions: a lane with nothing to
    // do traces the shadow ray of another lane, which meanwhile goes on with its next bounce, and reports through
    // s_mail.  The owner adds the selected contribution before it touches `light` again, so the order of the
    // additions -- and with it every bit of the result -- is unchanged.
    int owner = -1;          // >= 0: this lane is tracing a shadow ray for lane `owner`
    bool sh_out = false;     // this lane's last shadow query is with a helper

    // ---- this wave's pixels: chunks c = wave, wave + waves, ... of four consecutive slots ----
    const uint32_t n_waves = gridDim.x, wave = blockIdx.x;
    const uint32_t chunks = n_slots >> 2;
    const uint32_t my_slots = (chunks > wave ? (chunks - wave + n_waves - 1) / n_waves : 0u) * 4u;
    const uint32_t cap = ((chunks + n_waves - 1) / n_waves) * 4u;      // ring capacity = most slots any wave owns
    uint2* ring = ring_base + (size_t)wave * cap;
    uint32_t head = 0, tail = 0;                                       // wave-uniform, monotonic; entry i lives at i % cap
    if (n_samples > 0) {
        for (uint32_t i0 = 0; i0 < my_slots; i0 += 64) {
            const uint32_t i = i0 + lane;
            bool valid = false;
            uint32_t sl = 0;
            if (i < my_slots) {
                sl = ((wave + (i >> 2) * n_waves) << 2) | (i & 3u);
                const uint32_t tile = S.owned_tiles[sl >> 6], l = sl & 63;
                valid = (tile % S.tiles_x) * ER_TILE + (l & 7) < S.x_res && (tile / S.tiles_x) * ER_TILE + (l >> 3) < S.y_res;
            }
            const unsigned long long m = __ballot(valid);
            if (valid) ring[tail + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2(sl, n_samples);
            tail += (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }

    // contribution of this lane's own shadow query (HDRI or point light) by outcome
    auto own_contribution = [&](bool occluded) {
        const int b = (LIGHTS && lkind) ? (occluded ? PK_LOCC : PK_LVIS) : (occluded ? PK_COCC : PK_CVIS);
        return f3(park[b * 64], park[(b + 1) * 64], park[(b + 2) * 64]);
    };
    // what follows a finished own shadow query: the parked point-light query, else the end of the path, else the parked ray
    auto after_own_shadow = [&]() {
        if (LIGHTS && lnext) {
            lnext = false;
            lkind = true;
            trav_begin(T, f3(park[PK_LRAY * 64], park[(PK_LRAY + 1) * 64], park[(PK_LRAY + 2) * 64]),
                       f3(park[(PK_LRAY + 3) * 64], park[(PK_LRAY + 4) * 64], park[(PK_LRAY + 5) * 64]), true, -1, park[PK_LLIM * 64]);
            c_rays++;
            mode = M_TRACE;
        } else if (terminal) {
            mode = M_FINALIZE;
        } else {
            trav_begin(T, f3(park[0], park[64], park[128]), f3(park[192], park[256], park[320]), false, -1, __builtin_inff());
            c_rays++;
            mode = M_TRACE;
        }
    };

    while (true) {
        // (pixels are handed out inside the batch, right after the lanes that finished a sample put theirs back)
        const unsigned long long idle = __ballot(mode == M_IDLE);
        // idle lanes count as waiting for the batch while the ring has pixels for them
        // (a lane whose shadow query is still with a helper cannot shade yet)
        const bool mail_ok = !sh_out || mail[lane] != 0;
        const unsigned long long want_batch = __ballot((mode == M_SHADE || mode == M_RESOLVE || mode == M_FINALIZE) && mail_ok) | (tail != head ? idle : 0ull);
        const unsigned long long tracing = __ballot(mode == M_TRACE);
        if (want_batch == 0 && tracing == 0) break;      // nothing in flight and the ring is empty: this wave is done
        // (a lane that waits for mail has a helper that is tracing or waiting for the batch, so this cannot strand it)

        // ---- batch: the shading step (src/kernel.cpp:508-645) for the lanes that wait for it ----
        // A full wave batches 32 lanes at a time (the shading code costs the same for 1 lane as for 64).  A wave with
        // few pixels left -- the tail of a call, or a GPU that owns few pixels -- is bound by the latency of its
        // longest pixel (the samples of a pixel are sequential), so there a waiting lane is served sooner.
        const unsigned n_active = 64u - (unsigned)__popcll(idle);
        const unsigned batch_min = n_active >= 2u * FUSED_BATCH_MIN ? FUSED_BATCH_MIN : (n_active / 2u < FUSED_BATCH_LOW ? FUSED_BATCH_LOW : n_active / 2u);
        if (__popcll(want_batch) >= batch_min || tracing == 0) {
            bool put_back = false, offer = false;
            if ((mode == M_SHADE || mode == M_RESOLVE || mode == M_FINALIZE) && mail_ok) {
                bool fin = false;
                if (sh_out) {            // the helper has answered: the contribution of the previous bounce, in order
                    const int r = mail[lane];
                    light = light + (r == 2 ? f3(park[9 * 64], park[10 * 64], park[11 * 64]) : f3(park[6 * 64], park[7 * 64], park[8 * 64]));
                    mail[lane] = 0;
                    sh_out = false;
                }
                if (mode == M_FINALIZE) {
                    fin = true;
                } else if (mode == M_RESOLVE) {
                    // the traversal could not decide the shadow query from t-intervals: exact metric
                    Ray sr;
                    sr.o = T.o; sr.d = T.d;
                    bool occ = resolve_shadow<COUNT>(S, stack2, sr, T.skip, T.limit, occ_code, T.s0, T.s1, c_nodes, c_tris);
                    if (owner >= 0) {
                        mail[owner] = occ ? 2 : 1;
                        owner = -1;
                        mode = M_IDLE;
                    } else {
                    light = light + own_contribution(occ);
                    after_own_shadow();
                    if (mode == M_FINALIZE) fin = true;
                    }
                } else {   // M_SHADE: one iteration of the bounce loop (er_shade.h)
                    Ray ray;
                    ray.o = T.o; ray.d = T.d;
                    const int hslot = resolve_closest<COUNT>(S, stack2, ray, T.s0 >= 0 ? T.s0 : T.s1,
                                                             T.overflow ? -2 : ((T.s0 >= 0 && T.s1 >= 0) ? T.s1 : -1), c_nodes, c_tris);
                    c_bounce++;
                    float prev_pdf = LIGHTS ? park[PK_PDF * 64] : -1.0f;
                    bool done = false, pending = false, lpending = false;
                    constexpr bool EXT = LIGHTS;
                    float* job = s_job + lane;
                    // the hooks park the queries (and the AOVs) in this lane's LDS columns
#define ER_BOUNCE_HDRI_QUERY(sr, self_slot, d_self, cv, co)                                                          \
    job[0] = (sr).o.x; job[64] = (sr).o.y; job[128] = (sr).o.z;                                                       \
    job[192] = (sr).d.x; job[256] = (sr).d.y; job[320] = (sr).d.z;                                                    \
    job[384] = __builtin_bit_cast(float, (int)(self_slot)); job[448] = (d_self);                                      \
    park[PK_CVIS * 64] = (cv).x; park[(PK_CVIS + 1) * 64] = (cv).y; park[(PK_CVIS + 2) * 64] = (cv).z;                \
    park[PK_COCC * 64] = (co).x; park[(PK_COCC + 1) * 64] = (co).y; park[(PK_COCC + 2) * 64] = (co).z
#define ER_BOUNCE_LIGHT_QUERY(lr, limit, lv, lo)                                                                     \
    {                                                                                                                 \
        const F3 lv_ = (lv), lo_ = (lo);                                                                              \
        park[PK_LRAY * 64] = (lr).o.x; park[(PK_LRAY + 1) * 64] = (lr).o.y; park[(PK_LRAY + 2) * 64] = (lr).o.z;      \
        park[(PK_LRAY + 3) * 64] = (lr).d.x; park[(PK_LRAY + 4) * 64] = (lr).d.y; park[(PK_LRAY + 5) * 64] = (lr).d.z; \
        park[PK_LLIM * 64] = (limit);                                                                                 \
        park[PK_LVIS * 64] = lv_.x; park[(PK_LVIS + 1) * 64] = lv_.y; park[(PK_LVIS + 2) * 64] = lv_.z;                \
        park[PK_LOCC * 64] = lo_.x; park[(PK_LOCC + 1) * 64] = lo_.y; park[(PK_LOCC + 2) * 64] = lo_.z;                \
    }
#define ER_BOUNCE_FIRST_HIT(n, t, b)                                                                                 \
    aov[0] = (n).x; aov[64] = (n).y; aov[128] = (n).z;                                                                \
    aov[192] = (t).x; aov[256] = (t).y; aov[320] = (t).z;                                                             \
    aov[384] = (b).x; aov[448] = (b).y; aov[512] = (b).z
#include "er_bounce.inc"
#undef ER_BOUNCE_HDRI_QUERY
#undef ER_BOUNCE_LIGHT_QUERY
#undef ER_BOUNCE_FIRST_HIT
                    if (LIGHTS) park[PK_PDF * 64] = prev_pdf;
                    struct { bool done, shadow, lshadow; Ray next; } bo = {done, pending, lpending, ray};
                    const bool lsh = LIGHTS && bo.lshadow;
                    if (bo.shadow || lsh) {
                        // trace the shadow ray(s) first; the next bounce ray waits in LDS
                        park[0] = bo.next.o.x; park[64] = bo.next.o.y; park[128] = bo.next.o.z;
                        park[192] = bo.next.d.x; park[256] = bo.next.d.y; park[320] = bo.next.d.z;
                        terminal = bo.done;
                        if (bo.shadow) {
                            lnext = lsh;
                            lkind = false;
                            // (the hook left the HDRI shadow ray in s_job)
                            trav_begin(T, f3(job[0], job[64], job[128]), f3(job[192], job[256], job[320]), true, __builtin_bit_cast(int, job[384]), job[448]);
                            c_rays++;
                            mode = M_TRACE;
                            // offer the HDRI shadow ray to a free lane (below); taken -> this lane traces the parked ray now.
                            // (Not when a point-light query follows: its contribution must be added after the HDRI one.)
                            offer = !bo.done && !lsh;
                        } else {
                            lnext = true;
                            after_own_shadow();     // starts the point-light query
                        }
                    } else if (!bo.done) {
                        trav_begin(T, bo.next.o, bo.next.d, false, -1, __builtin_inff());
                        c_rays++;
                        mode = M_TRACE;
                    } else {
                        fin = true;
                    }
                }
                if (fin) {
                    const uint32_t sa = S.samples[idx];
                    const uint32_t sa2 = accumulate_sample(S, idx, sa, light, f3(aov[0], aov[64], aov[128]), f3(aov[192], aov[256], aov[320]),
                                                           f3(aov[384], aov[448], aov[512]));   // src/kernel.cpp:597-645
                    if (sa2 != sa) S.samples[idx] = sa2;
                    S.rng[idx] = rs;
                    c_paths++;
                    left--;
                    put_back = left > 0;      // the pixel's next sample goes to whichever lane reaches it first
                    mode = M_IDLE;
                }
            }
            // ---- ring exchange: pixels whose sample just finished go to the tail, idle lanes take the head ----
            {
                const uint32_t avail0 = tail - head;
                const unsigned long long pm = __ballot(put_back);
                if (put_back) ring[(tail + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull))) % cap] = make_uint2(slot, left);
                tail += (uint32_t)__popcll(pm);
                const unsigned long long im = __ballot(mode == M_IDLE);
                const uint32_t n_im = (uint32_t)__popcll(im);
                const uint32_t take = n_im < tail - head ? n_im : tail - head;
                // a pixel's state (rng, planes, sample count, its ring record) was stored by a lane of THIS wave:
                // same CU, same L1 -- the stores only have to have completed.  Records of earlier batches are
                // long complete; a wait is needed only when this batch's own records are handed out again.
                if (take > avail0) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                if (mode == M_IDLE) {
                    const uint32_t r = (uint32_t)__popcll(im & ((1ull << lane) - 1ull));
                    if (r < take) {
                        const uint2 e = ring[(head + r) % cap];
                        slot = e.x;
                        left = e.y;
                        const uint32_t tile = S.owned_tiles[slot >> 6], l = slot & 63;
                        idx = ((tile / S.tiles_x) * ER_TILE + (l >> 3)) * S.x_res + (tile % S.tiles_x) * ER_TILE + (l & 7);
                        mode = M_START;
                    }
                }
                head += take;
            }
            // ---- shadow helpers: lanes that are still free take the shadow rays offered in this batch ----
            {
                const unsigned long long om = __ballot(offer), fm = __ballot(mode == M_IDLE);
                const unsigned n_o = (unsigned)__popcll(om), n_f = (unsigned)__popcll(fm);
                const unsigned n_match = n_o < n_f ? n_o : n_f;
                if (n_match > 0) {
                    const unsigned below_o = (unsigned)__popcll(om & ((1ull << lane) - 1ull)), below_f = (unsigned)__popcll(fm & ((1ull << lane) - 1ull));
                    if (offer && below_o < n_match) s_list[below_o] = (unsigned char)lane;
                    __syncthreads();
                    if (mode == M_IDLE && below_f < n_match) {
                        const int o = s_list[below_f];
                        const float* job = s_job + o;
                        trav_begin(T, f3(job[0], job[64], job[128]), f3(job[192], job[256], job[320]), true, __builtin_bit_cast(int, job[384]), job[448]);
                        owner = o;
                        mode = M_TRACE;
                    }
                    if (offer && below_o < n_match) {
                        trav_begin(T, f3(park[0], park[64], park[128]), f3(park[192], park[256], park[320]), false, -1, __builtin_inff());
                        c_rays++;
                        sh_out = true;
                    }
                }
            }
            // ---- start the next sample of the pixels just taken (src/kernel.cpp:492-493: five draws, left to right) ----
            if (mode == M_START) {
                rs = S.rng[idx];
                float c1 = rng_next(rs), c2 = rng_next(rs), c3 = rng_next(rs), c4 = rng_next(rs), c5 = rng_next(rs);
                Ray ray = camera_ray(S.cam, (int)(idx % S.x_res), (int)(idx / S.x_res), S.x_res, S.y_res, c1, c2, c3, c4, c5);
                light = f3s(0);
                reduction = f3s(1);
                bounce = 0;
                terminal = false;
                if (LIGHTS) park[PK_PDF * 64] = -1.0f;
#pragma unroll
                for (int q = 0; q < 9; q++) aov[q * 64] = 0.0f;
                trav_begin(T, ray.o, ray.d, false, -1, __builtin_inff());
                c_rays++;
                mode = M_TRACE;
            }
        }

        // ---- one traversal step for every tracing lane ----
        if (__ballot(mode == M_TRACE) != 0) {
            bool finished = false, do_step = false;
            TravStep st;
            st.node = false; st.tri = false; st.two = false; st.tslot = 0; st.noff = 0; st.toff = 0;
            if (mode == M_TRACE) { do_step = trav_choose(T, S, stack, spill, st); finished = !do_step; }
            TravData D;
            trav_fetch(S, st, D);
            if (mode == M_TRACE) {
                int code = -1;
                if (do_step) {
                    if (trav_apply<COUNT>(T, S, st, D, c_nodes, c_tris)) { finished = true; code = 1; }
                } else if (T.shadow) {
                    code = T.overflow ? 3 : (T.s0 >= 0 ? 2 : 0);
                }
                if (finished) {
                    if (!T.shadow) {
                        mode = M_SHADE;
                    } else if (code <= 1 && owner >= 0) {
                        mail[owner] = code ? 2 : 1;
                        owner = -1;
                        mode = M_IDLE;
                    } else if (code <= 1) {
                        // the shadow query's outcome only selects which precomputed contribution is added
                        light = light + own_contribution(code != 0);
                        after_own_shadow();
                    } else {
                        occ_code = code;
                        mode = M_RESOLVE;
                    }
                }
            }
        }
    }
    unsigned t0 = fwave_sum(c_paths), t1 = fwave_sum(c_bounce), t2 = fwave_sum(c_rays), t3 = fwave_sum(c_shaded), t4 = fwave_sum(c_hdri);
    unsigned t5 = 0, t6 = 0, t7 = 0;
    if (COUNT) { t5 = fwave_sum(c_nodes); t6 = fwave_sum(c_tris); t7 = fwave_sum(c_texels); }
    if (lane == 0 && (t0 | t1 | t2)) {
        atomicAdd(&S.counters->paths, (unsigned long long)t0);
        atomicAdd(&S.counters->bounce_samples, (unsigned long long)t1);
        atomicAdd(&S.counters->rays, (unsigned long long)t2);
        atomicAdd(&S.counters->shaded_hits, (unsigned long long)t3);
        atomicAdd(&S.counters->hdri_samples, (unsigned long long)t4);
        if (COUNT) {
            atomicAdd(&S.counters->node_visits, (unsigned long long)t5);
            atomicAdd(&S.counters->tri_tests, (unsigned long long)t6);
            atomicAdd(&S.counters->texel_fetches, (unsigned long long)t7);
        }
    }
}

hipError_t er_probe_fused(const char** which) {
    hipFuncAttributes a;
    *which = "er_fused_kernel";
    hipError_t e = hipFuncGetAttributes(&a, (const void*)er_fused_kernel<false, false>);
    if (e != hipSuccess) return e;
    return hipFuncGetAttributes(&a, (const void*)er_fused_kernel<false, true>);
}

void er_launch_fused(const DevScene& S, void* ring, void* spill, uint32_t n_samples, bool count, uint32_t blocks, hipStream_t stream) {
    if (S.owned_tile_count == 0 || n_samples == 0) return;
    const bool lights = er_ext_active(S);   // bigger LDS park area
    auto k = count ? (lights ? er_fused_kernel<true, true> : er_fused_kernel<true, false>)
                   : (lights ? er_fused_kernel<false, true> : er_fused_kernel<false, false>);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, stream, S, (uint2*)ring, (uint2*)spill, n_samples);
}
