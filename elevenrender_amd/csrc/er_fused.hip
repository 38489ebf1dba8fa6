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
template <bool COUNT>
__global__ __launch_bounds__(64, FUSED_WAVES) void er_fused_kernel(DevScene S, uint2* ring_base, uint2* spill_base, uint32_t n_samples) {
    __shared__ uint2 s_stack[WF_LDS_STACK * 64];
    __shared__ float s_park[12 * 64];      // next bounce ray (o, d) + contribution if visible / if occluded
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
    volatile int* mail = s_mail;
    s_mail[lane] = 0;
    const uint32_t n_slots = S.owned_tile_count * 64u;
    const size_t npx = (size_t)S.x_res * S.y_res;
    const int hw = S.hdri_tex.width, hh = S.hdri_tex.height;
    unsigned c_paths = 0, c_bounce = 0, c_rays = 0, c_nodes = 0, c_tris = 0, c_shaded = 0, c_texels = 0, c_hdri = 0;

    Trav T;
    trav_begin(T, f3s(0), f3(0, 0, 1), false, -1, 0.0f);
    int mode = M_IDLE;
    uint32_t slot = 0, idx = 0, rs = 0, left = 0, bounce = 0;
    F3 light = f3s(0), reduction = f3s(1);
    bool terminal = false;   // the path is over but its last shadow ray is still being traced
    int occ_code = 0;
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
                    light = light + (occ ? f3(park[9 * 64], park[10 * 64], park[11 * 64]) : f3(park[6 * 64], park[7 * 64], park[8 * 64]));
                    if (terminal) {
                        fin = true;
                    } else {
                        trav_begin(T, f3(park[0], park[64], park[128]), f3(park[192], park[256], park[320]), false, -1, __builtin_inff());
                        c_rays++;
                        mode = M_TRACE;
                    }
                    }
                } else {   // M_SHADE: one iteration of the bounce loop
                    Ray ray;
                    ray.o = T.o; ray.d = T.d;
                    int hslot = resolve_closest<COUNT>(S, stack2, ray, T.s0 >= 0 ? T.s0 : T.s1,
                                                       T.overflow ? -2 : ((T.s0 >= 0 && T.s1 >= 0) ? T.s1 : -1), c_nodes, c_tris);
                    c_bounce++;
                    bool done = false, pending = false;
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
                        if (shader != -1) {   // asl_shade placeholder, src/shader.cpp:6-10
                            hd.albedo = f3s(0);
                            if (shader >= 0 && shader < 4) hd.albedo = f3(1, 1, 0);
                        }
                        Ray sr;
                        sr.o = f3s(0); sr.d = f3(0, 0, 1);
                        float d_self = __builtin_inff();
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
                                // shadow query needed (er_kernels.hip): occluded iff the closest hit is another triangle
                                F3 c_occ = reduction * (hd.emission + f3s(0) * evalh * absdot / hdripdf);
                                sr = make_ray(hd.position + N * 0.001f, wihdri);
                                F3 v0, v1, v2;
                                float4 qa, qb, qc4;
                                load_verts(S, (uint32_t)hslot, v0, v1, v2, qa, qb, qc4);
                                float su, sv, st;
                                if (tri_mt(v0, v1, v2, sr, su, sv, st)) d_self = candidate_distance(S, (uint32_t)hslot, v0, v1, v2, sr, su, sv, st);
                                park[6 * 64] = c_vis.x; park[7 * 64] = c_vis.y; park[8 * 64] = c_vis.z;
                                park[9 * 64] = c_occ.x; park[10 * 64] = c_occ.y; park[11 * 64] = c_occ.z;
                                pending = true;
                            } else {
                                light = light + c_vis;
                            }
                            float brdfpdf = DisneyPdf(hd, wo, N, wibrdf);
                            reduction = reduction * (DisneyEval(hd, wo, N, wibrdf) * __builtin_fabsf(dot(wibrdf, N)) / brdfpdf);
                            if (bounce == 0) {
                                aov[0] = hd.normal.x; aov[64] = hd.normal.y; aov[128] = hd.normal.z;
                                aov[192] = hd.tangent.x; aov[256] = hd.tangent.y; aov[320] = hd.tangent.z;
                                aov[384] = hd.bitangent.x; aov[448] = hd.bitangent.y; aov[512] = hd.bitangent.z;
                            }
                            ray = make_ray(hit.position + wibrdf * 0.001f, wibrdf);
                        } else {
                            ray = make_ray(hit.position + ray.d * 0.001f, ray.d);
                        }
                        bounce++;
                        if (bounce >= S.max_bounces) done = true;
                        if (pending) {
                            // trace the shadow ray first; the next bounce ray waits in LDS
                            park[0] = ray.o.x; park[64] = ray.o.y; park[128] = ray.o.z;
                            park[192] = ray.d.x; park[256] = ray.d.y; park[320] = ray.d.z;
                            trav_begin(T, sr.o, sr.d, true, hslot, d_self);
                            c_rays++;
                            terminal = done;
                            mode = M_TRACE;
                            if (!done) {   // offer the shadow ray to a free lane (below); taken -> this lane traces the parked ray now
                                float* job = s_job + lane;
                                job[0] = sr.o.x; job[64] = sr.o.y; job[128] = sr.o.z;
                                job[192] = sr.d.x; job[256] = sr.d.y; job[320] = sr.d.z;
                                job[384] = __builtin_bit_cast(float, hslot); job[448] = d_self;
                                offer = true;
                            }
                        } else if (!done) {
                            trav_begin(T, ray.o, ray.d, false, -1, __builtin_inff());
                            c_rays++;
                            mode = M_TRACE;
                        }
                    }
                    if (done && !pending) fin = true;
                }
                if (fin) {
                    // src/kernel.cpp:597-645: clamp, NaN gate, running mean over sa (starts at 1)
                    light = f3(clampf(light.x, 0, 10), clampf(light.y, 0, 10), clampf(light.z, 0, 10));
                    uint32_t sa = S.samples[idx];
                    if (!(light.x != light.x) && !(light.y != light.y) && !(light.z != light.z)) {
                        float k = ((float)sa) / ((float)(sa + 1));
                        float inv = (float)(sa + 1);
                        const F3 vals[4] = {light, f3(aov[0], aov[64], aov[128]), f3(aov[192], aov[256], aov[320]), f3(aov[384], aov[448], aov[512])};
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
                        light = light + (code ? f3(park[9 * 64], park[10 * 64], park[11 * 64]) : f3(park[6 * 64], park[7 * 64], park[8 * 64]));
                        if (terminal) {
                            mode = M_FINALIZE;
                        } else {
                            trav_begin(T, f3(park[0], park[64], park[128]), f3(park[192], park[256], park[320]), false, -1, __builtin_inff());
                            c_rays++;
                        }
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

void er_launch_fused(const DevScene& S, void* ring, void* spill, uint32_t n_samples, bool count, uint32_t blocks, hipStream_t stream) {
    if (S.owned_tile_count == 0 || n_samples == 0) return;
    if (count) hipLaunchKernelGGL(er_fused_kernel<true>, dim3(blocks), dim3(64), 0, stream, S, (uint2*)ring, (uint2*)spill, n_samples);
    else hipLaunchKernelGGL(er_fused_kernel<false>, dim3(blocks), dim3(64), 0, stream, S, (uint2*)ring, (uint2*)spill, n_samples);
}
