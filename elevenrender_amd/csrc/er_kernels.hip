// er_kernels.hip -- gfx950 kernels of the path-tracing hot path.
//
// er_render_kernel is the MI355X counterpart of renderingKernel (reference
// src/kernel.cpp:477-646) plus its launch loop kernel_render_enqueue (:680-706):
// one 64-lane wavefront owns one 8x8 pixel tile, each lane owns one pixel and runs that
// pixel's samples back to back (a pixel's samples form ONE RNG stream, so they are
// inherently sequential; different pixels are independent).  A lane whose path ends
// starts its next sample immediately ("path regeneration"), so wavefronts stay full
// until a lane has finished all n samples instead of idling to the longest path.
#include "er_kernels.h"
#include "er_device.h"

using namespace erd;

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// setupKernel, reference src/kernel.cpp:176-213 -- for EVERY pixel (the reference's launch
// rounds the range down to a multiple of the block size, leaving edge pixels uninitialised;
// that defect is not reproduced).
__global__ void er_setup_kernel(DevScene S) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t npx = S.x_res * S.y_res;
    if (idx >= npx) return;
    S.rng[idx] = jenkins_u32(idx + 1);
    for (int p = 0; p < ER_PASS_COUNT; p++) S.passes[(size_t)p * npx + idx] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    S.samples[idx] = 1;
}

template <bool COUNT>
__global__ __launch_bounds__(64) void er_render_kernel(DevScene S, uint32_t n_samples) {
    __shared__ int s_stack[ER_STACK * 64];
    const int lane = threadIdx.x;
    int* stack = s_stack + lane;
    const uint32_t tile = S.owned_tiles[blockIdx.x];
    const uint32_t tx = tile % S.tiles_x, ty = tile / S.tiles_x;
    const uint32_t px = tx * ER_TILE + (lane & 7), py = ty * ER_TILE + (lane >> 3);
    unsigned c_paths = 0, c_bounce = 0, c_rays = 0, c_nodes = 0, c_tris = 0, c_shaded = 0, c_texels = 0, c_hdri = 0;

    if (px < S.x_res && py < S.y_res && n_samples > 0) {
        const uint32_t idx = py * S.x_res + px;
        const size_t npx = (size_t)S.x_res * S.y_res;
        uint32_t rs = S.rng[idx];
        uint32_t sa = S.samples[idx];
        const int hw = S.hdri_tex.width, hh = S.hdri_tex.height;

        uint32_t s = 0;
        uint32_t bounce = 0;
        bool fresh = true;
        Ray ray;
        F3 light, reduction, aov_n, aov_t, aov_b;
        while (s < n_samples) {
            if (fresh) {
                // src/kernel.cpp:492-493 -- five draws, left to right
                float c1 = rng_next(rs), c2 = rng_next(rs), c3 = rng_next(rs), c4 = rng_next(rs), c5 = rng_next(rs);
                ray = camera_ray(S.cam, (int)px, (int)py, S.x_res, S.y_res, c1, c2, c3, c4, c5);
                light = f3s(0); reduction = f3s(1); aov_n = f3s(0); aov_t = f3s(0); aov_b = f3s(0);
                bounce = 0;
                fresh = false;
            }
            // ---- one iteration of the bounce loop, src/kernel.cpp:508-593 ----
            c_bounce++;
            bool done = false;
            float dist;
            c_rays++;
            int slot = trace<COUNT, false>(S, stack, ray, -1, __builtin_inff(), dist, c_nodes, c_tris);
            if (slot < 0) {
                float u, v;
                spherical_mapping(-1 * ray.d, u, v);
                light = light + reduction * tex_filtered(S, S.hdri_tex, u, v);
                if (COUNT) c_texels++;
                done = true;
            } else {
                c_shaded++;
                HitFull hit;
                full_hit(S, (uint32_t)slot, ray, hit);
                const ErMaterial& mat = S.materials[hit.material];
                HitData hd;
                generate_hit_data<COUNT>(S, mat, hit, hd, c_texels);
                int shader = mat.albedo_shader_id;
                if (shader != -1) {   // asl_shade placeholder, src/shader.cpp:6-10, src/shader.h:10-11
                    hd.albedo = f3s(0);
                    if (shader >= 0 && shader < 4) hd.albedo = f3(1, 1, 0);
                }
                if (rng_next(rs) <= hd.opacity) {
                    F3 wo = ray.d * -1.0f;
                    F3 N = hd.normal;
                    c_hdri++;
                    int count = hdri_binary_search(S.hdri_cdf, rng_next(rs), hw * hh);   // HDRI::sample
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
                    // The reference always traces the shadow ray (src/kernel.cpp:555-562).  When the
                    // BRDF term is exactly zero the product below is the same for either outcome, so
                    // the query is skipped; otherwise: occluded iff the closest hit is another triangle.
                    if (evalh.x != 0.0f || evalh.y != 0.0f || evalh.z != 0.0f) {
                        Ray sr = make_ray(hd.position + N * 0.001f, wihdri);
                        F3 v0, v1, v2;
                        float4 qa, qb, qc;
                        load_verts(S, (uint32_t)slot, v0, v1, v2, qa, qb, qc);
                        float su, sv, st, d_self = __builtin_inff();
                        if (tri_mt(v0, v1, v2, sr, su, sv, st)) d_self = candidate_distance(S, (uint32_t)slot, v0, v1, v2, sr, su, sv, st);
                        float sd;
                        c_rays++;
                        int occ = trace<COUNT, true>(S, stack, sr, slot, d_self, sd, c_nodes, c_tris);
                        if (occ >= 0) hdriValue = f3s(0);
                    }
                    float hdripdf = hdri_pdf(S, ermath::f2i(iu * hw), ermath::f2i(iv * hh));
                    F3 hdriInt = hdriValue * evalh * __builtin_fabsf(dot(wihdri, N)) / hdripdf;
                    float brdfpdf = DisneyPdf(hd, wo, N, wibrdf);
                    light = light + reduction * (hd.emission + hdriInt);
                    reduction = reduction * (DisneyEval(hd, wo, N, wibrdf) * __builtin_fabsf(dot(wibrdf, N)) / brdfpdf);
                    if (bounce == 0) { aov_n = hd.normal; aov_t = hd.tangent; aov_b = hd.bitangent; }
                    ray = make_ray(hit.position + wibrdf * 0.001f, wibrdf);
                } else {
                    ray = make_ray(hit.position + ray.d * 0.001f, ray.d);
                }
                bounce++;
                if (bounce >= S.max_bounces) done = true;
            }
            if (done) {
                // src/kernel.cpp:597-645: clamp, NaN gate, running mean over sa (starts at 1)
                light = f3(clampf(light.x, 0, 10), clampf(light.y, 0, 10), clampf(light.z, 0, 10));
                if (!(light.x != light.x) && !(light.y != light.y) && !(light.z != light.z)) {
                    float k = ((float)sa) / ((float)(sa + 1));
                    float inv = (float)(sa + 1);
                    const F3 vals[4] = {light, aov_n, aov_t, aov_b};
                    const int planes[4] = {ER_PASS_BEAUTY, ER_PASS_NORMAL, ER_PASS_TANGENT, ER_PASS_BITANGENT};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        float4* pp = S.passes + (size_t)planes[q] * npx + idx;
                        float4 p = *pp;
                        if (sa > 0) { p.x *= k; p.y *= k; p.z *= k; }
                        p.x += vals[q].x / inv; p.y += vals[q].y / inv; p.z += vals[q].z / inv;
                        *pp = p;
                    }
                    sa++;
                }
                c_paths++;
                s++;
                fresh = true;
            }
        }
        S.rng[idx] = rs;
        S.samples[idx] = sa;
    }
    // per-wave counter reduction, one atomic per counter per wave
    unsigned t0 = wave_sum(c_paths), t1 = wave_sum(c_bounce), t2 = wave_sum(c_rays), t3 = wave_sum(c_shaded), t4 = wave_sum(c_hdri);
    unsigned t5 = 0, t6 = 0, t7 = 0;
    if (COUNT) { t5 = wave_sum(c_nodes); t6 = wave_sum(c_tris); t7 = wave_sum(c_texels); }
    if (lane == 0) {
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

// owned pixels <-> compact buffer [owned_tile][64] float4 (lanes outside the image are zero)
__global__ __launch_bounds__(64) void er_pack_kernel(DevScene S, const uint32_t* tiles, int pass, float4* dst) {
    const int lane = threadIdx.x;
    const uint32_t tile = tiles[blockIdx.x];
    const uint32_t tx = tile % S.tiles_x, ty = tile / S.tiles_x;
    const uint32_t px = tx * ER_TILE + (lane & 7), py = ty * ER_TILE + (lane >> 3);
    float4 v = make_float4(0, 0, 0, 0);
    if (px < S.x_res && py < S.y_res) v = S.passes[(size_t)pass * S.x_res * S.y_res + (size_t)py * S.x_res + px];
    dst[(size_t)blockIdx.x * 64 + lane] = v;
}
__global__ __launch_bounds__(64) void er_unpack_kernel(DevScene S, const uint32_t* tiles, int pass, const float4* src) {
    const int lane = threadIdx.x;
    const uint32_t tile = tiles[blockIdx.x];
    const uint32_t tx = tile % S.tiles_x, ty = tile / S.tiles_x;
    const uint32_t px = tx * ER_TILE + (lane & 7), py = ty * ER_TILE + (lane >> 3);
    if (px < S.x_res && py < S.y_res)
        S.passes[(size_t)pass * S.x_res * S.y_res + (size_t)py * S.x_res + px] = src[(size_t)blockIdx.x * 64 + lane];
}

// ---- debug: closest hit of arbitrary rays through the exact routine (include/eleven_hip_debug.h) ----
__global__ __launch_bounds__(64) void er_debug_hit_kernel(DevScene S, const float* __restrict__ o, const float* __restrict__ d, uint32_t n,
                                                           int32_t* __restrict__ tri_out, float* __restrict__ pos_out, float* __restrict__ dist_out) {
    __shared__ int s_stack[ER_STACK * 64];
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    Ray ray;
    ray.o = f3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
    ray.d = f3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    float dist = 0;
    unsigned cn = 0, ct = 0;
    const int slot = trace<false, false>(S, s_stack + threadIdx.x, ray, -1, __builtin_inff(), dist, cn, ct);
    tri_out[i] = -1;
    pos_out[3 * i] = pos_out[3 * i + 1] = pos_out[3 * i + 2] = 0.0f;
    dist_out[i] = dist;
    if (slot >= 0) {
        HitFull h;
        full_hit(S, (uint32_t)slot, ray, h);
        tri_out[i] = __builtin_bit_cast(int, S.tri_isect[(size_t)slot * 3].w);
        pos_out[3 * i] = h.position.x; pos_out[3 * i + 1] = h.position.y; pos_out[3 * i + 2] = h.position.z;
    }
}
void er_launch_debug_hit(const DevScene& S, const float* o, const float* d, uint32_t n, int32_t* tri, float* pos, float* dist, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(er_debug_hit_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, S, o, d, n, tri, pos, dist);
}

void er_launch_setup(const DevScene& S, hipStream_t stream) {
    uint32_t npx = S.x_res * S.y_res;
    if (npx == 0) return;
    hipLaunchKernelGGL(er_setup_kernel, dim3((npx + 255) / 256), dim3(256), 0, stream, S);
}
void er_launch_render(const DevScene& S, uint32_t n_samples, bool count, hipStream_t stream) {
    if (S.owned_tile_count == 0 || n_samples == 0) return;
    if (count) hipLaunchKernelGGL(er_render_kernel<true>, dim3(S.owned_tile_count), dim3(64), 0, stream, S, n_samples);
    else hipLaunchKernelGGL(er_render_kernel<false>, dim3(S.owned_tile_count), dim3(64), 0, stream, S, n_samples);
}
// ---- denoise: edge-avoiding a-trous wavelet filter of the BEAUTY plane, guided by colour and
// by the first-bounce NORMAL plane, into the DENOISE plane that the reference allocates but never writes (reference
// src/kernel.cpp:604; its host-side `get_pass denoise` calls OIDN, src/Managers.cpp:319-343, a neural filter that is
// not on this path and has no arithmetic to match).  One launch per level; level k taps a 5x5 B3-spline stencil with
// holes of 2^k pixels (Dammertz et al. 2010).  Edge-stopping weights are rational -- w = 1 / (1 + kc |dc|^2) and
// max(0, n.n')^2 -- so the filter is plain IEEE arithmetic (tests/test_gpu_denoise.py replays it in numpy, bit for bit).
__global__ __launch_bounds__(256) void er_atrous_kernel(const float4* __restrict__ src, const float4* __restrict__ normal, float4* __restrict__ dst,
                                                         int w, int h, int step, float kc) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    const float kernel[5] = {1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};
    const float4 c = src[(size_t)y * w + x], n = normal[(size_t)y * w + x];
    float sx = 0, sy = 0, sz = 0, sw = 0;
    for (int j = -2; j <= 2; j++)
        for (int i = -2; i <= 2; i++) {
            int qx = x + i * step, qy = y + j * step;
            qx = qx < 0 ? 0 : (qx >= w ? w - 1 : qx);
            qy = qy < 0 ? 0 : (qy >= h ? h - 1 : qy);
            const float4 cq = src[(size_t)qy * w + qx], nq = normal[(size_t)qy * w + qx];
            const float dx = c.x - cq.x, dy = c.y - cq.y, dz = c.z - cq.z;
            const float d2 = dx * dx + dy * dy + dz * dz;
            const float wc = 1.0f / (1.0f + kc * d2);
            float nd = n.x * nq.x + n.y * nq.y + n.z * nq.z;
            // (pixels whose paths hit nothing have a zero normal: they only blend with each other)
            const bool none = (n.x == 0.0f && n.y == 0.0f && n.z == 0.0f), noneq = (nq.x == 0.0f && nq.y == 0.0f && nq.z == 0.0f);
            nd = (none && noneq) ? 1.0f : (nd < 0.0f ? 0.0f : nd);
            const float wgt = kernel[i + 2] * kernel[j + 2] * wc * (nd * nd);
            sx = sx + cq.x * wgt; sy = sy + cq.y * wgt; sz = sz + cq.z * wgt; sw = sw + wgt;
        }
    dst[(size_t)y * w + x] = make_float4(sx / sw, sy / sw, sz / sw, c.w);
}
void er_launch_atrous(const float4* src, const float4* normal, float4* dst, int w, int h, int step, float kc, hipStream_t stream) {
    hipLaunchKernelGGL(er_atrous_kernel, dim3((w + 15) / 16, (h + 15) / 16), dim3(256), 0, stream, src, normal, dst, w, h, step, kc);
}

void er_launch_pack(const DevScene& S, const uint32_t* tiles, uint32_t ntiles, int pass, void* dst, hipStream_t stream) {
    if (ntiles == 0) return;
    hipLaunchKernelGGL(er_pack_kernel, dim3(ntiles), dim3(64), 0, stream, S, tiles, pass, (float4*)dst);
}
void er_launch_unpack(const DevScene& S, const uint32_t* tiles, uint32_t ntiles, int pass, const void* src, hipStream_t stream) {
    if (ntiles == 0) return;
    hipLaunchKernelGGL(er_unpack_kernel, dim3(ntiles), dim3(64), 0, stream, S, tiles, pass, (const float4*)src);
}
