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
#include "er_stream.h"
#include "er_device.h"
#include "er_shade.h"

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
    for (int p = 0; p < ER_PASS_COUNT; p++) S.passes[er_pass_index(npx, p, idx)] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    S.samples[idx] = 1;
}

template <bool COUNT, bool EXT>
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
        uint32_t rs = S.rng[idx];
        uint32_t sa = S.samples[idx];

        uint32_t s = 0;
        uint32_t bounce = 0;
        float prev_pdf = -1.0f;
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
                prev_pdf = -1.0f;
                fresh = false;
            }
            // ---- one iteration of the bounce loop, src/kernel.cpp:508-593 (er_shade.h), rays traced inline ----
            c_bounce++;
            float dist;
            c_rays++;
            const int slot = trace<COUNT, false>(S, stack, ray, -1, __builtin_inff(), dist, c_nodes, c_tris);
            const int hslot = slot;
            bool done = false, pending = false, lpending = false;
            // the hooks trace the shadow rays at once through the exact binary-BVH routine and add the selected contribution
#define ER_BOUNCE_HDRI_QUERY(sr, self_slot, d_self, cv, co)                                                      \
    {   /* occluded iff the closest hit is another triangle (src/kernel.cpp:555-562) */                           \
        float sd_;                                                                                                \
        c_rays++;                                                                                                 \
        const int occ_ = trace<COUNT, true>(S, stack, (sr), (self_slot), (d_self), sd_, c_nodes, c_tris);         \
        light = light + (occ_ >= 0 ? (co) : (cv));                                                                \
    }
#define ER_BOUNCE_LIGHT_QUERY(lr, limit, lv, lo)                                                                 \
    {   /* point-light sample (ER_FLAG_POINT_LIGHTS): occluded iff a hit is nearer than the light */              \
        float sd_;                                                                                                \
        c_rays++;                                                                                                 \
        const int occ_ = trace<COUNT, true>(S, stack, (lr), -1, (limit), sd_, c_nodes, c_tris);                   \
        light = light + (occ_ >= 0 ? (lo) : (lv));                                                                \
    }
#define ER_BOUNCE_FIRST_HIT(n, t, b) aov_n = (n); aov_t = (t); aov_b = (b)
#include "er_bounce.inc"
#undef ER_BOUNCE_HDRI_QUERY
#undef ER_BOUNCE_LIGHT_QUERY
#undef ER_BOUNCE_FIRST_HIT
            (void)pending; (void)lpending;
            if (done) {
                sa = accumulate_sample(S, idx, sa, light, aov_n, aov_t, aov_b);   // src/kernel.cpp:597-645
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
    if (px < S.x_res && py < S.y_res) v = S.passes[er_pass_index((size_t)S.x_res * S.y_res, pass, (size_t)py * S.x_res + px)];
    dst[(size_t)blockIdx.x * 64 + lane] = v;
}
__global__ __launch_bounds__(64) void er_unpack_kernel(DevScene S, const uint32_t* tiles, int pass, const float4* src) {
    const int lane = threadIdx.x;
    const uint32_t tile = tiles[blockIdx.x];
    const uint32_t tx = tile % S.tiles_x, ty = tile / S.tiles_x;
    const uint32_t px = tx * ER_TILE + (lane & 7), py = ty * ER_TILE + (lane >> 3);
    if (px < S.x_res && py < S.y_res)
        S.passes[er_pass_index((size_t)S.x_res * S.y_res, pass, (size_t)py * S.x_res + px)] = src[(size_t)blockIdx.x * 64 + lane];
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

hipError_t er_probe_kernels(const char** which) {
    hipFuncAttributes a;
    hipError_t e;
    *which = "er_setup_kernel";
    if ((e = hipFuncGetAttributes(&a, (const void*)er_setup_kernel)) != hipSuccess) return e;
    *which = "er_render_kernel";
    if ((e = hipFuncGetAttributes(&a, (const void*)er_render_kernel<false, false>)) != hipSuccess) return e;
    if ((e = er_probe_wavefront(which)) != hipSuccess) return e;
    if ((e = er_probe_stream(which)) != hipSuccess) return e;
    if ((e = er_probe_gpu_build(which)) != hipSuccess) return e;
    *which = nullptr;
    return hipSuccess;
}

void er_launch_setup(const DevScene& S, hipStream_t stream) {
    uint32_t npx = S.x_res * S.y_res;
    if (npx == 0) return;
    hipLaunchKernelGGL(er_setup_kernel, dim3((npx + 255) / 256), dim3(256), 0, stream, S);
}
void er_launch_render(const DevScene& S, uint32_t n_samples, bool count, hipStream_t stream) {
    if (S.owned_tile_count == 0 || n_samples == 0) return;
    const bool ext = er_ext_active(S);
    auto k = count ? (ext ? er_render_kernel<true, true> : er_render_kernel<true, false>) : (ext ? er_render_kernel<false, true> : er_render_kernel<false, false>);
    hipLaunchKernelGGL(k, dim3(S.owned_tile_count), dim3(64), 0, stream, S, n_samples);
}
// ---- denoise: edge-avoiding a-trous wavelet filter of the BEAUTY plane, guided by colour and
// by the first-bounce NORMAL plane, into the DENOISE plane that the reference allocates but never writes (reference
// src/kernel.cpp:604; its host-side `get_pass denoise` calls OIDN, src/Managers.cpp:319-343, a neural filter that is
// not on this path and has no arithmetic to match).  One launch per level; level k taps a 5x5 B3-spline stencil with
// holes of 2^k pixels (Dammertz et al. 2010).  Edge-stopping weights are rational -- w = 1 / (1 + kc |dc|^2) and
// max(0, n.n')^2 -- so the filter is plain IEEE arithmetic (tests/test_gpu_denoise.py replays it in numpy, bit for bit).
// (src and normal are read with a stride in float4 units: 4 for a pass of the interleaved block of DevScene::passes, 1 for a plane)
__global__ __launch_bounds__(256) void er_atrous_kernel(const float4* __restrict__ src, int src_stride, const float4* __restrict__ normal, int normal_stride,
                                                         float4* __restrict__ dst, int w, int h, int step, float kc) {
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    const float kernel[5] = {1.0f / 16.0f, 1.0f / 4.0f, 3.0f / 8.0f, 1.0f / 4.0f, 1.0f / 16.0f};
    const float4 c = src[((size_t)y * w + x) * src_stride], n = normal[((size_t)y * w + x) * normal_stride];
    float sx = 0, sy = 0, sz = 0, sw = 0;
    for (int j = -2; j <= 2; j++)
        for (int i = -2; i <= 2; i++) {
            int qx = x + i * step, qy = y + j * step;
            qx = qx < 0 ? 0 : (qx >= w ? w - 1 : qx);
            qy = qy < 0 ? 0 : (qy >= h ? h - 1 : qy);
            const float4 cq = src[((size_t)qy * w + qx) * src_stride], nq = normal[((size_t)qy * w + qx) * normal_stride];
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
void er_launch_atrous(const float4* src, int src_stride, const float4* normal, int normal_stride, float4* dst, int w, int h, int step, float kc, hipStream_t stream) {
    hipLaunchKernelGGL(er_atrous_kernel, dim3((w + 15) / 16, (h + 15) / 16), dim3(256), 0, stream, src, src_stride, normal, normal_stride, dst, w, h, step, kc);
}

// one pass of every pixel as a plane (the ABI's view; er_read_pass)
__global__ __launch_bounds__(256) void er_plane_kernel(DevScene S, int pass, float4* __restrict__ dst) {
    const size_t npx = (size_t)S.x_res * S.y_res, idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < npx) dst[idx] = S.passes[er_pass_index(npx, pass, idx)];
}
void er_launch_plane(const DevScene& S, int pass, float4* dst, hipStream_t stream) {
    const size_t npx = (size_t)S.x_res * S.y_res;
    if (npx == 0) return;
    hipLaunchKernelGGL(er_plane_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, stream, S, pass, dst);
}

void er_launch_pack(const DevScene& S, const uint32_t* tiles, uint32_t ntiles, int pass, void* dst, hipStream_t stream) {
    if (ntiles == 0) return;
    hipLaunchKernelGGL(er_pack_kernel, dim3(ntiles), dim3(64), 0, stream, S, tiles, pass, (float4*)dst);
}
void er_launch_unpack(const DevScene& S, const uint32_t* tiles, uint32_t ntiles, int pass, const void* src, hipStream_t stream) {
    if (ntiles == 0) return;
    hipLaunchKernelGGL(er_unpack_kernel, dim3(ntiles), dim3(64), 0, stream, S, tiles, pass, (const float4*)src);
}
