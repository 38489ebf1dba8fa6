// er_debug.hip -- inspection and measurement kernels (include/eleven_hip_debug.h).  Not on the production path; they
// run the production path's FUNCTIONS on inputs the tests choose, so that a mismatch against the oracle can be localised
// to a ray or to a bounce (SURVEY.md section 4 levels 1-2; reference src/BVH.cpp:63-120, src/kernel.cpp:508-592).
#include "er_debug.h"

#include "er_device.h"
#include "er_shade.h"
#include "er_trav.h"

using namespace erd;

// ---- arbitrary rays through the production traversal ----
__global__ __launch_bounds__(64) void er_debug_trace_kernel(DevScene S, const float* __restrict__ o, const float* __restrict__ d, uint32_t n,
                                                             const int32_t* __restrict__ self, const float* __restrict__ limit,
                                                             int32_t* __restrict__ tri_out, int32_t* __restrict__ slot_out, float* __restrict__ pos_out,
                                                             float* __restrict__ dist_out, int32_t* __restrict__ info_out, uint2* spill_base) {
    __shared__ uint2 s_stack[WF_LDS_STACK * 64];
    __shared__ int s_stack2[ER_STACK * 64];
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    uint2* stack = s_stack + threadIdx.x;
    uint2* spill = spill_base + (size_t)blockIdx.x * (ER_STACK8 * 64) + threadIdx.x;
    int* stack2 = s_stack2 + threadIdx.x;
    Ray ray;
    ray.o = f3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
    ray.d = f3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    unsigned cn = 0, ct = 0;
    int info = 0;
    if (self) {
        const bool occ = trav_run_shadow<false>(S, stack, spill, stack2, ray, self[i], limit[i], info, cn, ct);
        tri_out[i] = occ ? 1 : 0;
        slot_out[i] = -1;
        pos_out[3 * i] = pos_out[3 * i + 1] = pos_out[3 * i + 2] = 0.0f;
        dist_out[i] = 0.0f;
    } else {
        const int slot = trav_run_closest<false>(S, stack, spill, stack2, ray, __builtin_inff(), info, cn, ct);
        tri_out[i] = -1;
        slot_out[i] = slot;
        pos_out[3 * i] = pos_out[3 * i + 1] = pos_out[3 * i + 2] = 0.0f;
        dist_out[i] = __builtin_inff();
        if (slot >= 0) {
            HitFull h;
            full_hit(S, (uint32_t)slot, ray, h);
            tri_out[i] = __builtin_bit_cast(int, S.tri_isect[(size_t)slot * 3].w);
            pos_out[3 * i] = h.position.x; pos_out[3 * i + 1] = h.position.y; pos_out[3 * i + 2] = h.position.z;
            dist_out[i] = length(h.position - ray.o);
        }
    }
    info_out[i] = info;
}

void er_launch_debug_trace(const DevScene& S, const float* o, const float* d, uint32_t n, const int32_t* self, const float* limit, int32_t* tri,
                           int32_t* slot, float* pos, float* dist, int32_t* info, void* spill, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(er_debug_trace_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, S, o, d, n, self, limit, tri, slot, pos, dist, info, (uint2*)spill);
}

// ---- per-bounce trace of one pixel-sample: er_bounce.inc over the production traversal, queries traced at once ----
template <bool EXT>
__global__ __launch_bounds__(64) void er_debug_pixel_kernel(DevScene S, uint32_t idx, ErTraceRec* recs, int max_recs, int* count, uint2* spill) {
    // (one lane, speed irrelevant.  The first levels of the traversal stack live in LDS as in the production kernels -- trav_choose
    // addresses them as LDS -- the deeper levels and the exact re-trace's stack in the HBM scratch buffer)
    __shared__ uint2 sh_stack[WF_LDS_STACK * 64];
    uint2* s_stack = sh_stack;
    int* s_stack2 = (int*)(spill + (size_t)ER_STACK8 * 64);
    static_assert((size_t)ER_STACK8 * 64 + ((size_t)ER_STACK * 64 * sizeof(int) + sizeof(uint2) - 1) / sizeof(uint2) <= (size_t)ER_DEBUG_PIXEL_SCRATCH,
                  "er_debug_trace_pixel's scratch must hold the spill levels and the exact routine's int stack");
    if (threadIdx.x != 0) return;
    int nrec = 0;
    unsigned c_rays = 0, c_nodes = 0, c_tris = 0, c_shaded = 0, c_texels = 0, c_hdri = 0;
    uint32_t rs = S.rng[idx];
    const uint32_t px = idx % S.x_res, py = idx / S.x_res;
    float c1 = rng_next(rs), c2 = rng_next(rs), c3 = rng_next(rs), c4 = rng_next(rs), c5 = rng_next(rs);
    Ray ray = camera_ray(S.cam, (int)px, (int)py, S.x_res, S.y_res, c1, c2, c3, c4, c5);
    F3 light = f3s(0), reduction = f3s(1), aov_n = f3s(0), aov_t = f3s(0), aov_b = f3s(0);
    uint32_t bounce = 0;
    float prev_pdf = -1.0f;
    while (true) {
        int info;
        c_rays++;
        const int hslot = trav_run_closest<false>(S, s_stack, spill, s_stack2, ray, __builtin_inff(), info, c_nodes, c_tris);
        ErTraceRec* rec = nrec < max_recs ? &recs[nrec++] : nullptr;
        if (rec) {
            rec->bounce = (int32_t)bounce;
            rec->tri = hslot < 0 ? -1 : __builtin_bit_cast(int, S.tri_isect[(size_t)hslot * 3].w);
            rec->shadow_tri = -1; rec->opaque = 0; rec->shadow_occ = -1; rec->light_occ = -1;
            for (int k = 0; k < 3; k++) { rec->position[k] = 0; rec->wi[k] = 0; }
            if (hslot >= 0) {
                HitFull h;
                full_hit(S, (uint32_t)hslot, ray, h);
                rec->position[0] = h.position.x; rec->position[1] = h.position.y; rec->position[2] = h.position.z;
            }
        }
        constexpr bool COUNT = false;
        const Ray traced = ray;
        const uint32_t rs_before = rs;
        bool done = false, pending = false, lpending = false;
#define ER_BOUNCE_HDRI_QUERY(sr, self_slot, d_self, cv, co)                                                                          \
    {                                                                                                                                 \
        int info_;                                                                                                                    \
        c_rays++;                                                                                                                     \
        const bool occ_ = trav_run_shadow<false>(S, s_stack, spill, s_stack2, (sr), (self_slot), (d_self), info_, c_nodes, c_tris);   \
        light = light + (occ_ ? (co) : (cv));                                                                                         \
        if (rec) {                                                                                                                    \
            rec->shadow_occ = occ_ ? 1 : 0;                                                                                           \
            /* what the reference's throwRay(shadowRay) returns (src/kernel.cpp:556), for the record only */                          \
            const int cs_ = trav_run_closest<false>(S, s_stack, spill, s_stack2, (sr), __builtin_inff(), info_, c_nodes, c_tris);     \
            rec->shadow_tri = cs_ < 0 ? -1 : __builtin_bit_cast(int, S.tri_isect[(size_t)cs_ * 3].w);                                 \
        }                                                                                                                             \
    }
#define ER_BOUNCE_LIGHT_QUERY(lr, limit, lv, lo)                                                                                     \
    {                                                                                                                                 \
        int info_;                                                                                                                    \
        c_rays++;                                                                                                                     \
        const bool occ_ = trav_run_shadow<false>(S, s_stack, spill, s_stack2, (lr), -1, (limit), info_, c_nodes, c_tris);             \
        light = light + (occ_ ? (lo) : (lv));                                                                                         \
        if (rec) rec->light_occ = occ_ ? 1 : 0;                                                                                       \
    }
#define ER_BOUNCE_FIRST_HIT(n, t, b) aov_n = (n); aov_t = (t); aov_b = (b)
#include "er_bounce.inc"
#undef ER_BOUNCE_HDRI_QUERY
#undef ER_BOUNCE_LIGHT_QUERY
#undef ER_BOUNCE_FIRST_HIT
        (void)pending; (void)lpending; (void)traced;
        if (rec) {
            // the opacity test passed iff more than its one draw was taken (src/kernel.cpp:539: the opaque branch draws 4+)
            uint32_t r1 = rs_before;
            (void)rng_next(r1);
            rec->opaque = (hslot >= 0 && rs != r1) ? 1 : 0;
            if (hslot >= 0) { rec->wi[0] = ray.d.x; rec->wi[1] = ray.d.y; rec->wi[2] = ray.d.z; }
            rec->light[0] = light.x; rec->light[1] = light.y; rec->light[2] = light.z;
            rec->reduction[0] = reduction.x; rec->reduction[1] = reduction.y; rec->reduction[2] = reduction.z;
        }
        if (done) break;
    }
    const uint32_t sa = S.samples[idx];
    const uint32_t sa2 = accumulate_sample(S, idx, sa, light, aov_n, aov_t, aov_b);
    if (sa2 != sa) S.samples[idx] = sa2;
    S.rng[idx] = rs;
    *count = nrec;
    atomicAdd(&S.counters->paths, 1ull);
    atomicAdd(&S.counters->bounce_samples, (unsigned long long)(bounce + (bounce < S.max_bounces ? 1u : 0u)));
    atomicAdd(&S.counters->rays, (unsigned long long)c_rays);
    atomicAdd(&S.counters->shaded_hits, (unsigned long long)c_shaded);
    atomicAdd(&S.counters->hdri_samples, (unsigned long long)c_hdri);
}

void er_launch_debug_pixel(const DevScene& S, uint32_t idx, ErTraceRec* recs, int max_recs, int* count, void* spill, hipStream_t stream) {
    if (er_ext_active(S)) hipLaunchKernelGGL(er_debug_pixel_kernel<true>, dim3(1), dim3(64), 0, stream, S, idx, recs, max_recs, count, (uint2*)spill);
    else hipLaunchKernelGGL(er_debug_pixel_kernel<false>, dim3(1), dim3(64), 0, stream, S, idx, recs, max_recs, count, (uint2*)spill);
}

// ---- the device functions of the path, one item per thread (known-answer tests against the oracle's entry points) ----
namespace {
ERD HitData hd_from(const float* h) {     // the oracle's hd[20] layout
    HitData d;
    d.metallic = h[0]; d.roughness = h[1]; d.clearcoatGloss = h[2]; d.clearcoat = h[3]; d.anisotropic = h[4];
    d.transmission = h[5]; d.specular = h[6]; d.specularTint = h[7]; d.sheenTint = h[8]; d.subsurface = h[9];
    d.sheen = h[10]; d.opacity = 1.0f;
    d.albedo = f3(h[11], h[12], h[13]); d.tangent = f3(h[14], h[15], h[16]); d.bitangent = f3(h[17], h[18], h[19]);
    d.emission = f3s(0); d.position = f3s(0); d.normal = f3s(0);
    d.gtr1_log = __builtin_nanf("");      // (no material behind a hand-made HitData: GTR1 evaluates its logarithm itself)
    return d;
}
}  // namespace

__global__ __launch_bounds__(64) void er_debug_eval_kernel(DevScene S, int kind, const float* __restrict__ in, uint32_t n, uint32_t in_stride,
                                                           float* __restrict__ out, uint32_t out_stride) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float* a = in + (size_t)i * in_stride;
    float* o = out + (size_t)i * out_stride;
    auto bits = [](float f) { return __builtin_bit_cast(int, f); };
    auto fbits = [](uint32_t u) { return __builtin_bit_cast(float, u); };
    switch (kind) {
        case ER_FN_RNG: {
            uint32_t st = jenkins_u32((uint32_t)bits(a[0]) + 1u);      // RngGenerator(idx), src/kernel.cpp:38-40,183
            for (int k = 0; k < 16; k++) { o[k] = rng_next(st); o[16 + k] = fbits(st); }
            break;
        }
        case ER_FN_CAMERA_RAY: {
            const Ray r = camera_ray(S.cam, (int)a[0], (int)a[1], S.x_res, S.y_res, a[2], a[3], a[4], a[5], a[6]);
            o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
            break;
        }
        case ER_FN_TRI_HIT: {
            const int id = bits(a[0]);
            int slot = -1;
            for (uint32_t s = 0; s < S.tri_count; s++)
                if (__builtin_bit_cast(int, S.tri_isect[(size_t)s * 3].w) == id) { slot = (int)s; break; }
            Ray ray;
            ray.o = f3(a[1], a[2], a[3]);
            ray.d = f3(a[4], a[5], a[6]);
            for (int k = 0; k < 18; k++) o[k] = 0.0f;
            if (slot < 0) break;
            F3 v0, v1, v2;
            float4 qa, qb, qc;
            load_verts(S, (uint32_t)slot, v0, v1, v2, qa, qb, qc);
            float u, v, t;
            if (!tri_mt(v0, v1, v2, ray, u, v, t)) break;
            HitFull h;
            full_hit(S, (uint32_t)slot, ray, h);
            o[0] = 1.0f;
            const F3 vs[5] = {h.position, h.normal, h.gnormal, h.tangent, h.bitangent};
            for (int k = 0; k < 5; k++) { o[1 + 3 * k] = vs[k].x; o[2 + 3 * k] = vs[k].y; o[3 + 3 * k] = vs[k].z; }
            o[16] = h.tu; o[17] = h.tv;
            break;
        }
        case ER_FN_DISNEY_EVAL: {
            const F3 r = DisneyEval(hd_from(a), f3(a[20], a[21], a[22]), f3(a[23], a[24], a[25]), f3(a[26], a[27], a[28]));
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case ER_FN_DISNEY_PDF:
            o[0] = DisneyPdf(hd_from(a), f3(a[20], a[21], a[22]), f3(a[23], a[24], a[25]), f3(a[26], a[27], a[28]));
            break;
        case ER_FN_DISNEY_SAMPLE: {
            const F3 r = DisneySample(hd_from(a), f3(a[20], a[21], a[22]), f3(a[23], a[24], a[25]), a[26], a[27], a[28]);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case ER_FN_SPHERICAL:
            spherical_mapping(f3(a[0], a[1], a[2]), o[0], o[1]);
            break;
        case ER_FN_REV_SPHERICAL: {
            const F3 r = reverse_spherical_mapping(a[0], a[1]);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case ER_FN_TEXTURE: {
            const int id = bits(a[0]);
            const DevTex t = id < 0 ? S.hdri_tex : S.textures[id];
            const F3 r = a[3] != 0.0f ? tex_filtered(S, t, a[1], a[2]) : tex_uv(S, t, a[1], a[2]);
            o[0] = r.x; o[1] = r.y; o[2] = r.z;
            break;
        }
        case ER_FN_HDRI_SEARCH:
            o[0] = fbits((uint32_t)er_cdf_search(S.hdri_cdf, S.hdri_tex.width * S.hdri_tex.height, S.hdri_guide, S.hdri_buckets, a[0]));
            break;
        case ER_FN_HDRI_PDF:
            o[0] = hdri_pdf(S, bits(a[0]), bits(a[1]));
            break;
        case ER_FN_MATH: {
            const int op = bits(a[0]);
            const float x = a[1], y = a[2];
            o[0] = op == 0 ? ermath::er_sin(x) : op == 1 ? ermath::er_cos(x) : op == 2 ? ermath::er_acos(x) : op == 3 ? ermath::er_log(x)
                 : op == 4 ? ermath::er_pow(x, y) : ermath::er_atan2(x, y);
            break;
        }
        default: break;
    }
}
void er_launch_debug_eval(const DevScene& S, int kind, const float* in, uint32_t n, uint32_t in_stride, float* out, uint32_t out_stride, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(er_debug_eval_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, S, kind, in, n, in_stride, out, out_stride);
}

// ---- streaming kernels for the measured HBM peak ("measured peak from a device-to-device copy / triad
// kernel run in the same job").  Grid-stride over float4, 4 independent 16-byte accesses in flight per lane. ----
typedef float VF4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void er_hbm_copy_kernel(const VF4* __restrict__ src, VF4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const VF4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        const VF4 c = __builtin_nontemporal_load(src + i + 2 * stride), e = __builtin_nontemporal_load(src + i + 3 * stride);
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + stride);
        __builtin_nontemporal_store(c, dst + i + 2 * stride);
        __builtin_nontemporal_store(e, dst + i + 3 * stride);
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void er_hbm_read_kernel(const VF4* __restrict__ src, float* __restrict__ sink, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.0f;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const VF4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        const VF4 c = __builtin_nontemporal_load(src + i + 2 * stride), e = __builtin_nontemporal_load(src + i + 3 * stride);
        acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (e.x + e.y + e.z + e.w);
    }
    for (; i < n; i += stride) { const VF4 a = src[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 1.2345e-30f) sink[0] = acc;     // (never true for the probe's data: keeps the loads alive)
}
void er_launch_hbm_copy(const float4* src, float4* dst, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(er_hbm_copy_kernel, dim3(256 * 16), dim3(256), 0, stream, (const VF4*)src, (VF4*)dst, n);
}
void er_launch_hbm_read(const float4* src, float* sink, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(er_hbm_read_kernel, dim3(256 * 16), dim3(256), 0, stream, (const VF4*)src, sink, n);
}
