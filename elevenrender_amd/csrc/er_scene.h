// er_scene.h -- host-side state behind the opaque ErScene handle of include/eleven_hip.h, and the small helpers the
// translation units of the boundary share (er_api.cpp: the drop-in entry points; er_debug_api.cpp: inspection hooks of
// include/eleven_hip_debug.h; er_collective.cpp: the RCCL framebuffer combine).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/eleven_hip.h"
#include "er_bvh.h"
#include "er_device.h"
#include "er_gpu_build.h"
#include "er_kernels.h"
#include "er_wavefront.h"
#include "er_stream.h"

namespace erh {


inline thread_local std::string g_err;

inline int fail(int code, const std::string& msg) noexcept {
    try { g_err = msg; } catch (...) { /* keeping the previous text beats unwinding through the C ABI */ }
    return code;
}

// Every extern "C" entry point runs its body through this: the header promises "never throws", and the bodies
// allocate (std::vector, std::string, std::map) -- a std::bad_alloc must come back as ER_ERR_OOM, not unwind
// through the C ABI into a host that may not even be C++ (reference: errors are logged and never propagated,
// src/Managers.cpp:244-246).
template <class F>
int guarded(const char* who, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(ER_ERR_OOM, std::string(who) + ": out of host memory");
    } catch (const std::exception& e) {
        return fail(ER_ERR_STATE, std::string(who) + ": " + e.what());
    } catch (...) {
        return fail(ER_ERR_STATE, std::string(who) + ": unknown exception");
    }
}

// Test hook (include/eleven_hip_debug.h, er_debug_set_host_alloc_limit): the library's large host allocations
// announce their size here first, so the out-of-memory path can be exercised without exhausting the machine.
inline std::atomic<uint64_t> g_host_alloc_limit{0};
inline void host_reserve(uint64_t bytes) {
    const uint64_t lim = g_host_alloc_limit.load();
    if (lim && bytes > lim) throw std::bad_alloc();
}

struct EventPair {      // two timing events that do not outlive the function that made them
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail(e__ == hipErrorOutOfMemory ? ER_ERR_OOM : ER_ERR_HIP,                           \
                        std::string(#expr) + ": " + hipGetErrorString(e__));                            \
    } while (0)

struct HostTex {
    int32_t width, height, channels, filter;
    std::vector<float> data;
};

// Texture::getValueFromCoordinates, reference src/Texture.cpp:172-200 (host copy, used by the CDF)
inline void host_tex_coords(const HostTex& t, int x, int y, float out[3]) {
    x %= t.width;
    y %= t.height;
    if (x < 0) x *= -1;
    if (y < 0) y *= -1;
    out[0] = out[1] = out[2] = 0.0f;
    const float* d = t.data.data();
    if (t.channels == 1) {
        out[0] = out[1] = out[2] = d[y * t.width + x];
    } else if (t.channels == 2) {
        out[0] = d[t.channels * (y * t.width + x) + 0];
        out[1] = d[t.channels * (y * t.width + x) + 1];
    } else if (t.channels >= 3) {
        out[0] = d[t.channels * (y * t.width + x) + 0];
        out[1] = d[t.channels * (y * t.width + x) + 1];
        out[2] = d[t.channels * (y * t.width + x) + 2];
    }
}

// HDRI::generateCDF, reference src/HDRI.cpp:62-83 (host preparation, same float sequence)
inline void host_generate_cdf(const HostTex& t, std::vector<float>& cdf, float& radianceSum) {
    int c = 0;
    radianceSum = 0;
    cdf.assign((size_t)t.width * t.height + 1, 0.0f);
    float p[3];
    for (int j = 0; j < t.height; j++)
        for (int i = 0; i < t.width; i++) {
            host_tex_coords(t, i, j, p);
            radianceSum += p[0] + p[1] + p[2];
        }
    for (int j = 0; j < t.height; j++)
        for (int i = 0; i < t.width; i++) {
            host_tex_coords(t, i, j, p);
            cdf[c + 1] = cdf[c] + (p[0] + p[1] + p[2]) / radianceSum;
            c++;
        }
}

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

template <class T>
struct ScopedDevBuf : DevBuf<T> {   // a temporary: freed on every way out of the function
    ScopedDevBuf() = default;
    ScopedDevBuf(const ScopedDevBuf&) = delete;
    ScopedDevBuf& operator=(const ScopedDevBuf&) = delete;
    ~ScopedDevBuf() { this->release(); }
};


}  // namespace erh

using erh::DevBuf;
using erh::HostTex;

struct ErScene {
    // host copy of the description
    uint32_t tri_count = 0;
    std::vector<float> vertices, normals, tangents, uvs, tangent_sign;
    std::vector<int32_t> material_id;
    std::vector<ErMaterial> materials;
    std::vector<HostTex> textures;
    HostTex hdri_tex;
    std::vector<float> hdri_cdf;
    float hdri_radiance_sum = 0;
    ErCamera camera;
    std::vector<ErPointLight> point_lights;
    uint32_t x_res = 0, y_res = 0;

    // render state
    bool begun = false;
    int device = 0;
    ErRenderParams params{};
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timing_open = false;
    DevScene dev{};
    ErAccelInfo accel{};
    DevBuf<float4> d_nodes, d_nodes8, d_isect, d_attr, d_passes;
    DevBuf<float4> d_plane;      // staging: one pass gathered as a plane for er_read_pass
    DevBuf<ErMaterial> d_materials;
    DevBuf<float4> d_mat_pre;    // DevScene::mat_pre
    DevBuf<ErPointLight> d_lights;
    DevBuf<DevTex> d_textures;
    DevBuf<float> d_tex_pool, d_cdf;
    DevBuf<uint32_t> d_samples, d_rng, d_owned;
    DevBuf<DevCounters> d_counters;
    std::vector<uint8_t> tex_mode;   // per texture, how the device table holds it: 0 as it came, 1 first channel alone, 2 first channel alone to the power 2.2 (er_render_begin)
    bool fused_any = false;          // some material's textures are fused
    DevBuf<DevFused> d_mat_fused;   // per material: its fused albedo / roughness / metallic texels in d_tex_pool, or width 0
    DevBuf<DevScene> d_dev;      // device copy of `dev`: the streaming kernel reads the scene descriptor through a pointer
    DevBuf<float4> d_wf4;        // 11 float4 arrays of the wavefront state, back to back
    DevBuf<uint32_t> d_wf1;      // hit, left, occluded, 4 queues, counts
    DevBuf<uint2> d_spill;
    DevBuf<uint32_t> d_guide, d_ticket, d_deal;      // d_deal: the streaming schedule's deal of tiles to workgroups (er_stream_deal_tiles)
    uint32_t stream_blocks = 0, stream_tracers = 0, stream_waves = 16, stream_ring_cap = 0;     // streaming schedule (er_stream.hip): workgroups; tracer waves of the 16
    bool stream_lights = false;                         //   slot records carry the point-light query's line
    uint32_t* stream_ctl = nullptr;                     //   [0] pixel ticket, [1] status word, [2..3] tracer iterations, [4..5] busy tracer lanes of the last call, [6..7] its start, [8..23] its end per XCD (100 MHz)
    bool stream_adapt = false;                          //   move a wave between the roles by how full the tracer lanes were (er_stream_adapt)
    double stream_busy = 0.0;                           //   tracer lanes that held a ray, last completed call
    double stream_launch_ms = 0.0;                      //   device time of that call's launch (start stamp to the last XCD's end stamp)
    uint32_t stream_low_streak = 0, stream_up_budget = 1, stream_tracers_start = 0, stream_readings = 0;   //   er_stream_adapt: consecutive low readings; steps back up left; the split the render began with
    uint32_t stream_deal_off = 0, stream_deal_n = 0;    //   the deal in use inside d_deal (entries): the one of large super-tiles first, the default edge's after it
    uint32_t stream_deal_alt_off = 0, stream_deal_alt_n = 0;   //   the deal of large screen regions beside it (0 entries: none)
    uint64_t stream_spec[3] = {0, 0, 0}, stream_spec_seen = 0;   //   speculative samples started / whose guess was right / wrong, summed over the render's completed launches (small shares: er_stream.hip ST_PRED_BIT)
    bool stream_spec_form = false;                      //   launch the kernel's form with speculative samples (small shares)
    bool stream_keep = false;                           //   a pixel that is behind its workgroup's most advanced one keeps its slot (er_stream.hip s_front)
    bool stream_probe_launch = false;                   //   the launch just completed was the first sample of a render's first call, run alone to decide the deal (er_render_samples)
    bool stream_deal_pending = false;                   //   the first completed call decides between the two (er_stream_adapt), from ...
    DevBuf<uint32_t> d_px_draws;                         //   ... DevScene::px_draws
    DevBuf<uint32_t> d_tile_cost;                       //   ... DevScene::tile_cost: per tile of the frame, the summed path lengths of its finished samples
    std::vector<uint32_t> stream_deal_large;            //   host copy of the large deal until then (which XCD gets which tile under it)
    double stream_cost_spread = -1.0;                   //   (max - min) / mean of the XCDs' counted work under the large deal; < 0: not decided yet
    double stream_xcd_spread = 0.0;                     //   (latest - earliest XCD) / launch duration of the last completed call; < 0: not measured
    uint64_t stream_launches = 0, stream_adapted = 0;   //   launches enqueued / the launch whose measurements er_stream_adapt has already used
    std::vector<WfState> wf;              // slot pools (see er_render_begin)
    std::vector<hipStream_t> pool_streams;   // pool 0 runs on `stream`, pool p > 0 on pool_streams[p - 1]
    std::vector<hipEvent_t> pool_events;     // [0] fork; [p] pool p has finished

    uint32_t trace_blocks = 0, shade_blocks = 0;
    std::vector<hipEvent_t> prof_events;   // ER_FLAG_PROFILE: e[3i], e[3i+1], e[3i+2] = before trace, between, after shade
    DevBuf<uint32_t> d_ray_log;            // ER_FLAG_PROFILE, wavefront: rays found by the i-th trace launch (same order)
    size_t prof_used = 0;
    ErProfile profile{};
    std::map<uint32_t, DevBuf<uint32_t>> d_rank_tiles;   // tile lists of other ranks (for unpack)
    // er_gather_pass: the packed owned pixels of this rank (non-root) and one receive buffer per peer (root), allocated at the first
    // gather and kept for the scene's life -- a read-back of five planes from seven peers was 35 hipMalloc / hipFree pairs inside the
    // time the gather is measured by; now it is pack + wire + unpack
    DevBuf<float4> d_gather_mine;
    std::map<uint32_t, DevBuf<float4>> d_gather_in;
    // which ranks' pixels of each plane were unpacked into this scene since the last sample was enqueued: er_denoise on a
    // sharded frame needs the whole BEAUTY and NORMAL planes (the rank the frame was gathered to)
    std::set<uint32_t> unpacked[ER_PASS_COUNT];
    std::mutex mtx;

    std::vector<uint32_t> tiles_of(uint32_t rank, uint32_t world) const {
        std::vector<uint32_t> t;
        uint32_t tiles_x = (x_res + ER_TILE - 1) / ER_TILE, tiles_y = (y_res + ER_TILE - 1) / ER_TILE;
        for (uint32_t ty = 0; ty < tiles_y; ty++)
            for (uint32_t tx = 0; tx < tiles_x; tx++)
                if ((tx + ty) % world == rank) t.push_back(ty * tiles_x + tx);
        return t;
    }
    void release_device() {
        d_nodes.release(); d_nodes8.release(); d_isect.release(); d_attr.release(); d_passes.release(); d_plane.release(); d_materials.release();
        d_textures.release(); d_tex_pool.release(); d_lights.release(); d_cdf.release(); d_samples.release(); d_rng.release();
        d_owned.release(); d_counters.release(); d_wf4.release(); d_wf1.release(); d_spill.release(); d_guide.release(); d_ticket.release(); d_deal.release(); d_ray_log.release(); d_mat_fused.release(); d_mat_pre.release(); d_dev.release(); d_tile_cost.release(); d_px_draws.release();
        for (auto& kv : d_rank_tiles) kv.second.release();
        d_rank_tiles.clear();
        d_gather_mine.release();
        for (auto& kv : d_gather_in) kv.second.release();
        d_gather_in.clear();
        for (hipEvent_t e : prof_events) (void)hipEventDestroy(e);
        prof_events.clear();
        prof_used = 0;
        for (hipEvent_t e : pool_events) (void)hipEventDestroy(e);
        pool_events.clear();
        for (hipStream_t st : pool_streams) (void)hipStreamDestroy(st);
        pool_streams.clear();
        wf.clear();
        if (ev_start) (void)hipEventDestroy(ev_start);
        if (ev_stop) (void)hipEventDestroy(ev_stop);
        if (stream) (void)hipStreamDestroy(stream);
        ev_start = ev_stop = nullptr;
        stream = nullptr;
        begun = false;
        timing_open = false;
    }
};

// ER_OK unless the streaming schedule's watchdog ended a call early (er_api.cpp); call with the scene's stream idle and its
// mutex held.  Every entry point that hands planes out (read-backs, snapshot, gather, pack, denoise) ends with it.
extern "C" __attribute__((visibility("hidden"))) int er_scene_stream_status(ErScene* s, const char* who);   // (library-internal)

namespace erh {

// guide table of er_cdf.h: guide[j] = first i in [0,length] with cdf[i] >= j/buckets
inline int er_build_cdf_guide(const float* cdf, int length, std::vector<uint32_t>& guide) {
    // entries per bucket on average (buckets = a power of two >= length / per): every halving takes one dependent load off the
    // CDF search of every shading step and doubles the table (4 bytes per bucket).  C2 (2048x1024 HDRI): 8 -> 1362 / 1367,
    // 2 -> 1374 / 1375, 1 -> 1373 / 1372 Msamples/s on one box (profiles/r03_ab_cdf_buckets.log): 2 (a 4 MB table there).
    static const int per = [] { const char* e = getenv("ER_CDF_ENTRIES_PER_BUCKET"); int v = e ? atoi(e) : 2; return v < 1 ? 1 : v; }();
    int buckets = 1;
    while (buckets < length / per && buckets < (1 << 22)) buckets <<= 1;
    guide.assign((size_t)buckets + 1, (uint32_t)length);
    int i = 0;
    for (int j = 0; j <= buckets; j++) {
        float thr = (float)j / (float)buckets;
        while (i < length && cdf[i] < thr) i++;
        guide[j] = (uint32_t)i;
    }
    return buckets;
}

template <class T>
int upload(DevBuf<T>& b, const void* src, size_t count, hipStream_t s) {
    b.release();
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    HIP_TRY(hipMalloc((void**)&b.p, bytes));
    b.n = count;
    if (count && src) HIP_TRY(hipMemcpyAsync(b.p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
    return ER_OK;
}

inline int copy_tex(const ErTexture& in, HostTex& out, const char* what) {
    if (in.width <= 0 || in.height <= 0 || in.channels < 0 || (in.channels > 0 && !in.data))
        return fail(ER_ERR_INVALID_ARG, std::string("bad texture: ") + what);
    out.width = in.width; out.height = in.height; out.channels = in.channels; out.filter = in.filter;
    size_t n = (size_t)in.width * in.height * in.channels;
    out.data.assign(in.data, in.data + n);
    return ER_OK;
}


}  // namespace erh
