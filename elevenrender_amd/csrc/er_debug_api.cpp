// er_debug_api.cpp -- the inspection hooks of include/eleven_hip_debug.h (host side).  Not part of the drop-in boundary:
// they exist so that the test-suite can compare single functions of the production path with the oracle -- the traversal
// (level 1) and the per-bounce trace of one pixel-sample (level 2; reference src/kernel.cpp:508-592) --
// and exercise the boundary's failure paths.
#include "er_scene.h"
#include "er_debug.h"
#include "er_stream.h"

using namespace erh;

// ---- host-only debug hook (include/eleven_hip_debug.h) ----
#include "../../include/eleven_hip_debug.h"

static int er_debug_closest_hit_impl(ErScene* s, const float* origins, const float* dirs, uint32_t n, int32_t* tri_ids, float* positions, float* distances) {
    if (!s || !origins || !dirs || !tri_ids || !positions || !distances) return fail(ER_ERR_INVALID_ARG, "er_debug_closest_hit: NULL argument");
    std::lock_guard<std::mutex> lk(s->mtx);
    if (!s->begun) return fail(ER_ERR_STATE, "er_debug_closest_hit: er_render_begin has not succeeded");
    HIP_TRY(hipSetDevice(s->device));
    ScopedDevBuf<float> d_o, d_d, d_pos, d_dist;
    ScopedDevBuf<int32_t> d_tri;
    int rc;
    if ((rc = upload(d_o, origins, (size_t)n * 3, s->stream)) != ER_OK) return rc;
    if ((rc = upload(d_d, dirs, (size_t)n * 3, s->stream)) != ER_OK) return rc;
    if ((rc = upload(d_pos, (const float*)nullptr, (size_t)n * 3, s->stream)) != ER_OK) return rc;
    if ((rc = upload(d_dist, (const float*)nullptr, (size_t)n, s->stream)) != ER_OK) return rc;
    if ((rc = upload(d_tri, (const int32_t*)nullptr, (size_t)n, s->stream)) != ER_OK) return rc;
    er_launch_debug_hit(s->dev, d_o.p, d_d.p, n, d_tri.p, d_pos.p, d_dist.p, s->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(tri_ids, d_tri.p, (size_t)n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(positions, d_pos.p, (size_t)n * 12, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(distances, d_dist.p, (size_t)n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return ER_OK;
}

static int er_debug_cdf_search_impl(const float* cdf, int length, const float* values, int32_t* out, int count) {
    if (!cdf || !values || !out || length <= 0) return fail(ER_ERR_INVALID_ARG, "er_debug_cdf_search: bad argument");
    std::vector<uint32_t> guide;
    int buckets = er_build_cdf_guide(cdf, length, guide);
    for (int i = 0; i < count; i++) out[i] = er_cdf_search(cdf, length, guide.data(), buckets, values[i]);
    return ER_OK;
}

static int er_debug_stream_deal_impl(const uint32_t* owned, uint32_t count, uint32_t tiles_x, uint32_t blocks, int xcd_aware, uint32_t edge, uint32_t* out, uint32_t out_cap,
                                     uint32_t* most) {
    if ((count && !owned) || !tiles_x || !blocks || !most) return fail(ER_ERR_INVALID_ARG, "er_debug_stream_deal: bad argument");
    std::vector<uint32_t> deal;
    *most = er_stream_deal_tiles(owned, count, tiles_x, blocks, xcd_aware != 0, deal, edge);
    if (deal.size() > out_cap || (deal.size() && !out)) return fail(ER_ERR_INVALID_ARG, "er_debug_stream_deal: out holds fewer than blocks * most entries");
    for (size_t i = 0; i < deal.size(); i++) out[i] = deal[i];
    return ER_OK;
}

static int er_debug_bvh_check_impl(const float* vertices, const float* normals, uint32_t tri_count, int threads, ErBvhCheck* out) {
    if (!out || (tri_count && (!vertices || !normals))) return fail(ER_ERR_INVALID_ARG, "er_debug_bvh_check: NULL argument");
    ErBvhBuild b;
    er_build_bvh(vertices, normals, tri_count, threads, &b);
    memset(out, 0, sizeof(*out));
    out->node_count = (uint32_t)b.nodes.size();
    out->leaf_count = b.leaf_count;
    out->max_depth = b.max_depth;
    out->lift_bound = b.lift_bound;
    out->build_ms = (float)b.build_ms;
    std::vector<uint8_t> seen(tri_count, 0), visited(b.nodes.size(), 0);
    struct Item { int32_t ref; float lo[3], hi[3]; bool has_box; };
    std::vector<Item> stack;
    if (!b.nodes.empty()) stack.push_back(Item{0, {0, 0, 0}, {0, 0, 0}, false});
    auto area = [](const float* lo, const float* hi) {
        float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return 2.0 * ((double)x * y + (double)x * z + (double)y * z);
    };
    double root_area = 0;
    if (!b.nodes.empty()) {
        const ErNode& r = b.nodes[0];
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            lo[a] = r.c1 == ER_BVH_NO_CHILD ? r.lo0[a] : std::min(r.lo0[a], r.lo1[a]);
            hi[a] = r.c1 == ER_BVH_NO_CHILD ? r.hi0[a] : std::max(r.hi0[a], r.hi1[a]);
        }
        root_area = area(lo, hi);
    }
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        if (it.ref == ER_BVH_NO_CHILD) continue;
        if (it.ref >= 0) {
            if ((size_t)it.ref >= b.nodes.size()) { out->uncontained++; continue; }
            if (visited[it.ref]++) { out->duplicate_tris++; continue; }
            const ErNode& n = b.nodes[it.ref];
            const float* los[2] = {n.lo0, n.lo1};
            const float* his[2] = {n.hi0, n.hi1};
            const int32_t cs[2] = {n.c0, n.c1};
            for (int k = 0; k < 2; k++) {
                if (cs[k] == ER_BVH_NO_CHILD) continue;
                if (it.has_box)
                    for (int a = 0; a < 3; a++)
                        if (los[k][a] < it.lo[a] || his[k][a] > it.hi[a]) out->uncontained++;
                Item c;
                c.ref = cs[k];
                c.has_box = true;
                memcpy(c.lo, los[k], 12);
                memcpy(c.hi, his[k], 12);
                uint32_t cnt = 0;
                if (cs[k] < 0) cnt = ((uint32_t)~cs[k] & 7u) + 1;
                if (root_area > 0) out->sah_cost += area(los[k], his[k]) / root_area * (cs[k] < 0 ? cnt : 1.0);
                stack.push_back(c);
            }
        } else {
            uint32_t v = (uint32_t)~it.ref, first = v >> 3, cnt = (v & 7u) + 1;
            out->max_leaf_size = std::max(out->max_leaf_size, cnt);
            for (uint32_t i = 0; i < cnt; i++) {
                uint32_t slot = first + i;
                if (slot >= tri_count) { out->uncontained++; continue; }
                uint32_t id = b.slot_to_tri[slot];
                if (seen[id]++) out->duplicate_tris++;
                out->tris_in_leaves++;
                for (int k = 0; k < 3; k++)
                    for (int a = 0; a < 3; a++) {
                        float p = vertices[(size_t)id * 9 + k * 3 + a];
                        if (p < it.lo[a] || p > it.hi[a]) out->uncontained++;
                    }
            }
        }
    }
    for (uint8_t v : visited) if (!v) out->unreachable_nodes++;
    return ER_OK;
}

static int er_debug_trace_rays_impl(ErScene* s, const float* origins, const float* dirs, uint32_t n, const int32_t* self_slots, const float* limits,
                                    int32_t* tri_ids, int32_t* slots, float* positions, float* distances, int32_t* info) {
    if (!s || !origins || !dirs || !tri_ids || !slots || !positions || !distances || !info || (self_slots && !limits))
        return fail(ER_ERR_INVALID_ARG, "er_debug_trace_rays: NULL argument");
    std::lock_guard<std::mutex> lk(s->mtx);
    if (!s->begun) return fail(ER_ERR_STATE, "er_debug_trace_rays: er_render_begin has not succeeded");
    HIP_TRY(hipSetDevice(s->device));
    ScopedDevBuf<float> d_o, d_d, d_pos, d_dist, d_lim;
    ScopedDevBuf<int32_t> d_tri, d_slot, d_info, d_self;
    ScopedDevBuf<uint2> d_spill;
    int rc;
    if ((rc = upload(d_o, origins, (size_t)n * 3, s->stream)) != ER_OK) return rc;
    if ((rc = upload(d_d, dirs, (size_t)n * 3, s->stream)) != ER_OK) return rc;
    if (self_slots) {
        // a slot handle must be a triangle slot of this scene (or -1): the kernel indexes the records with it
        for (uint32_t i = 0; i < n; i++)
            if (self_slots[i] < -1 || self_slots[i] >= (int32_t)s->tri_count) return fail(ER_ERR_INVALID_ARG, "er_debug_trace_rays: self slot out of range");
        if ((rc = upload(d_self, self_slots, (size_t)n, s->stream)) != ER_OK) return rc;
        if ((rc = upload(d_lim, limits, (size_t)n, s->stream)) != ER_OK) return rc;
    }
    if ((rc = upload(d_pos, (const float*)nullptr, (size_t)n * 3, s->stream)) != ER_OK) return rc;
    if ((rc = upload(d_dist, (const float*)nullptr, (size_t)n, s->stream)) != ER_OK) return rc;
    if ((rc = upload(d_tri, (const int32_t*)nullptr, (size_t)n, s->stream)) != ER_OK) return rc;
    if ((rc = upload(d_slot, (const int32_t*)nullptr, (size_t)n, s->stream)) != ER_OK) return rc;
    if ((rc = upload(d_info, (const int32_t*)nullptr, (size_t)n, s->stream)) != ER_OK) return rc;
    if ((rc = upload(d_spill, (const uint2*)nullptr, (size_t)((n + 63) / 64) * ER_STACK8 * 64, s->stream)) != ER_OK) return rc;
    er_launch_debug_trace(s->dev, d_o.p, d_d.p, n, self_slots ? d_self.p : nullptr, self_slots ? d_lim.p : nullptr, d_tri.p, d_slot.p, d_pos.p,
                          d_dist.p, d_info.p, d_spill.p, s->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(tri_ids, d_tri.p, (size_t)n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(slots, d_slot.p, (size_t)n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(info, d_info.p, (size_t)n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(positions, d_pos.p, (size_t)n * 12, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(distances, d_dist.p, (size_t)n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return ER_OK;
}

static int er_debug_trace_pixel_impl(ErScene* s, uint32_t idx, ErTraceRec* recs, int max_recs, int* count) {
    if (!s || !recs || !count || max_recs <= 0) return fail(ER_ERR_INVALID_ARG, "er_debug_trace_pixel: bad argument");
    std::lock_guard<std::mutex> lk(s->mtx);
    if (!s->begun) return fail(ER_ERR_STATE, "er_debug_trace_pixel: er_render_begin has not succeeded");
    if (idx >= s->x_res * s->y_res) return fail(ER_ERR_INVALID_ARG, "er_debug_trace_pixel: pixel index out of range");
    HIP_TRY(hipSetDevice(s->device));
    ScopedDevBuf<ErTraceRec> d_recs;
    ScopedDevBuf<int> d_count;
    ScopedDevBuf<uint2> d_spill;
    int rc;
    if ((rc = upload(d_recs, (const ErTraceRec*)nullptr, (size_t)max_recs, s->stream)) != ER_OK) return rc;
    if ((rc = upload(d_count, (const int*)nullptr, 1, s->stream)) != ER_OK) return rc;
    if ((rc = upload(d_spill, (const uint2*)nullptr, (size_t)ER_DEBUG_PIXEL_SCRATCH, s->stream)) != ER_OK) return rc;
    er_launch_debug_pixel(s->dev, idx, d_recs.p, max_recs, d_count.p, d_spill.p, s->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(count, d_count.p, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (*count > 0) HIP_TRY(hipMemcpy(recs, d_recs.p, sizeof(ErTraceRec) * (size_t)*count, hipMemcpyDeviceToHost));
    return ER_OK;
}

static int er_debug_eval_impl(ErScene* s, int kind, const float* in, uint32_t n, uint32_t in_stride, float* out, uint32_t out_stride) {
    static const uint32_t need_in[ER_FN_COUNT] = {1, 7, 7, 29, 29, 29, 3, 2, 4, 1, 2, 3};
    static const uint32_t need_out[ER_FN_COUNT] = {32, 6, 18, 3, 1, 3, 2, 3, 3, 1, 1, 1};
    if (!s || !in || !out) return fail(ER_ERR_INVALID_ARG, "er_debug_eval: NULL argument");
    if (kind < 0 || kind >= ER_FN_COUNT) return fail(ER_ERR_INVALID_ARG, "er_debug_eval: unknown kind");
    if (in_stride < need_in[kind] || out_stride < need_out[kind]) return fail(ER_ERR_INVALID_ARG, "er_debug_eval: stride smaller than the kind's item");
    std::lock_guard<std::mutex> lk(s->mtx);
    if (!s->begun) return fail(ER_ERR_STATE, "er_debug_eval: er_render_begin has not succeeded");
    if (kind == ER_FN_TEXTURE)
        for (uint32_t i = 0; i < n; i++) {
            int32_t id;
            memcpy(&id, &in[(size_t)i * in_stride], 4);
            if (id < -1 || id >= (int32_t)s->textures.size()) return fail(ER_ERR_INVALID_ARG, "er_debug_eval: texture id out of range");
            // The device table may hold a scalar-only texture with its first channel alone, possibly already to the power 2.2
            // (er_render_begin): a fetch from that entry is not a fetch from the scene's texture, so it is refused, not answered.
            if (id >= 0 && (size_t)id < s->tex_mode.size() && s->tex_mode[(size_t)id] != 0)
                return fail(ER_ERR_STATE, "er_debug_eval: texture " + std::to_string(id) + " is kept compacted on the device (first channel" +
                                              (s->tex_mode[(size_t)id] == 2 ? ", to the power 2.2" : "") + "); begin the render with ER_TEX_COMPACT=0 to evaluate ER_FN_TEXTURE on it");
        }
    HIP_TRY(hipSetDevice(s->device));
    ScopedDevBuf<float> d_in, d_out;
    int rc;
    if ((rc = upload(d_in, in, (size_t)n * in_stride, s->stream)) != ER_OK) return rc;
    if ((rc = upload(d_out, (const float*)nullptr, (size_t)n * out_stride, s->stream)) != ER_OK) return rc;
    HIP_TRY(hipMemsetAsync(d_out.p, 0, std::max<size_t>((size_t)n * out_stride, 1) * sizeof(float), s->stream));
    er_launch_debug_eval(s->dev, kind, d_in.p, n, in_stride, d_out.p, out_stride, s->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_out.p, (size_t)n * out_stride * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return ER_OK;
}

static int er_measure_hbm_peak_impl(int device, uint64_t bytes, uint32_t iters, float* copy_GBps, float* read_GBps) {
    if (!copy_GBps || !read_GBps) return fail(ER_ERR_INVALID_ARG, "er_measure_hbm_peak: NULL argument");
    int ndev = er_device_count();
    if (device < 0 || device >= ndev) return fail(ER_ERR_NO_DEVICE, "er_measure_hbm_peak: no such HIP device");
    if (bytes == 0) bytes = 2ull << 30;
    if (iters == 0) iters = 5;
    HIP_TRY(hipSetDevice(device));
    const size_t n = (size_t)(bytes / 16);
    ScopedDevBuf<float4> a, b;
    ScopedDevBuf<float> sink;
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{st};
    int rc;
    if ((rc = upload(a, (const float4*)nullptr, n, st)) != ER_OK) return rc;
    if ((rc = upload(b, (const float4*)nullptr, n, st)) != ER_OK) return rc;
    if ((rc = upload(sink, (const float*)nullptr, 1, st)) != ER_OK) return rc;
    HIP_TRY(hipMemsetAsync(a.p, 0, n * 16, st));
    HIP_TRY(hipMemsetAsync(b.p, 0, n * 16, st));
    EventPair ev;
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
    float best_copy = 0, best_read = 0;
    for (uint32_t it = 0; it < iters + 1; it++) {      // the first round warms up
        float ms = 0;
        HIP_TRY(hipEventRecord(ev.a, st));
        er_launch_hbm_copy(a.p, b.p, n, st);
        HIP_TRY(hipEventRecord(ev.b, st));
        HIP_TRY(hipEventSynchronize(ev.b));
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        if (it > 0 && ms > 0) best_copy = std::max(best_copy, (float)(2.0 * n * 16 / (ms * 1e-3) / 1e9));
        HIP_TRY(hipEventRecord(ev.a, st));
        er_launch_hbm_read(a.p, sink.p, n, st);
        HIP_TRY(hipEventRecord(ev.b, st));
        HIP_TRY(hipEventSynchronize(ev.b));
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        if (it > 0 && ms > 0) best_read = std::max(best_read, (float)(1.0 * n * 16 / (ms * 1e-3) / 1e9));
    }
    HIP_TRY(hipGetLastError());
    *copy_GBps = best_copy;
    *read_GBps = best_read;
    return ER_OK;
}

extern "C" void er_debug_set_host_alloc_limit(uint64_t bytes) { g_host_alloc_limit.store(bytes); }


static int er_debug_stream_info_impl(ErScene* s, ErStreamInfo* out) {
    if (!s || !out) return fail(ER_ERR_INVALID_ARG, "er_debug_stream_info: NULL argument");
    std::lock_guard<std::mutex> lk(s->mtx);
    memset(out, 0, sizeof(*out));
    out->cost_spread = -1.0;
    if (!s->begun || !(s->params.flags & ER_FLAG_STREAM)) return ER_OK;
    out->waves = s->stream_waves; out->tracers = s->stream_tracers;
    out->large_regions = (s->stream_deal_alt_n != 0u && s->stream_deal_off == s->stream_deal_alt_off && s->stream_deal_n == s->stream_deal_alt_n) ? 1u : 0u;
    out->deal_pending = s->stream_deal_pending ? 1u : 0u;
    out->launches = (uint32_t)s->stream_launches;
    out->pixels_per_cu = (uint32_t)((size_t)s->dev.owned_tile_count * 64 / std::max<uint32_t>(1u, s->stream_blocks));
    out->lanes_busy = s->stream_busy; out->launch_ms = s->stream_launch_ms; out->cost_spread = s->stream_cost_spread;
    out->spec_started = s->stream_spec[0]; out->spec_right = s->stream_spec[1]; out->spec_wrong = s->stream_spec[2];
    // (er_launch_stream's choice: the speculative form only for scenes it starts speculative samples in, the keep form at 16 waves)
    const bool spec_scene = s->dev.max_bounces <= 1000u && s->tri_count >= ER_STREAM_SPEC_MIN_TRIS;
    out->form = (s->stream_spec_form && spec_scene) ? 2u : ((s->stream_keep && s->stream_waves == 16u) ? 1u : 0u);
    return ER_OK;
}

extern "C" {
void er_debug_set_gpu_build_failure(int kind) { er_debug_gpu_build_failure.store(kind < 0 || kind > 2 ? 0 : kind); }
int er_debug_stream_info(ErScene* s, ErStreamInfo* out) { return guarded("er_debug_stream_info", [&]() -> int { return er_debug_stream_info_impl(s, out); }); }
int er_debug_closest_hit(ErScene* s, const float* origins, const float* dirs, uint32_t n, int32_t* tri_ids, float* positions, float* distances) { return guarded("er_debug_closest_hit", [&]() -> int { return er_debug_closest_hit_impl(s, origins, dirs, n, tri_ids, positions, distances); }); }
int er_debug_cdf_search(const float* cdf, int length, const float* values, int32_t* out, int count) { return guarded("er_debug_cdf_search", [&]() -> int { return er_debug_cdf_search_impl(cdf, length, values, out, count); }); }
int er_debug_stream_deal(const uint32_t* owned, uint32_t count, uint32_t tiles_x, uint32_t blocks, int xcd_aware, uint32_t edge, uint32_t* out, uint32_t out_cap, uint32_t* most) {
    return guarded("er_debug_stream_deal", [&]() -> int { return er_debug_stream_deal_impl(owned, count, tiles_x, blocks, xcd_aware, edge, out, out_cap, most); });
}
int er_debug_bvh_check(const float* vertices, const float* normals, uint32_t tri_count, int threads, ErBvhCheck* out) { return guarded("er_debug_bvh_check", [&]() -> int { return er_debug_bvh_check_impl(vertices, normals, tri_count, threads, out); }); }
int er_debug_trace_rays(ErScene* s, const float* origins, const float* dirs, uint32_t n, const int32_t* self_slots, const float* limits, int32_t* tri_ids,
                        int32_t* slots, float* positions, float* distances, int32_t* info) {
    return guarded("er_debug_trace_rays", [&]() -> int { return er_debug_trace_rays_impl(s, origins, dirs, n, self_slots, limits, tri_ids, slots, positions, distances, info); });
}
int er_debug_trace_pixel(ErScene* s, uint32_t idx, ErTraceRec* recs, int max_recs, int* count) {
    return guarded("er_debug_trace_pixel", [&]() -> int { return er_debug_trace_pixel_impl(s, idx, recs, max_recs, count); });
}
int er_debug_eval(ErScene* s, int kind, const float* in, uint32_t n, uint32_t in_stride, float* out, uint32_t out_stride) {
    return guarded("er_debug_eval", [&]() -> int { return er_debug_eval_impl(s, kind, in, n, in_stride, out, out_stride); });
}
int er_measure_hbm_peak(int device, uint64_t bytes, uint32_t iters, float* copy_GBps, float* read_GBps) {
    return guarded("er_measure_hbm_peak", [&]() -> int { return er_measure_hbm_peak_impl(device, bytes, iters, copy_GBps, read_GBps); });
}
}  // extern "C"
