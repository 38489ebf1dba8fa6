// er_api.cpp -- implementation of the C ABI declared in include/eleven_hip.h.
// Host side of the device boundary: the MI355X replacement for dev_Scene's constructor,
// copy_scene, renderSetup, kernel_render_enqueue and RenderingManager::get_pass
// (reference src/kernel.cpp:244-266,651-706; src/SYCLCopy.cpp:3-104; src/Managers.cpp:287-302).
// There is NO CPU fallback: without a usable HIP device every compute entry point fails.
#include <chrono>

#include "er_scene.h"
#include <thread>
#include <system_error>

using namespace erh;

extern "C" {

int er_abi_version(void) { return ER_ABI_VERSION; }
const char* er_last_error(void) { return g_err.c_str(); }

int er_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int er_device_info_impl(int index, ErDeviceInfo* out) {
    if (!out) return fail(ER_ERR_INVALID_ARG, "er_device_info: out is NULL");
    int n = er_device_count();
    if (index < 0 || index >= n) return fail(ER_ERR_NO_DEVICE, "er_device_info: no such HIP device");
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, index));
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", p.name);
    snprintf(out->platform, sizeof(out->platform), "AMD HIP");
    snprintf(out->arch, sizeof(out->arch), "%s", p.gcnArchName);
    out->memory_bytes = p.totalGlobalMem;
    out->compute_units = (uint32_t)p.multiProcessorCount;
    out->compatible = strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
    return ER_OK;
}

static int er_device_find_impl(const char* selector) {
    if (!selector) return fail(ER_ERR_INVALID_ARG, "er_device_find: selector is NULL");
    int n = er_device_count();
    for (int i = 0; i < n; i++) {
        ErDeviceInfo info;
        if (er_device_info(i, &info) != ER_OK) continue;
        std::string s = std::string(info.name) + "|" + info.platform;   // NameSelector, reference src/Managers.cpp:201
        if (s == selector) return i;
    }
    return fail(ER_ERR_NO_DEVICE, std::string("er_device_find: no device matches '") + selector + "'");
}

static int er_scene_create_impl(const ErSceneDesc* d, ErScene** out) {
    if (!d || !out) return fail(ER_ERR_INVALID_ARG, "er_scene_create: NULL argument");
    *out = nullptr;
    if (d->x_res == 0 || d->y_res == 0) return fail(ER_ERR_INVALID_ARG, "er_scene_create: zero resolution");
    if ((uint64_t)d->x_res * d->y_res > 0x7fffffffull) return fail(ER_ERR_INVALID_ARG, "er_scene_create: resolution too large");
    if (d->tri_count >= (1u << 28)) return fail(ER_ERR_INVALID_ARG, "er_scene_create: too many triangles");
    if (d->tri_count && (!d->vertices || !d->normals || !d->tangents || !d->uvs || !d->tangent_sign || !d->material_id))
        return fail(ER_ERR_INVALID_ARG, "er_scene_create: triangle arrays missing");
    if (d->material_count == 0 || !d->materials) return fail(ER_ERR_INVALID_ARG, "er_scene_create: at least one material is required");
    size_t n = d->tri_count;
    {   // what the host copy of the description will take (texel payloads + 31 floats per triangle + the HDRI CDF)
        uint64_t bytes = (uint64_t)n * (9 * 3 + 6 + 1 + 1) * 4 + (uint64_t)d->material_count * sizeof(ErMaterial);
        for (uint32_t i = 0; i < d->texture_count && d->textures; i++)
            bytes += (uint64_t)std::max(0, d->textures[i].width) * std::max(0, d->textures[i].height) * std::max(0, d->textures[i].channels) * 4;
        bytes += (uint64_t)std::max(0, d->hdri.texture.width) * std::max(0, d->hdri.texture.height) * (std::max(0, d->hdri.texture.channels) + 1) * 4;
        host_reserve(bytes);
    }
    if (d->texture_count && !d->textures) return fail(ER_ERR_INVALID_ARG, "er_scene_create: texture table missing");
    std::unique_ptr<ErScene> owner(new ErScene());   // released to the caller only on success
    ErScene* s = owner.get();
    s->tri_count = d->tri_count;
    if (n) {
        s->vertices.assign(d->vertices, d->vertices + n * 9);
        s->normals.assign(d->normals, d->normals + n * 9);
        s->tangents.assign(d->tangents, d->tangents + n * 9);
        s->uvs.assign(d->uvs, d->uvs + n * 6);
        s->tangent_sign.assign(d->tangent_sign, d->tangent_sign + n);
        s->material_id.assign(d->material_id, d->material_id + n);
    }
    for (size_t i = 0; i < n; i++)
        if (s->material_id[i] < 0 || (uint32_t)s->material_id[i] >= d->material_count) {
            return fail(ER_ERR_INVALID_ARG, "er_scene_create: material_id out of range");
        }
    s->materials.assign(d->materials, d->materials + d->material_count);
    for (const ErMaterial& m : s->materials) {
        const int32_t ids[7] = {m.albedo_tex, m.emission_tex, m.roughness_tex, m.metallic_tex, m.normal_tex, m.opacity_tex, m.transmission_tex};
        for (int32_t id : ids)
            if (id >= (int32_t)d->texture_count) {
                return fail(ER_ERR_INVALID_ARG, "er_scene_create: texture id out of range");
            }
    }
    s->textures.resize(d->texture_count);
    for (uint32_t i = 0; i < d->texture_count; i++) {
        int rc = copy_tex(d->textures[i], s->textures[i], "scene texture");
        if (rc != ER_OK) return rc;
    }
    int rc = copy_tex(d->hdri.texture, s->hdri_tex, "hdri");
    if (rc != ER_OK) return rc;
    if (d->hdri.cdf) {
        s->hdri_cdf.assign(d->hdri.cdf, d->hdri.cdf + (size_t)s->hdri_tex.width * s->hdri_tex.height + 1);
        s->hdri_radiance_sum = d->hdri.radiance_sum;
    } else {
        host_generate_cdf(s->hdri_tex, s->hdri_cdf, s->hdri_radiance_sum);
    }
    s->camera = d->camera;
    if (d->point_light_count && d->point_lights) s->point_lights.assign(d->point_lights, d->point_lights + d->point_light_count);
    s->x_res = d->x_res;
    s->y_res = d->y_res;
    *out = owner.release();
    return ER_OK;
}

void er_scene_destroy(ErScene* s) {
    if (!s) return;
    if (s->begun || s->stream) {
        (void)hipSetDevice(s->device);
        if (s->stream) (void)hipStreamSynchronize(s->stream);
    }
    s->release_device();
    delete s;
}

// which deal of tiles the streaming schedule uses for a share of `owned_tiles` tiles on `blocks` workgroups
static bool stream_xcd_aware(size_t owned_tiles, uint32_t blocks) {
    const char* xe = getenv("ER_STREAM_XCD_TILES");        // A/B knob: 0 = tiles dealt round-robin to the workgroups (round 2)
    // (a share with no more pixels than slots -- an eighth of a 1080p frame -- has nothing waiting in its pixel rings; there the
    // plain round-robin deal balances a little better: 1.392 vs 1.41 ms per pass, profiles/r03_ab_sim_world8_knobs.log)
    return xe ? atoi(xe) != 0 : owned_tiles * 64 > (size_t)blocks * ER_STREAM_SLOTS;
}

static int er_render_begin_impl(ErScene* s, const ErRenderParams* p) {
    if (!s || !p) return fail(ER_ERR_INVALID_ARG, "er_render_begin: NULL argument");
    std::lock_guard<std::mutex> lk(s->mtx);
    uint32_t world = p->world ? p->world : 1;
    if (p->rank >= world) return fail(ER_ERR_INVALID_ARG, "er_render_begin: rank >= world");
    int ndev = er_device_count();
    if (ndev <= 0) return fail(ER_ERR_NO_DEVICE, "er_render_begin: no HIP device available (there is no CPU fallback)");
    if (p->device < 0 || p->device >= ndev) return fail(ER_ERR_NO_DEVICE, "er_render_begin: device ordinal out of range");
    {
        // the library carries gfx950 code objects only; launching on anything else faults inside the runtime
        ErDeviceInfo info;
        int irc = er_device_info(p->device, &info);
        if (irc != ER_OK) return irc;
        if (!info.compatible) return fail(ER_ERR_NO_DEVICE, std::string("er_render_begin: device arch '") + info.arch + "' is not gfx950");
    }
    if (s->begun || s->stream) {
        (void)hipSetDevice(s->device);
        if (s->stream) (void)hipStreamSynchronize(s->stream);
        s->release_device();
    }
    s->device = p->device;
    s->params = *p;
    s->params.world = world;
    if (s->params.max_bounces == 0) s->params.max_bounces = 5;   // the literal of reference src/kernel.cpp:508
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&s->ev_start));
    HIP_TRY(hipEventCreate(&s->ev_stop));

    // ---- acceleration structure: host binned-SAH build (default) or device linear-BVH build ----
    // What the rest of this function needs from either builder:
    size_t n = s->tri_count, n8_pieces = 0;
    uint32_t bvh2_nodes = 0, wide_nodes = 0, wide_depth = 0, leaf_count = 0;
    float bvh_lo[3] = {0, 0, 0}, bvh_hi[3] = {0, 0, 0}, lift_bound = 0;
    double build_ms = 0;
    bool built = false;
    int rc;
    EventPair ev;                       // destroyed on every return path
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
    const hipEvent_t u0 = ev.a, u1 = ev.b;
    // Staging buffers of the asynchronous uploads below.  They live until the stream has been synchronised at the
    // end of this function (HIP happens to make pageable host-to-device copies host-synchronous; this code does
    // not rely on it).
    ErBvhBuild bvh;
    std::vector<ErTriIsect> isect;
    std::vector<ErTriAttr> attr;
    std::vector<float4> geom;
    // Probe one kernel of every translation unit before the first launch: a library that carries no code object
    // this device can run (a stale or mis-targeted build) makes hipLaunchKernelGGL dereference a null function
    // inside the runtime (recorded in round 1: SIGSEGV under er_launch_setup after "No compatible code objects
    // found for gfx950:sramecc+:xnack-").  hipFuncGetAttributes reports the same condition as an error code.
    {
        const char* which = nullptr;
        hipError_t pe = er_probe_kernels(&which);
        if (pe != hipSuccess)
            return fail(ER_ERR_HIP, std::string("er_render_begin: libeleven_hip.so has no usable gfx950 code object for ") + (which ? which : "?") +
                                        " on this device (" + hipGetErrorString(pe) + "); rebuild with `make -C elevenrender_amd/csrc`");
    }
    // Which builder: the device build (er_gpu_build.hip: the host builder's binned SAH, level by level on the GPU -- the same tree in a
    // fraction of the time) unless the scene is small enough for the host to be done before the device has started, or the caller says so.
    const bool force_dev = (p->flags & ER_FLAG_GPU_BUILD) || getenv("ER_GPU_BUILD");
    const bool force_host = !force_dev && ((p->flags & ER_FLAG_HOST_BUILD) || getenv("ER_HOST_BUILD"));
    uint32_t min_tris = ER_GPU_BUILD_MIN_TRIS;
    if (const char* e = getenv("ER_GPU_BUILD_MIN_TRIS")) min_tris = (uint32_t)std::max(0, atoi(e));
    if (!force_host && (force_dev || s->tri_count >= min_tris) && s->tri_count > ER_BVH_LEAF_MAX) {
        // whole structure on the device: tree, wide-node collapse, slot order, triangle records
        std::string why;
        ErGpuSceneArrays arrays{s->vertices.data(), s->normals.data(), s->tangents.data(), s->uvs.data(), s->tangent_sign.data(), s->material_id.data()};
        ErGpuBvhDevice g;
        int brc = er_gpu_build_device(arrays, s->tri_count, s->device, &g, why);
        if (brc < 0 && force_dev) return fail(brc == -2 ? ER_ERR_OOM : ER_ERR_HIP, "er_render_begin: device BVH build: " + why);
        // What a default device build that did not deliver means: a DECLINE (> 0: too few triangles, a tree deeper than the
        // traversal stacks) and OUT OF DEVICE MEMORY (-2: the build's transient buffers beside other scenes or ranks) are ordinary -- the
        // host builder takes over silently (ER_GPU_BUILD_VERBOSE says so).  Anything else (-1: a HIP error, one of the builder's own
        // guards) is a fault of the builder: the host build still takes over -- the caller gets its image -- but never silently:
        // one line on stderr, unconditionally, with the reason.
        if (brc < 0) (void)hipGetLastError();
        if (brc == -1) fprintf(stderr, "[eleven_hip] er_render_begin: the device BVH build FAILED (%s); the host builder takes over (seconds instead of milliseconds at this size) -- please report\n", why.c_str());
        else if (brc != 0 && getenv("ER_GPU_BUILD_VERBOSE")) fprintf(stderr, "[er_gpu_build] %s: the host builds instead\n", why.c_str());
        if (brc == 0) {
            // the scene owns the three buffers from here on (release_device frees them on any later error)
            s->d_nodes.release(); s->d_nodes8.release(); s->d_attr.release();
            s->d_nodes.p = g.nodes; s->d_nodes.n = g.nodes_f4;
            s->d_nodes8.p = g.geom; s->d_nodes8.n = g.geom_f4;
            s->d_attr.p = g.attr; s->d_attr.n = g.attr_f4;
            if (g.geom_f4 >= (1ull << 30)) return fail(ER_ERR_INVALID_ARG, "er_render_begin: geometry exceeds the 16 GB addressable by the wide traversal");
            n8_pieces = g.n8_pieces;
            bvh2_nodes = (uint32_t)(g.nodes_f4 / 4); wide_nodes = g.nodes8_count; wide_depth = g.max_depth8; leaf_count = g.leaf_count;
            for (int a = 0; a < 3; a++) { bvh_lo[a] = g.lo[a]; bvh_hi[a] = g.hi[a]; }
            lift_bound = g.lift_bound;
            build_ms = g.build_ms;
            built = true;
        }   // brc > 0: the device builder declined (tree too deep for the traversal stacks); brc < 0 without the flag: it failed -- host build
    }
    if (built) HIP_TRY(hipEventRecord(u0, s->stream));
    if (!built) {
        host_reserve((uint64_t)n * (sizeof(ErTriIsect) * 2 + sizeof(ErTriAttr) + 2 * sizeof(ErNode)));
        er_build_bvh(s->vertices.data(), s->normals.data(), s->tri_count, 0, &bvh);
        HIP_TRY(hipEventRecord(u0, s->stream));
        if (bvh.max_depth > ER_BVH_MAX_DEPTH) return fail(ER_ERR_STATE, "er_render_begin: BVH deeper than the traversal stack");
        if (bvh.max_depth8 > ER_STACK8) return fail(ER_ERR_STATE, "er_render_begin: wide BVH deeper than the traversal stack (ER_STACK8)");
        isect.resize(n + 1);      // +1: the wide traversal fetches triangles in pairs
        attr.resize(n);
        for (size_t slot = 0; slot < n; slot++) {
            uint32_t id = bvh.slot_to_tri[slot];
            const float* v = &s->vertices[(size_t)id * 9];
            ErTriIsect& r = isect[slot];
            memcpy(r.v0, v, 12); memcpy(r.v1, v + 3, 12); memcpy(r.v2, v + 6, 12);
            r.tri_id = (int32_t)id;
            r.lift = bvh.tri_lift[id];
            r.sign = s->tangent_sign[id];
            ErTriAttr& a = attr[slot];
            memcpy(a.n, &s->normals[(size_t)id * 9], 36);
            memcpy(a.t, &s->tangents[(size_t)id * 9], 36);
            memcpy(a.uv, &s->uvs[(size_t)id * 6], 24);
            a.material = s->material_id[id];
            a.pad[0] = a.pad[1] = a.pad[2] = 0;
        }
        if ((rc = upload(s->d_nodes, bvh.nodes.data(), bvh.nodes.size() * 4, s->stream)) != ER_OK) return rc;
        memset(&isect[n], 0, sizeof(ErTriIsect));
        // wide nodes and triangle records share ONE buffer: the wide traversal addresses both with a 32-bit
        // offset in 16-byte units
        n8_pieces = bvh.nodes8.size() * ER_NODE8_PIECES + 8;   // + padding: a step fetches 96 B from a node's start
        geom.assign(n8_pieces + (n + 1) * 3, make_float4(0, 0, 0, 0));
        for (size_t k = 0; k < bvh.nodes8.size(); k++) memcpy(geom.data() + k * ER_NODE8_PIECES, &bvh.nodes8[k], sizeof(ErNode8));
        memcpy(geom.data() + n8_pieces, isect.data(), (n + 1) * sizeof(ErTriIsect));
        if (geom.size() >= (1ull << 30)) return fail(ER_ERR_INVALID_ARG, "er_render_begin: geometry exceeds the 16 GB addressable by the wide traversal");
        if ((rc = upload(s->d_nodes8, geom.data(), geom.size(), s->stream)) != ER_OK) return rc;
        if ((rc = upload(s->d_attr, attr.data(), n * ER_ATTR_PIECES, s->stream)) != ER_OK) return rc;
        bvh2_nodes = (uint32_t)bvh.nodes.size(); wide_nodes = (uint32_t)bvh.nodes8.size(); wide_depth = bvh.max_depth8; leaf_count = bvh.leaf_count;
        for (int a = 0; a < 3; a++) { bvh_lo[a] = bvh.lo[a]; bvh_hi[a] = bvh.hi[a]; }
        lift_bound = bvh.lift_bound;
        build_ms = bvh.build_ms;
    }
    if ((rc = upload(s->d_materials, s->materials.data(), s->materials.size(), s->stream)) != ER_OK) return rc;
    std::vector<float4> mat_pre(s->materials.size());
    for (size_t i = 0; i < s->materials.size(); i++) {      // DevScene::mat_pre: the per-material constants of generate_hit_data and GTR1
        const ErMaterial& m = s->materials[i];
        const float a = 0.1f + m.clearcoat_gloss * (0.001f - 0.1f);      // lerpf(0.1f, 0.001f, clearcoatGloss), src/Math.hpp:38-41
        const float a2 = a * a;
        mat_pre[i] = make_float4(ermath::er_pow(m.roughness, 2.2f), ermath::er_pow(m.metallic, 2.2f), a < 1.0f ? ermath::er_log(a2) : 0.0f, a < 1.0f ? 1.0f : 0.0f);
    }
    if (getenv("ER_MAT_PRE_ON_DEVICE")) for (auto& v : mat_pre) v.w = 0.0f;      // (A/B and test knob: GTR1's logarithm on the device)
    if ((rc = upload(s->d_mat_pre, mat_pre.data(), mat_pre.size(), s->stream)) != ER_OK) return rc;
    if ((rc = upload(s->d_lights, s->point_lights.data(), s->point_lights.size(), s->stream)) != ER_OK) return rc;
    // point-light queries double the shadow records and the shadow queues of the wavefront schedule (er_wavefront.h)
    const bool lights_on = (p->flags & ER_FLAG_POINT_LIGHTS) != 0 && !s->point_lights.empty();

    // textures: one float pool + a table
    // A texture that materials use ONLY for scalar channels -- opacity, roughness, metallic, transmission take `.x` of the fetched value
    // (src/kernel.cpp:100-150) -- is kept on the device with its first channel alone: a one-channel fetch returns that value in .x
    // (src/Texture.cpp:181-184), the filter's arithmetic on .x is the same, and the pool of C5 (64 x 3 noise textures of 3 channels, two
    // of the three used for roughness and metallic) shrinks from 151 MB to 84 MB of the caches it shares with the tree.
    // And where such a texture is read UNFILTERED and only as roughness or metallic, what it holds is the value to the power 2.2 that
    // generateHitData takes of every fetch (src/kernel.cpp:152-153), computed here with the device's own er_pow (er_math.h: one
    // implementation, the same bits -- as for the constants of DevScene::mat_pre); DevTex::filter = 2 marks it (fetched like filter 0).
    std::vector<uint8_t> vec_use(s->textures.size(), 0), scal_use(s->textures.size(), 0), plain_use(s->textures.size(), 0);
    auto mark = [&](std::vector<uint8_t>& v, int32_t id) { if (id >= 0 && (size_t)id < v.size()) v[(size_t)id] = 1; };
    for (const ErMaterial& m : s->materials) {
        mark(vec_use, m.albedo_tex); mark(vec_use, m.emission_tex); mark(vec_use, m.normal_tex);
        mark(scal_use, m.opacity_tex); mark(scal_use, m.roughness_tex); mark(scal_use, m.metallic_tex); mark(scal_use, m.transmission_tex);
        mark(plain_use, m.opacity_tex); mark(plain_use, m.transmission_tex);      // (scalar channels that are NOT raised to a power)
    }
    const char* compact_knob = getenv("ER_TEX_COMPACT");      // (A/B and test knob: 0 = every texture as it came)
    const bool compact = !(compact_knob && atoi(compact_knob) == 0);
    std::vector<DevTex> table(s->textures.size());
    std::vector<float> pool;
    const bool pow_on_host = !getenv("ER_MAT_PRE_ON_DEVICE");
    // mode per texture: 0 as it came, 1 first channel alone, 2 first channel alone to the power 2.2; the one-channel copies are made by a
    // few threads (C5: 8.4 M er_pow, ~0.2 s on one core) and appended in order
    std::vector<uint8_t> mode(s->textures.size(), 0);
    for (size_t i = 0; i < s->textures.size(); i++) {
        const HostTex& t = s->textures[i];
        if (compact && t.channels >= 1 && scal_use[i] && !vec_use[i]) {
            const bool powered = t.filter != 1 && !plain_use[i] && pow_on_host;
            if (t.channels > 1 || powered) mode[i] = powered ? 2 : 1;
        }
    }
    s->tex_mode = mode;
    std::vector<std::vector<float>> one(s->textures.size());
    // (sized HERE, on the calling thread: a std::bad_alloc then unwinds into guarded() -> ER_ERR_OOM; thrown inside a worker it
    // would be an uncaught exception of that thread, i.e. std::terminate.  The workers below only compute.)
    for (size_t i = 0; i < s->textures.size(); i++)
        if (mode[i]) one[i].resize((size_t)s->textures[i].width * (size_t)s->textures[i].height);
    {
        std::atomic<size_t> next{0};
        auto work = [&]() noexcept {
            for (size_t i = next.fetch_add(1); i < s->textures.size(); i = next.fetch_add(1)) {
                if (!mode[i]) continue;
                const HostTex& t = s->textures[i];
                const size_t n = one[i].size();
                for (size_t k = 0; k < n; k++) {
                    const float v = t.data[k * (size_t)t.channels];
                    one[i][k] = mode[i] == 2 ? ermath::er_pow(v, 2.2f) : v;
                }
            }
        };
        const unsigned nt = std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
        // (joined by a scope guard: if a std::thread constructor throws while earlier workers run, they are joined before the
        // exception leaves this block -- a joinable thread destroyed is std::terminate -- and the work they did not get to is done here)
        struct JoinAll {
            std::vector<std::thread> th;
            ~JoinAll() { for (auto& x : th) if (x.joinable()) x.join(); }
        } pool_threads;
        pool_threads.th.reserve(nt);
        try {
            for (unsigned k = 1; k < nt; k++) pool_threads.th.emplace_back(work);
        } catch (const std::system_error&) {
            // fewer workers than asked for: the calling thread takes what is left
        }
        work();
    }
    for (size_t i = 0; i < s->textures.size(); i++) {
        const HostTex& t = s->textures[i];
        if (mode[i]) {
            table[i] = DevTex{t.width, t.height, 1, mode[i] == 2 ? 2 : (t.filter == 1 ? 1 : 0), (uint32_t)pool.size()};
            pool.insert(pool.end(), one[i].begin(), one[i].end());
            std::vector<float>().swap(one[i]);
            continue;
        }
        table[i] = DevTex{t.width, t.height, t.channels, t.filter == 1 ? 1 : 0, (uint32_t)pool.size()};      // (anything but BILINEAR fetches unfiltered, src/Texture.cpp:229-236; 2 is the library's own mark)
        pool.insert(pool.end(), t.data.begin(), t.data.end());
    }
    // A material whose albedo, roughness and metallic textures have one size and one filter gets them texel by texel in one record of
    // five floats (DevFused, er_device.h): one fetch, one cache line and one coordinate computation per hit instead of three.  The
    // texel values are what Texture::getValueFromCoordinates returns for each (src/Texture.cpp:172-200); unfiltered, roughness and
    // metallic are stored to the power 2.2 as above.  The textures themselves stay where they are for every other use.
    std::vector<DevFused> fused(std::max<size_t>(1, s->materials.size()), DevFused{0, 0, 0, 0});
    const char* fuse_knob = getenv("ER_TEX_FUSE");      // (A/B knob; ER_TEX_COMPACT=0 = "every texture as it came" switches this off as well)
    if (compact && !(fuse_knob && atoi(fuse_knob) == 0)) {
        for (size_t m = 0; m < s->materials.size(); m++) {
            const ErMaterial& M = s->materials[m];
            const int32_t ids[3] = {M.albedo_tex, M.roughness_tex, M.metallic_tex};
            bool ok = true;
            for (int32_t id : ids) ok = ok && id >= 0 && (size_t)id < s->textures.size();
            if (!ok) continue;
            const HostTex &A = s->textures[(size_t)ids[0]], &R = s->textures[(size_t)ids[1]], &K = s->textures[(size_t)ids[2]];
            if (A.width != R.width || A.width != K.width || A.height != R.height || A.height != K.height) continue;
            if ((A.filter == 1) != (R.filter == 1) || (A.filter == 1) != (K.filter == 1)) continue;
            if (A.channels < 1 || R.channels < 1 || K.channels < 1) continue;
            const bool bilinear = A.filter == 1, powered = !bilinear && pow_on_host;
            const size_t n = (size_t)A.width * (size_t)A.height;
            if (pool.size() + 5 * n >= (1ull << 32)) continue;
            fused[m] = DevFused{A.width, A.height, bilinear ? 1 : (powered ? 2 : 0), (uint32_t)pool.size()};
            // (a texture the pass above already raised to the power is copied from the pool, not raised again: C5's 8.4 M texels)
            const DevTex tr = table[(size_t)ids[1]], tk = table[(size_t)ids[2]];
            const bool r_done = powered && tr.filter == 2, k_done = powered && tk.filter == 2;
            for (size_t k = 0; k < n; k++) {
                const float* a = A.data.data() + k * (size_t)A.channels;
                const float r = r_done ? pool[tr.offset + k] : R.data[k * (size_t)R.channels], mt = k_done ? pool[tk.offset + k] : K.data[k * (size_t)K.channels];
                // (one channel: the value three times; two: x, y, 0; three or more: the first three -- src/Texture.cpp:181-197)
                pool.push_back(a[0]);
                pool.push_back(A.channels == 1 ? a[0] : a[1]);
                pool.push_back(A.channels == 1 ? a[0] : (A.channels == 2 ? 0.0f : a[2]));
                pool.push_back(powered && !r_done ? ermath::er_pow(r, 2.2f) : r);
                pool.push_back(powered && !k_done ? ermath::er_pow(mt, 2.2f) : mt);
            }
        }
    }
    s->fused_any = false;
    for (const DevFused& f : fused) s->fused_any = s->fused_any || f.width > 0;
    if ((rc = upload(s->d_mat_fused, fused.data(), fused.size(), s->stream)) != ER_OK) return rc;
    DevTex hd{s->hdri_tex.width, s->hdri_tex.height, s->hdri_tex.channels, s->hdri_tex.filter, (uint32_t)pool.size()};
    pool.insert(pool.end(), s->hdri_tex.data.begin(), s->hdri_tex.data.end());
    if (pool.size() >= (1ull << 32)) return fail(ER_ERR_INVALID_ARG, "er_render_begin: texture pool exceeds 2^32 floats");
    if ((rc = upload(s->d_textures, table.data(), table.size(), s->stream)) != ER_OK) return rc;
    if ((rc = upload(s->d_tex_pool, pool.data(), pool.size(), s->stream)) != ER_OK) return rc;
    if ((rc = upload(s->d_cdf, s->hdri_cdf.data(), s->hdri_cdf.size(), s->stream)) != ER_OK) return rc;
    std::vector<uint32_t> guide;
    int buckets = er_build_cdf_guide(s->hdri_cdf.data(), s->hdri_tex.width * s->hdri_tex.height, guide);
    if ((rc = upload(s->d_guide, guide.data(), guide.size(), s->stream)) != ER_OK) return rc;

    size_t npx = (size_t)s->x_res * s->y_res;
    if ((rc = upload(s->d_passes, nullptr, npx * ER_PASS_COUNT, s->stream)) != ER_OK) return rc;
    if ((rc = upload(s->d_samples, nullptr, npx, s->stream)) != ER_OK) return rc;
    if ((rc = upload(s->d_rng, nullptr, npx, s->stream)) != ER_OK) return rc;
    std::vector<uint32_t> owned = s->tiles_of(s->params.rank, world);
    if ((rc = upload(s->d_owned, owned.data(), owned.size(), s->stream)) != ER_OK) return rc;
    if ((rc = upload(s->d_counters, nullptr, 1, s->stream)) != ER_OK) return rc;
    HIP_TRY(hipMemsetAsync(s->d_counters.p, 0, sizeof(DevCounters), s->stream));
    // schedule: forced by a flag, else the streaming schedule unless this rank's share is beyond its pixel rings (see eleven_hip.h)
    {
        uint32_t forced = p->flags & (ER_FLAG_MEGAKERNEL | ER_FLAG_FUSED | ER_FLAG_WAVEFRONT | ER_FLAG_STREAM);
        // (ER_FLAG_FUSED named round 1's lane-asynchronous single kernel, which the streaming schedule has overtaken at every frame size --
        // C1 at 256 x 256: 1 186 against 849 Msamples/s, profiles/r05_schedules_small_frames.log -- and which round 5 removed: the flag
        // is still accepted and means the streaming schedule)
        if (forced == ER_FLAG_FUSED) forced = ER_FLAG_STREAM;
        uint32_t sched = forced;
        if (forced == 0) {
            int cus = 1;
            HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device));
            cus = std::max(1, cus);
            // beyond the streaming schedule's pixel rings -> wavefront.  Decided on the REAL deal of tiles to workgroups, not on an
            // estimate of its largest share: the XCD-aware deal hands out whole super-tiles and can be less balanced than
            // ceil(tiles / CUs) (near the limit the estimate chose ER_FLAG_STREAM and er_render_begin then failed
            // with INVALID_ARG instead of taking the other schedule)
            std::vector<uint32_t> deal;
            const uint32_t most = er_stream_deal_tiles(owned.data(), (uint32_t)owned.size(), (s->x_res + ER_TILE - 1) / ER_TILE, (uint32_t)cus,
                                                       stream_xcd_aware(owned.size(), (uint32_t)cus), deal);
            // (and the streaming schedule carries a pixel as px | py << 16)
            sched = ((size_t)most * 64u > ER_STREAM_MAX_RING || s->x_res > 65535u || s->y_res > 65535u) ? ER_FLAG_WAVEFRONT : ER_FLAG_STREAM;
        }
        else if (forced & (forced - 1)) return fail(ER_ERR_INVALID_ARG, "er_render_begin: more than one schedule flag");
        s->params.flags = (s->params.flags & ~(uint32_t)(ER_FLAG_MEGAKERNEL | ER_FLAG_FUSED | ER_FLAG_WAVEFRONT | ER_FLAG_STREAM)) | sched;
    }
    if (s->params.flags & ER_FLAG_STREAM) {
        // streaming schedule: one workgroup per CU with ER_STREAM_SLOTS slots of the wavefront schedule's records each
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, s->device));
        s->stream_blocks = (uint32_t)prop.multiProcessorCount;
        // The split between tracer and shader waves (shader waves at issue priority 1; finished and escaped paths handled in batches
        // of their own, er_stream.hip).  Round 2: 10 + 6, round 3: 12 + 4 (11 + 5 with the point-light extension, whose shading
        // step is a third longer); since round 4's shorter shading step:
        // 13 tracer + 3 shader waves where the shading step is at its cheapest -- plain materials, a scene that lives in the caches
        // (C2: 1 787 vs 1 715 Msamples/s at 12 + 4) -- and 12 + 4 where it costs more: textured materials (C5 without lights: 1 560 vs
        // 1 489 at 13 + 3), point lights (C5: 1 350 vs 1 234), or a scene beyond the Infinity Cache (C4, 10 M triangles: 1 562 vs 1 483)
        // (profiles/r04_sweep_split_after_shader_diet.log).  That is the split to begin with; after every completed call it follows how
        // full the tracer lanes were (er_stream_adapt: a scene of another kind that starves 13 tracers gets 12 after its first call).
        // (textured materials started at 12 + 4 until their textures were fused / pre-powered / one-channel, er_render_begin above: C5 without
        // lights now 1 712 vs 1 640 at 12 + 4; a textured scene whose shading step is still too long for 13 tracers reads < 0.85 full lanes
        // after its first call and gets 12)
        s->stream_tracers = (lights_on || s->tri_count > 4000000u) ? 12 : 13;
        // A workgroup that owns hardly more pixels than it has slots (an eighth of a 1080p frame: 1 012 pixels per CU) cannot fill 12 tracer
        // waves -- a pixel's samples are one RNG stream, so pixels in flight are all the parallelism there is -- and runs faster as 9 tracer +
        // 3 shader waves of 168 registers (the shading step then spills 34 registers instead of 111 and three shader waves serve what four
        // did): 1.23 vs 1.35 ms per pass at 1/8 (1 012 pixels per CU); at 1/6 (1 350 pixels) 16 waves are ahead again, 1.43 vs 1.47 (profiles/r04_sweep_small_shares.log)
        s->stream_waves = 16;
        uint32_t small_tracers = 9;
        {
            const size_t px_per_cu = owned.size() * 64 / std::max<uint32_t>(1u, s->stream_blocks);
            // (round 6, profiles/r06_ab_long_pixels_and_split.log: with the shading step as short as it has become two shader waves serve ten tracers where
            // the slots are nearly all taken -- 1/8 ... 1/11 of the C2 frame 4.5 ... 1.5 % faster, C5's 1/8 share 1 ... 3 % -- and from 1/12 down 9 + 3 is ahead by 2 %)
            small_tracers = px_per_cu > ER_STREAM_TEN_TRACERS_SHARE ? 10u : 9u;
            if (px_per_cu <= ER_STREAM_SMALL_SHARE) { s->stream_waves = 12; s->stream_tracers = small_tracers; }
            // the lane occupancy says something about the balance of the two roles only where pixels are plentiful: a share of a few
            // pixels per slot cannot fill the lanes whatever the split (an eighth of a 1080p frame: 0.59 at the fastest split)
            // (nor in the instrumented kernel of ER_FLAG_COUNTERS, whose slower tracer loop shifts the balance)
            s->stream_adapt = px_per_cu >= 4u * ER_STREAM_SLOTS && !(p->flags & ER_FLAG_COUNTERS);
            s->stream_keep = px_per_cu <= ER_STREAM_KEEP_SHARE;      // (er_stream.hip s_front)
            s->stream_spec_form = px_per_cu <= ER_STREAM_SPEC_SHARE;      // few pixels per slot: slots fall free, speculative samples can use them (er_stream.hip)
        }
        if (const char* e = getenv("ER_STREAM_SPEC_FORM")) s->stream_spec_form = atoi(e) != 0;      // A/B knob
        if (const char* e = getenv("ER_STREAM_KEEP")) s->stream_keep = atoi(e) != 0;                // A/B knob
        if (const char* e = getenv("ER_STREAM_WAVES")) { s->stream_waves = atoi(e) == 12 ? 12 : 16; s->stream_tracers = s->stream_waves == 12 ? small_tracers : ((lights_on || s->tri_count > 4000000u) ? 12u : 13u); }   // A/B knob
        if (const char* e = getenv("ER_STREAM_TRACERS")) { s->stream_tracers = (uint32_t)std::min(13, std::max(1, atoi(e))); s->stream_adapt = false; }   // tuning knob: fixed split
        if (const char* e = getenv("ER_STREAM_ADAPT")) s->stream_adapt = atoi(e) != 0;
        s->stream_tracers_start = s->stream_tracers; s->stream_low_streak = 0; s->stream_up_budget = 1; s->stream_readings = 0;
        const size_t slots = (size_t)s->stream_blocks * ER_STREAM_SLOTS;
        if ((rc = upload(s->d_wf4, nullptr, slots * er_stream_record_bytes(lights_on) / sizeof(float4), s->stream)) != ER_OK) return rc;
        if ((rc = upload(s->d_wf1, nullptr, 32, s->stream)) != ER_OK) return rc;       // [1] status word, [2..5] the tracers' lane occupancy, [6..23] start / end per XCD, [24..26] speculative samples started / right / wrong
        if ((rc = upload(s->d_spill, nullptr, er_stream_spill_entries(s->stream_blocks), s->stream)) != ER_OK) return rc;
        s->stream_ctl = s->d_wf1.p;
        s->stream_lights = lights_on;
        HIP_TRY(hipMemsetAsync(s->stream_ctl, 0, 32 * sizeof(uint32_t), s->stream));
        s->stream_spec[0] = s->stream_spec[1] = s->stream_spec[2] = 0;
        // the workgroups' pixel rings: (pixel, samples left) entries, one per pixel of the workgroup's share
        // (capacity rounded up to a power of two: positions are monotonic 32-bit counters and may wrap)
        // (no minimum beyond one tile: a producer that comes round to a cell whose entry has not been read yet waits for its
        // reader, er_ring.h -- round 2 relied on "a lap of >= 4096 cells takes longer than a read")
        std::vector<uint32_t> deal;
        const bool xcd_aware = stream_xcd_aware(owned.size(), s->stream_blocks);
        const uint32_t tiles_x = (s->x_res + ER_TILE - 1) / ER_TILE;
        uint32_t most = er_stream_deal_tiles(owned.data(), (uint32_t)owned.size(), tiles_x, s->stream_blocks, xcd_aware, deal);
        s->stream_deal_off = 0; s->stream_deal_n = (uint32_t)deal.size();
        s->stream_deal_alt_off = 0; s->stream_deal_alt_n = 0;
        s->stream_xcd_spread = -1.0;
        // Larger screen regions per XCD are faster where a frame's cost is even and slower where it is not (er_stream.h), and only the run
        // can tell which.  With the knob unset a render STARTS on the default deal -- it spreads any frame's cost over the XCDs -- and keeps
        // the deal of ER_STREAM_SUPER_TILE_LARGE beside it in d_deal; during the first call the kernel adds every finished path's length to
        // its tile's sum (DevScene::tile_cost: counted work, not a measured time), and er_stream_adapt takes the large regions, for good,
        // if under THAT deal the XCDs' shares of the counted work are within ER_STREAM_COST_SPREAD_MAX of each other.  (Round 4 started
        // on the large deal and fell back on the XCDs' measured finish times: an uneven frame paid 8-18 % for its first call, and the
        // decision -- and the test of it -- hung on clocks.)
        const char* adapt_knob = getenv("ER_STREAM_ADAPT");
        s->stream_deal_pending = false;
        s->stream_deal_large.clear();
        s->d_tile_cost.release();
        if (xcd_aware && s->stream_blocks % 8u == 0u && owned.size() * 64 / s->stream_blocks >= ER_STREAM_SLOTS * 3u / 2u && !(p->flags & ER_FLAG_COUNTERS) &&      // (a half / a quarter of a 1080p frame: +1.2 % / +0.7 %)
            !getenv("ER_STREAM_SUPER_TILE") && !(adapt_knob && atoi(adapt_knob) == 0)) {
            std::vector<uint32_t> large;
            const uint32_t most_large = er_stream_deal_tiles(owned.data(), (uint32_t)owned.size(), tiles_x, s->stream_blocks, xcd_aware, large, ER_STREAM_SUPER_TILE_LARGE);
            if ((size_t)most_large * 64u <= ER_STREAM_MAX_RING) {      // (levelled, the two deals have the same largest share; never let the optional one fail the call)
                s->stream_deal_alt_off = (uint32_t)deal.size(); s->stream_deal_alt_n = (uint32_t)large.size();
                deal.insert(deal.end(), large.begin(), large.end());
                most = std::max(most, most_large);
                s->stream_deal_large.swap(large);
                const size_t n_tiles = (size_t)tiles_x * ((s->y_res + ER_TILE - 1) / ER_TILE);
                if ((rc = upload(s->d_tile_cost, nullptr, n_tiles, s->stream)) != ER_OK) return rc;
                HIP_TRY(hipMemsetAsync(s->d_tile_cost.p, 0, n_tiles * sizeof(uint32_t), s->stream));
                s->stream_deal_pending = true;
            }
        }
        if ((rc = upload(s->d_px_draws, nullptr, npx, s->stream)) != ER_OK) return rc;
        HIP_TRY(hipMemsetAsync(s->d_px_draws.p, 0, npx * sizeof(uint32_t), s->stream));
        if ((rc = upload(s->d_deal, deal.data(), deal.size(), s->stream)) != ER_OK) return rc;
        HIP_TRY(hipStreamSynchronize(s->stream));          // (`deal` goes out of scope)
        if (s->x_res > 65535u || s->y_res > 65535u)
            return fail(ER_ERR_INVALID_ARG, "er_render_begin: ER_FLAG_STREAM carries a pixel as x | y << 16: frames up to 65535 x 65535; use ER_FLAG_WAVEFRONT (the automatic choice does)");
        s->stream_ring_cap = 64u;
        while (s->stream_ring_cap < most * 64u) s->stream_ring_cap <<= 1;
        if (s->stream_ring_cap > ER_STREAM_MAX_RING)
            return fail(ER_ERR_INVALID_ARG, "er_render_begin: ER_FLAG_STREAM serves at most " + std::to_string((size_t)ER_STREAM_MAX_RING * s->stream_blocks) +
                                                " owned pixels per rank; use ER_FLAG_WAVEFRONT (the automatic choice does)");
        if ((rc = upload(s->d_ticket, nullptr, (size_t)s->stream_blocks * (size_t)s->stream_ring_cap * 2, s->stream)) != ER_OK) return rc;
        s->wf.clear();
    } else if (s->params.flags & ER_FLAG_WAVEFRONT) {
        // wavefront path state: one slot per owned pixel lane.
        // SLOT POOLS: the owned tiles are dealt round-robin to `pools` independent path pools, each with its own
        // queues and counters and its own HIP stream.  A ray is a sequential chain of dependent fetches, so every
        // trace launch ends in a tail in which the last long rays keep a few lanes busy while the chip idles
        // (measured: 29 % of all lane slots fall after the queue has run dry, at 19 % lane occupancy); the launches
        // of one pool fill the tails of the others.  Pools never exchange data (disjoint pixels), so the planes
        // do not depend on their number.  Measured on C2: 1 pool 901, 2 pools 959, 3 pools 969, 4 pools 947.
        uint32_t pools = 3;
        if (const char* e = getenv("ER_WF_POOLS")) pools = (uint32_t)std::min(8, std::max(1, atoi(e)));   // tuning knob
        pools = std::min<uint32_t>(pools, std::max<uint32_t>(1u, (uint32_t)owned.size()));
        const size_t pool_tiles = (owned.size() + pools - 1) / pools;
        const size_t qcap = pool_tiles * 64;          // queue capacity of one pool
        size_t slots = owned.size() * 64;
        const size_t sh = lights_on ? 2 : 1;          // shadow records per slot: [slot] HDRI query, [slot + slots] point light
        const size_t qs_cap = qcap * sh;              // shadow-queue capacity of one pool
        const size_t q_words = 2 * qcap + 2 * qs_cap; // two closest + two shadow queues per pool
        if ((rc = upload(s->d_wf4, nullptr, slots * (7 + 4 * sh), s->stream)) != ER_OK) return rc;
        if ((rc = upload(s->d_wf1, nullptr, slots * (3 + 3 * sh) + q_words * pools + (size_t)WF_COUNTS * pools, s->stream)) != ER_OK) return rc;
        WfState W{};
        float4* f = s->d_wf4.p;
        W.ray_o = f; W.ray_d = f + slots; W.light = f + 2 * slots; W.reduc = f + 3 * slots;
        W.aov_n = f + 4 * slots; W.aov_t = f + 5 * slots; W.aov_b = f + 6 * slots;
        W.sh_o = f + 7 * slots; W.sh_d = W.sh_o + sh * slots; W.c_vis = W.sh_d + sh * slots; W.c_occ = W.c_vis + sh * slots;
        uint32_t* u = s->d_wf1.p;
        W.hit = (int*)u; W.left = u + slots; W.hit2 = (int*)(u + 2 * slots);
        W.occluded = (int*)(u + 3 * slots); W.occ_a = W.occluded + sh * slots; W.occ_b = W.occ_a + sh * slots;
        uint32_t* qbase = u + (3 + 3 * sh) * slots;
        uint32_t* cbase = qbase + q_words * pools;
        W.pools = pools;
        W.slots = (uint32_t)slots;
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, s->device));
        uint32_t cus = (uint32_t)prop.multiProcessorCount;
        // persistent waves per CU, tuned together with the pools (the launches of different pools share the CUs):
        // trace 8/10/12/14/16 -> 1016/1046/1043/1022/1042 Msamples/s at shade 5; shade 4/5/6/7 -> 1019/1051/1012/987
        s->trace_blocks = cus * 12;
        s->shade_blocks = cus * 5;
        if (const char* e = getenv("ER_TRACE_WAVES_PER_CU")) s->trace_blocks = cus * (uint32_t)std::max(1, atoi(e));   // tuning knob
        if (const char* e = getenv("ER_SHADE_WAVES_PER_CU")) s->shade_blocks = cus * (uint32_t)std::max(1, atoi(e));
        // (uint2 entries: the trace waves' stack levels beyond the LDS ones, then the shade waves' exact re-trace stacks, two ints per entry)
        const size_t trace_spill = (size_t)s->trace_blocks * ER_STACK8 * 64;
        const size_t spill_per_pool = trace_spill + (size_t)s->shade_blocks * ER_BVH_MAX_DEPTH * 32;
        if ((rc = upload(s->d_spill, nullptr, spill_per_pool * pools, s->stream)) != ER_OK) return rc;
        s->wf.clear();
        for (uint32_t p2 = 0; p2 < pools; p2++) {
            W.pool = p2;
            uint32_t* q = qbase + (size_t)p2 * q_words;
            W.q[0] = q; W.q[1] = q + qcap; W.qs[0] = q + 2 * qcap; W.qs[1] = q + 2 * qcap + qs_cap;
            W.counts = cbase + (size_t)p2 * WF_COUNTS;
            W.spill = s->d_spill.p + (size_t)p2 * spill_per_pool;
            W.shade_stack = (int*)(W.spill + trace_spill);
            s->wf.push_back(W);
        }
        for (uint32_t p2 = 1; p2 < pools; p2++) {
            hipStream_t st;
            HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            s->pool_streams.push_back(st);
        }
        for (uint32_t p2 = 0; p2 < pools; p2++) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            s->pool_events.push_back(e);
        }
    }
    HIP_TRY(hipEventRecord(u1, s->stream));

    float scene_scale = 0;
    for (int a = 0; a < 3; a++) scene_scale = std::max(scene_scale, std::max(std::fabs(bvh_lo[a]), std::fabs(bvh_hi[a])));
    scene_scale = std::max(scene_scale, std::max(std::fabs(s->camera.position.x), std::max(std::fabs(s->camera.position.y), std::fabs(s->camera.position.z))));

    DevScene& D = s->dev;
    memset(&D, 0, sizeof(D));
    D.nodes = s->d_nodes.p;
    D.nodes8 = s->d_nodes8.p;
    D.tri_isect = s->d_nodes8.p + n8_pieces;
    D.tri_base_pieces = (uint32_t)n8_pieces;
    D.tri_attr = s->d_attr.p;
    D.tri_count = s->tri_count;
    D.node_count = bvh2_nodes;
    D.node8_count = wide_nodes;
    D.prune_margin = lift_bound + 1e-5f * scene_scale;
    D.max_lift = lift_bound;
    D.scene_scale = scene_scale;
    D.materials = s->d_materials.p;
    D.mat_pre = s->d_mat_pre.p;
    D.textures = s->d_textures.p;
    D.tex_pool = s->d_tex_pool.p;
    D.mat_fused = s->d_mat_fused.p;
    D.fused_any = s->fused_any ? 1u : 0u;
    D.hdri_tex = hd;
    D.hdri_cdf = s->d_cdf.p;
    D.hdri_guide = s->d_guide.p;
    D.hdri_buckets = buckets;
    D.hdri_radiance_sum = s->hdri_radiance_sum;
    D.cam = s->camera;
    D.x_res = s->x_res;
    D.y_res = s->y_res;
    D.tiles_x = (s->x_res + ER_TILE - 1) / ER_TILE;
    D.tiles_y = (s->y_res + ER_TILE - 1) / ER_TILE;
    D.max_bounces = s->params.max_bounces;
    {   // power-of-two sides everywhere: texture coordinates wrap with a mask instead of a signed division (er_device.h wrap_abs)
        auto pow2 = [](int32_t v) { return v > 0 && (v & (v - 1)) == 0; };
        bool all = pow2(hd.width) && pow2(hd.height);
        for (const HostTex& t : s->textures) all = all && pow2(t.width) && pow2(t.height);
        D.tex_pow2 = all ? 1u : 0u;
        if (const char* e = getenv("ER_TEX_POW2")) D.tex_pow2 = (atoi(e) != 0 && all) ? 1u : 0u;      // A/B knob: 0 = always the division
    }
    D.ext_flags = s->params.flags & (ER_FLAG_POINT_LIGHTS | ER_FLAG_MIS);
    D.lights = s->d_lights.p;
    D.light_count = (uint32_t)s->point_lights.size();
    D.passes = s->d_passes.p;
    D.samples = s->d_samples.p;
    D.rng = s->d_rng.p;
    D.owned_tiles = s->d_owned.p;
    D.owned_tile_count = (uint32_t)owned.size();
    D.counters = s->d_counters.p;
    D.tile_cost = ((s->params.flags & ER_FLAG_STREAM) && s->stream_deal_pending) ? s->d_tile_cost.p : nullptr;
    D.px_draws = (s->params.flags & ER_FLAG_STREAM) ? s->d_px_draws.p : nullptr;
    {   // the camera's rotation sines / cosines, once, with the functions the device would call (er_math.h: one implementation for both sides)
        const erd::CamTrig t = erd::camera_trig(D.cam);
        D.cam_cx = t.cx; D.cam_sx = t.sx; D.cam_cy = t.cy; D.cam_sy = t.sy; D.cam_cz = t.cz; D.cam_sz = t.sz;
        D.cam_trig_valid = getenv("ER_CAM_TRIG_ON_DEVICE") ? 0u : 1u;      // (A/B and test knob: 1 = use the host's values)
    }
    if ((rc = upload(s->d_dev, &D, 1, s->stream)) != ER_OK) return rc;

    er_launch_setup(D, s->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    float up_ms = 0;
    (void)hipEventElapsedTime(&up_ms, u0, u1);

    s->accel.node_count = wide_nodes;
    s->accel.node_bytes = sizeof(ErNode8);
    s->accel.max_depth = wide_depth;
    s->accel.leaf_count = leaf_count;
    s->accel.tri_record_bytes = sizeof(ErTriIsect);
    s->accel.build_ms = (float)build_ms;
    s->accel.upload_ms = up_ms;
    s->accel.lift_bound = lift_bound;
    s->accel.builder = built ? 1u : 0u;
    s->begun = true;
    return ER_OK;
}

static void er_stream_adapt(ErScene* s);

static int er_render_samples_async_impl(ErScene* s, uint32_t n) {
    if (!s) return fail(ER_ERR_INVALID_ARG, "er_render_samples: NULL scene");
    std::lock_guard<std::mutex> lk(s->mtx);
    if (!s->begun) return fail(ER_ERR_STATE, "er_render_samples: er_render_begin has not succeeded");
    if ((s->params.flags & ER_FLAG_STREAM) && n >= (1u << 24))      // (its pixel-ring entries keep the samples left in 24 bits)
        return fail(ER_ERR_INVALID_ARG, "er_render_samples: at most 16777215 samples per call in the streaming schedule");
    HIP_TRY(hipSetDevice(s->device));
    if (!s->timing_open) {
        HIP_TRY(hipEventRecord(s->ev_start, s->stream));
        s->timing_open = true;
    }
    if (n > 0) for (auto& u : s->unpacked) u.clear();     // other ranks' pixels gathered earlier are stale from here on
    const bool count = (s->params.flags & ER_FLAG_COUNTERS) != 0;
    const bool single = (s->params.flags & (ER_FLAG_MEGAKERNEL | ER_FLAG_STREAM)) != 0;
    if (single && (s->params.flags & ER_FLAG_PROFILE) && n > 0) {
        while (s->prof_events.size() < s->prof_used + 3) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            s->prof_events.push_back(e);
        }
        HIP_TRY(hipEventRecord(s->prof_events[s->prof_used++], s->stream));
    }
    if (s->params.flags & ER_FLAG_STREAM) {
        auto launch = [&](uint32_t k) -> int {
            HIP_TRY(hipMemsetAsync(s->stream_ctl + 2, 0, 30 * sizeof(uint32_t), s->stream));      // the call's lane-occupancy counts, its end per XCD, its speculation counts ...
            HIP_TRY(hipMemsetAsync(s->stream_ctl + 6, 0xFF, 2 * sizeof(uint32_t), s->stream));    // ... and its start (a minimum)
            if (k > 0) s->stream_launches++;
            er_launch_stream(s->dev, s->d_dev.p, s->d_wf4.p, s->stream_blocks * ER_STREAM_SLOTS, s->stream_lights, s->d_spill.p, s->d_deal.p + s->stream_deal_off, s->stream_deal_n, s->d_ticket.p, s->stream_ring_cap, s->stream_ctl + 1, k, count,
                             s->stream_blocks, s->stream_tracers, s->stream_waves, s->stream_spec_form, s->stream_keep, s->stream);
            return ER_OK;
        };
        // Round 6: a render that is ONE call must get the deal its frame deserves too.  While the deal is undecided, the first call's
        // first sample is a launch of its own: the kernel counts that pass's path lengths per tile (a count of work, the same on every
        // run), the library decides -- and stops the counting -- and the other n - 1 samples run on the deal decided.  Until round 6 the
        // decision came after the first CALL, so a host that issued one er_render_samples(256) never left the default deal (C2 -1.5 ... -4 %,
        // C4 -3 ... -5.7 %) and `bench.py --warmup 0` measured another kernel configuration than `--warmup 5`.  Cost: one more launch per
        // render (~0.8 ms) and a host wait of one sample pass inside this call, once; the image does not depend on the deal.
        static const bool split_first = [] { const char* e = getenv("ER_STREAM_SPLIT_FIRST"); return !(e && atoi(e) == 0); }();      // (A/B knob)
        if (s->stream_deal_pending && n >= 2u && split_first) {
            int rc = launch(1u);
            if (rc != ER_OK) return rc;
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(s->stream));
            if ((rc = er_scene_stream_status(s, "er_render_samples")) != ER_OK) return rc;
            s->stream_probe_launch = true;      // (a one-pass launch is no reading of the tracer lanes' occupancy: only the deal is decided on it)
            er_stream_adapt(s);
            s->stream_probe_launch = false;
            n -= 1u;
        }
        int rc = launch(n);
        if (rc != ER_OK) return rc;
    } else if (s->params.flags & ER_FLAG_MEGAKERNEL) {
        er_launch_render(s->dev, n, count, s->stream);
    }
    if (single && (s->params.flags & ER_FLAG_PROFILE) && n > 0) {
        HIP_TRY(hipEventRecord(s->prof_events[s->prof_used++], s->stream));
        HIP_TRY(hipEventRecord(s->prof_events[s->prof_used++], s->stream));   // (trace, shade) triple: shade = 0
    }
    if (!single && n > 0) {
        // a path takes at most max_bounces ray steps plus one finalize-only step
        const uint32_t iters = n * (s->params.max_bounces + 1);
        const uint32_t pools = (uint32_t)s->wf.size();
        const bool prof = (s->params.flags & ER_FLAG_PROFILE) != 0;
        if (prof) {
            const size_t need = s->prof_used / 3 + (size_t)iters * pools;      // launches profiled since the last er_wait
            while (s->prof_events.size() < 3 * need) {
                hipEvent_t e;
                HIP_TRY(hipEventCreate(&e));
                s->prof_events.push_back(e);
            }
            if (s->d_ray_log.n < need) {
                // grow, keeping what earlier calls of this timing window logged (they are still in flight or done)
                DevBuf<uint32_t> bigger;
                int rc2 = upload(bigger, nullptr, need * 2, s->stream);
                if (rc2 != ER_OK) return rc2;
                if (s->d_ray_log.p && s->prof_used) {
                    HIP_TRY(hipStreamSynchronize(s->stream));
                    HIP_TRY(hipMemcpy(bigger.p, s->d_ray_log.p, (s->prof_used / 3) * sizeof(uint32_t), hipMemcpyDeviceToDevice));
                }
                s->d_ray_log.release();
                s->d_ray_log = bigger;
            }
        }
        // fork: the pool streams start after everything already enqueued on the scene's stream
        if (pools > 1) {
            HIP_TRY(hipEventRecord(s->pool_events[0], s->stream));
            for (uint32_t p = 1; p < pools; p++) HIP_TRY(hipStreamWaitEvent(s->pool_streams[p - 1], s->pool_events[0], 0));
        }
        auto pool_stream = [&](uint32_t p) { return p == 0 ? s->stream : s->pool_streams[p - 1]; };
        for (uint32_t p = 0; p < pools; p++) er_launch_wf_begin(s->dev, s->wf[p], n, pool_stream(p));
        for (uint32_t it = 0; it < iters; it++) {
            for (uint32_t p = 0; p < pools; p++) {
                hipStream_t st = pool_stream(p);
                uint32_t* ray_log = prof ? s->d_ray_log.p + s->prof_used / 3 : nullptr;
                if (prof) HIP_TRY(hipEventRecord(s->prof_events[s->prof_used++], st));
                er_launch_wf_trace(s->dev, s->wf[p], it & 1, count, s->trace_blocks, ray_log, st);
                if (prof) HIP_TRY(hipEventRecord(s->prof_events[s->prof_used++], st));
                er_launch_wf_shade(s->dev, s->wf[p], it & 1, count, s->shade_blocks, st);
                if (prof) HIP_TRY(hipEventRecord(s->prof_events[s->prof_used++], st));
            }
        }
        // join: the scene's stream continues once every pool has drained
        for (uint32_t p = 1; p < pools; p++) {
            HIP_TRY(hipEventRecord(s->pool_events[p], s->pool_streams[p - 1]));
            HIP_TRY(hipStreamWaitEvent(s->stream, s->pool_events[p], 0));
        }
    }
    HIP_TRY(hipGetLastError());
    return ER_OK;
}

// The streaming kernel's waves give up instead of spinning forever if their workgroup makes no progress (er_stream.hip) and
// say so in a status word.  Called with the scene's stream idle (after a wait or a read-back): an unfinished call must not pass
// for a finished one.  The word stays set until the next er_render_begin.
int er_scene_stream_status(ErScene* s, const char* who) {
    if (!(s->params.flags & ER_FLAG_STREAM) || !s->stream_ctl) return ER_OK;
    uint32_t st[32] = {0};
    HIP_TRY(hipMemcpy(st, s->stream_ctl, sizeof(st), hipMemcpyDeviceToHost));
    if (s->stream_spec_seen != s->stream_launches) {      // (once per launch: a read-back after the same launch finds the same words)
        s->stream_spec_seen = s->stream_launches;
        for (int k = 0; k < 3; k++) s->stream_spec[k] += st[24 + k];
    }
    if (st[1] != 0)
        return fail(ER_ERR_STATE, std::string(who) + ": the streaming schedule stopped without finishing (watchdog status " + std::to_string(st[1]) + "); the planes are incomplete");
    const unsigned long long iters = (unsigned long long)st[2] | ((unsigned long long)st[3] << 32), busy = (unsigned long long)st[4] | ((unsigned long long)st[5] << 32);
    s->stream_busy = iters ? (double)busy / (64.0 * (double)iters) : 0.0;
    // how far apart the XCDs finished, as a share of the launch (100 MHz ticks; a launch under 2 ms says nothing: start-up and tail)
    auto u64 = [&](int i) { return (unsigned long long)st[i] | ((unsigned long long)st[i + 1] << 32); };
    const unsigned long long t0 = u64(6);
    unsigned long long lo = ~0ull, hi = 0;
    for (int x = 0; x < 8; x++) { const unsigned long long e = u64(8 + 2 * x); lo = std::min(lo, e); hi = std::max(hi, e); }
    s->stream_xcd_spread = (t0 != ~0ull && lo > t0 && hi - t0 >= 200000ull) ? (double)(hi - lo) / (double)(hi - t0) : -1.0;
    s->stream_launch_ms = (t0 != ~0ull && hi > t0) ? (double)(hi - t0) * 1e-5 : 0.0;
    return ER_OK;
}

// The two roles of the streaming kernel feed each other, and which one is short depends on the scene: how long a ray's traversal is
// against how long its shading step is.  What the tracers' lanes say after a call (counted by the kernel itself, two scalar
// operations per iteration): clearly not full = the shader waves cannot produce rays fast enough, and one tracer wave becomes a
// shader wave for the next call.  Measured with the product kernel (profiles/r04_sweep_split_after_shader_diet.log): C2 0.89-0.90 full
// at 13 + 3 (its best split); C4 0.81-0.85 at 13 + 3 and 0.92 at 12 + 4 (its best); C5 with lights 0.75 at 13 + 3, 0.90 at 12 + 4 (its
// best).  Lanes that ARE full say little (C4 looks alike at 12 + 4 and 11 + 5), so the split moves down: one wave after TWO
// consecutive calls whose lanes were under 0.85 full, down to 10 + 6 (7 + 5 of 12 waves).  A call shorter than ER_STREAM_ADAPT_MIN_MS
// of device time is not a reading at all (its lanes are mostly ramp-up and tail: a 1-spp preview would otherwise walk the split
// down for good), and ONE step back up is allowed per render when a later call reads above 0.93 (a demotion caused by two
// unrepresentative calls is undone; a second demotion after that stays).  The image does not depend on the split.
static void er_stream_adapt(ErScene* s) {
    const bool verbose = getenv("ER_STREAM_VERBOSE") != nullptr;      // (read per call: a test turns it on for one render)
    if (s->stream_adapted == s->stream_launches) return;              // (a second er_wait after the same launch: its measurements have been used)
    s->stream_adapted = s->stream_launches;
    if (verbose && (s->params.flags & ER_FLAG_STREAM) && !s->stream_adapt) fprintf(stderr, "[er_stream] tracer lanes %.3f full at %u + %u waves (fixed split)\n", s->stream_busy, s->stream_tracers, s->stream_waves - s->stream_tracers);
    // the deal: large screen regions per XCD if the XCDs' shares of the COUNTED work of the first call are alike under them (er_stream.h,
    // er_render_begin).  Decided once, from counts: the same decision on every run of the same frame.
    if ((s->params.flags & ER_FLAG_STREAM) && s->stream_deal_pending) {
        s->stream_deal_pending = false;
        std::vector<uint32_t> cost(s->d_tile_cost.n);
        if (hipMemcpy(cost.data(), s->d_tile_cost.p, cost.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess) {
            double x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const std::vector<uint32_t>& L = s->stream_deal_large;
            for (size_t i = 0; i < L.size(); i++)
                if (L[i] != 0xFFFFFFFFu && L[i] < cost.size()) x[(i % s->stream_blocks) % 8u] += (double)cost[L[i]];      // entry b + k * blocks belongs to workgroup b, XCD b % 8
            double lo = x[0], hi = x[0], sum = 0;
            for (double v : x) { lo = std::min(lo, v); hi = std::max(hi, v); sum += v; }
            const double spread = sum > 0 ? (hi - lo) / (sum / 8.0) : 0.0;
            const char* lim_env = getenv("ER_STREAM_COST_SPREAD_MAX");      // (test knob, read per decision: 0 keeps the default deal, a large value takes the large one)
            const double limit = lim_env ? atof(lim_env) : (double)ER_STREAM_COST_SPREAD_MAX;
            const bool take = sum > 0 && spread <= limit;
            s->stream_cost_spread = spread;
            if (verbose) fprintf(stderr, "[er_stream] counted work of the XCDs' shares on super-tiles of %u: %.4f of the mean apart (limit %.4f) -> %s\n", (unsigned)ER_STREAM_SUPER_TILE_LARGE, spread,
                                 limit, take ? "large regions" : "the default deal stays");
            if (take) { s->stream_deal_off = s->stream_deal_alt_off; s->stream_deal_n = s->stream_deal_alt_n; }
        }
        // the kernel stops counting: the scene descriptor it reads loses the pointer (ordered on the stream before the next launch)
        s->dev.tile_cost = nullptr;
        (void)hipMemcpyAsync(s->d_dev.p, &s->dev, sizeof(DevScene), hipMemcpyHostToDevice, s->stream);
        (void)hipStreamSynchronize(s->stream);
        s->stream_deal_large.clear(); s->stream_deal_large.shrink_to_fit();
    }
    if (verbose && (s->params.flags & ER_FLAG_STREAM) && s->stream_xcd_spread >= 0.0)
        fprintf(stderr, "[er_stream] XCDs finished %.3f of the launch apart (a measured time: printed, nothing is decided on it)\n", s->stream_xcd_spread);
    if (!s->stream_adapt || !(s->params.flags & ER_FLAG_STREAM) || s->stream_busy <= 0.0 || s->stream_probe_launch) return;
    const uint32_t lo = s->stream_waves == 12 ? 7u : 10u;
    const uint32_t before = s->stream_tracers;
    // (ER_STREAM_FORCE_BUSY: test knob -- the reading the mechanism is driven with instead of the measured one, whatever the launch's length;
    // the occupancy itself depends on clocks and may not be asserted on)
    // (a comma-separated list gives the k-th reading of the render its k-th value, the last one from then on)
    const char* forced = getenv("ER_STREAM_FORCE_BUSY");
    double busy = s->stream_busy;
    if (forced) {
        const char* q = forced;
        for (uint32_t k = 0; k < s->stream_readings; k++) { const char* c = strchr(q, ','); if (!c) break; q = c + 1; }
        busy = atof(q);
    }
    s->stream_readings++;
    if (!forced && s->stream_launch_ms < ER_STREAM_ADAPT_MIN_MS) {
        if (verbose) fprintf(stderr, "[er_stream] tracer lanes %.3f full in a launch of %.2f ms: too short to be a reading\n", s->stream_busy, s->stream_launch_ms);
        return;
    }
    if (busy < 0.85) {
        if (++s->stream_low_streak >= 2u && s->stream_tracers > lo) { s->stream_tracers--; s->stream_low_streak = 0; }
    } else {
        s->stream_low_streak = 0;
        if (busy > 0.93 && s->stream_up_budget > 0u && s->stream_tracers < s->stream_tracers_start) { s->stream_tracers++; s->stream_up_budget--; }
    }
    if (verbose) fprintf(stderr, "[er_stream] tracer lanes %.3f full%s at %u + %u waves -> %u + %u\n", busy, forced ? " (forced reading)" : "", before, s->stream_waves - before, s->stream_tracers, s->stream_waves - s->stream_tracers);
}

static int er_wait_impl(ErScene* s, float* elapsed_ms) {
    if (!s) return fail(ER_ERR_INVALID_ARG, "er_wait: NULL scene");
    std::lock_guard<std::mutex> lk(s->mtx);
    if (!s->begun) return fail(ER_ERR_STATE, "er_wait: er_render_begin has not succeeded");
    HIP_TRY(hipSetDevice(s->device));
    float ms = 0;
    if (s->timing_open) {
        HIP_TRY(hipEventRecord(s->ev_stop, s->stream));
        HIP_TRY(hipEventSynchronize(s->ev_stop));
        HIP_TRY(hipEventElapsedTime(&ms, s->ev_start, s->ev_stop));
        s->timing_open = false;
    } else {
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    if (elapsed_ms) *elapsed_ms = ms;
    { int rc = er_scene_stream_status(s, "er_wait"); if (rc != ER_OK) { s->prof_used = 0; return rc; } }   // (the profiling window ends with the call either way)
    er_stream_adapt(s);
    s->profile = ErProfile{};
    s->profile.schedule = s->params.flags & (ER_FLAG_MEGAKERNEL | ER_FLAG_WAVEFRONT | ER_FLAG_STREAM);
    s->profile.concurrency = (s->params.flags & ER_FLAG_WAVEFRONT) ? (uint32_t)std::max<size_t>(1, s->wf.size()) : 1u;
    // The wavefront host loop always enqueues n * (max_bounces + 1) iterations per pool; the last ones find empty
    // queues (a path rarely takes every bounce).  Those launches are reported apart, so that per-launch figures are
    // averages over launches that traced something.
    std::vector<uint32_t> ray_log;
    const bool wf = (s->params.flags & ER_FLAG_WAVEFRONT) != 0;
    if (wf && s->prof_used && s->d_ray_log.p) {
        ray_log.resize(s->prof_used / 3);
        HIP_TRY(hipMemcpy(ray_log.data(), s->d_ray_log.p, ray_log.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i + 2 < s->prof_used; i += 3) {
        float a = 0, b = 0;
        HIP_TRY(hipEventElapsedTime(&a, s->prof_events[i], s->prof_events[i + 1]));
        HIP_TRY(hipEventElapsedTime(&b, s->prof_events[i + 1], s->prof_events[i + 2]));
        if (wf && i / 3 < ray_log.size() && ray_log[i / 3] == 0) {
            s->profile.empty_launches++;
            s->profile.empty_ms += a + b;
            continue;
        }
        s->profile.trace_ms += a;
        s->profile.shade_ms += b;
        s->profile.trace_launches++;
        s->profile.shade_launches++;
        if (wf && i / 3 < ray_log.size()) s->profile.rays_logged += ray_log[i / 3];
    }
    s->prof_used = 0;
    return ER_OK;
}

static int er_render_samples_impl(ErScene* s, uint32_t n) {
    int rc = er_render_samples_async(s, n);
    if (rc != ER_OK) return rc;
    return er_wait(s, nullptr);
}

// the copy itself; the caller holds s->mtx
static int read_back_locked(ErScene* s, const void* src, void* dst, size_t bytes, const char* who) {
    if (!s->begun) return fail(ER_ERR_STATE, std::string(who) + ": er_render_begin has not succeeded");
    HIP_TRY(hipSetDevice(s->device));
    // ordered after everything enqueued so far: a sample-boundary snapshot, never a torn read
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return er_scene_stream_status(s, who);
}

static int read_back(ErScene* s, const void* src, void* dst, size_t bytes, const char* who) {
    if (!s || !dst) return fail(ER_ERR_INVALID_ARG, std::string(who) + ": NULL argument");
    std::lock_guard<std::mutex> lk(s->mtx);
    return read_back_locked(s, src, dst, bytes, who);
}

static int er_samples_done_impl(ErScene* s, uint32_t* out) {
    if (!s) return fail(ER_ERR_INVALID_ARG, "er_samples_done: NULL scene");
    // dev_samples[0], reference src/Managers.cpp:217-221; with tile sharding pixel 0 may belong to another
    // rank, so report the first owned pixel instead.
    uint32_t first = 0;
    std::unique_lock<std::mutex> lk(s->mtx);
    if (s->begun && s->params.world > 1) {
        std::vector<uint32_t> t = s->tiles_of(s->params.rank, s->params.world);
        if (!t.empty()) {
            uint32_t tiles_x = (s->x_res + ER_TILE - 1) / ER_TILE;
            first = (t[0] / tiles_x) * ER_TILE * s->x_res + (t[0] % tiles_x) * ER_TILE;
        }
    }
    lk.unlock();
    return read_back(s, s->d_samples.p + first, out, sizeof(uint32_t), "er_samples_done");
}

static int er_read_pass_impl(ErScene* s, int pass, float* dst) {
    if (pass < 0 || pass >= ER_PASS_COUNT) return fail(ER_ERR_INVALID_ARG, "er_read_pass: pass out of range");
    if (!s) return fail(ER_ERR_INVALID_ARG, "er_read_pass: NULL scene");
    if (!dst) return fail(ER_ERR_INVALID_ARG, "er_read_pass: NULL argument");
    // the device keeps the accumulated passes interleaved per pixel (er_pass_index, er_device.h): the plane the ABI hands out is
    // gathered into a staging plane first, on the library's stream (so it is the same sample-boundary snapshot as before).
    // ONE staging plane per scene: the lock is held from the gather kernel to the end of the copy, so that two threads reading
    // two passes of one scene cannot interleave as gather A, gather B, copy, copy.
    std::lock_guard<std::mutex> lk(s->mtx);
    if (!s->begun) return fail(ER_ERR_STATE, "er_read_pass: er_render_begin has not succeeded");
    size_t npx = (size_t)s->x_res * s->y_res;
    HIP_TRY(hipSetDevice(s->device));
    int rc;
    if (s->d_plane.n < npx && (rc = upload(s->d_plane, nullptr, npx, s->stream)) != ER_OK) return rc;
    er_launch_plane(s->dev, pass, s->d_plane.p, s->stream);
    HIP_TRY(hipGetLastError());
    return read_back_locked(s, s->d_plane.p, dst, npx * sizeof(float4), "er_read_pass");
}
// ---- checkpoint / resume: passes + samples + rng are the whole progressive state (reference src/kernel.h:44-46) ----
namespace {
struct StateHeader {
    char magic[8];            // "ERSTATE2" (2: the accumulated passes interleaved per pixel, er_pass_index; a snapshot of layout 1 is refused)
    uint32_t x_res, y_res, passes, reserved;
    uint64_t bytes;
    uint8_t pad[32];
};
static_assert(sizeof(StateHeader) == 64, "snapshot header is 64 bytes");
uint64_t state_bytes(const ErScene* s) {
    const uint64_t npx = (uint64_t)s->x_res * s->y_res;
    return sizeof(StateHeader) + npx * (ER_PASS_COUNT * sizeof(float4) + 2 * sizeof(uint32_t));
}
}  // namespace

static int er_state_size_impl(ErScene* s, uint64_t* bytes) {
    if (!s || !bytes) return fail(ER_ERR_INVALID_ARG, "er_state_size: NULL argument");
    *bytes = state_bytes(s);
    return ER_OK;
}

static int er_state_export_impl(ErScene* s, void* dst, uint64_t bytes) {
    if (!s || !dst) return fail(ER_ERR_INVALID_ARG, "er_state_export: NULL argument");
    std::lock_guard<std::mutex> lk(s->mtx);
    if (!s->begun) return fail(ER_ERR_STATE, "er_state_export: er_render_begin has not succeeded");
    if (bytes < state_bytes(s)) return fail(ER_ERR_INVALID_ARG, "er_state_export: buffer smaller than er_state_size");
    HIP_TRY(hipSetDevice(s->device));
    const size_t npx = (size_t)s->x_res * s->y_res;
    StateHeader h{};
    memcpy(h.magic, "ERSTATE2", 8);
    h.x_res = s->x_res; h.y_res = s->y_res; h.passes = ER_PASS_COUNT; h.bytes = state_bytes(s);
    uint8_t* p = (uint8_t*)dst;
    memcpy(p, &h, sizeof(h));
    p += sizeof(h);
    // ordered after everything enqueued so far (the pool streams join the scene's stream at the end of every call)
    HIP_TRY(hipMemcpyAsync(p, s->d_passes.p, npx * ER_PASS_COUNT * sizeof(float4), hipMemcpyDeviceToHost, s->stream));
    p += npx * ER_PASS_COUNT * sizeof(float4);
    HIP_TRY(hipMemcpyAsync(p, s->d_samples.p, npx * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    p += npx * sizeof(uint32_t);
    HIP_TRY(hipMemcpyAsync(p, s->d_rng.p, npx * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return er_scene_stream_status(s, "er_state_export");     // never checkpoint planes the library itself calls incomplete
}

static int er_state_import_impl(ErScene* s, const void* src, uint64_t bytes) {
    if (!s || !src) return fail(ER_ERR_INVALID_ARG, "er_state_import: NULL argument");
    std::lock_guard<std::mutex> lk(s->mtx);
    if (!s->begun) return fail(ER_ERR_STATE, "er_state_import: er_render_begin has not succeeded");
    StateHeader h;
    if (bytes < sizeof(h)) return fail(ER_ERR_INVALID_ARG, "er_state_import: truncated snapshot");
    memcpy(&h, src, sizeof(h));
    if (memcmp(h.magic, "ERSTATE2", 8) != 0) return fail(ER_ERR_INVALID_ARG, "er_state_import: not a snapshot of this library version (bad magic)");
    if (h.x_res != s->x_res || h.y_res != s->y_res || h.passes != ER_PASS_COUNT)
        return fail(ER_ERR_INVALID_ARG, "er_state_import: the snapshot is of a " + std::to_string(h.x_res) + "x" + std::to_string(h.y_res) + " frame");
    if (h.bytes != state_bytes(s) || bytes < h.bytes) return fail(ER_ERR_INVALID_ARG, "er_state_import: truncated snapshot");
    HIP_TRY(hipSetDevice(s->device));
    const size_t npx = (size_t)s->x_res * s->y_res;
    const uint8_t* p = (const uint8_t*)src + sizeof(h);
    HIP_TRY(hipMemcpyAsync(s->d_passes.p, p, npx * ER_PASS_COUNT * sizeof(float4), hipMemcpyHostToDevice, s->stream));
    p += npx * ER_PASS_COUNT * sizeof(float4);
    HIP_TRY(hipMemcpyAsync(s->d_samples.p, p, npx * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    p += npx * sizeof(uint32_t);
    HIP_TRY(hipMemcpyAsync(s->d_rng.p, p, npx * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));      // the caller's buffer may go away
    return ER_OK;
}

static int er_denoise_impl(ErScene* s, uint32_t levels, float colour_sigma) {
    if (!s) return fail(ER_ERR_INVALID_ARG, "er_denoise: NULL scene");
    std::lock_guard<std::mutex> lk(s->mtx);
    if (!s->begun) return fail(ER_ERR_STATE, "er_denoise: er_render_begin has not succeeded");
    // a sharded frame: only on the rank that BEAUTY and NORMAL were gathered to (er_gather_pass / er_unpack_owned of every
    // other rank) after the last sample; DenoiseManager of the reference likewise runs where the whole pass is (src/CommandManager.cpp:265-274)
    if (s->params.world > 1 && (s->unpacked[ER_PASS_BEAUTY].size() + 1 < s->params.world || s->unpacked[ER_PASS_NORMAL].size() + 1 < s->params.world))
        return fail(ER_ERR_STATE, "er_denoise: the frame is sharded over several ranks; gather BEAUTY and NORMAL to this rank first (er_gather_pass)");
    if (levels == 0) levels = 5;
    if (levels > 8) return fail(ER_ERR_INVALID_ARG, "er_denoise: at most 8 levels");
    if (!(colour_sigma >= 0)) return fail(ER_ERR_INVALID_ARG, "er_denoise: colour_sigma must be >= 0");
    if (colour_sigma == 0) colour_sigma = 1.0f;
    HIP_TRY(hipSetDevice(s->device));
    const size_t npx = (size_t)s->x_res * s->y_res;
    ScopedDevBuf<float4> tmp;
    int rc;
    if ((rc = upload(tmp, (const void*)nullptr, npx, s->stream)) != ER_OK) return rc;
    const float4* beauty = s->d_passes.p + er_pass_index(npx, ER_PASS_BEAUTY, 0);      // (interleaved passes: stride 4, er_device.h)
    const float4* normal = s->d_passes.p + er_pass_index(npx, ER_PASS_NORMAL, 0);
    float4* out = s->d_passes.p + er_pass_index(npx, ER_PASS_DENOISE, 0);
    // ping-pong so that the last level lands in the DENOISE plane
    const float4* src = beauty;
    for (uint32_t k = 0; k < levels; k++) {
        float4* dst = ((levels - 1 - k) & 1u) ? tmp.p : out;
        // the colour edge-stop tightens with the level, as the residual noise shrinks
        const float kc = 1.0f / (colour_sigma * colour_sigma) * (float)(1u << k);
        er_launch_atrous(src, src == beauty ? 4 : 1, normal, 4, dst, (int)s->x_res, (int)s->y_res, 1 << k, kc, s->stream);
        src = dst;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    return er_scene_stream_status(s, "er_denoise");
}

static int er_read_samples_impl(ErScene* s, uint32_t* dst) {
    if (!s) return fail(ER_ERR_INVALID_ARG, "er_read_samples: NULL scene");
    return read_back(s, s->d_samples.p, dst, (size_t)s->x_res * s->y_res * 4, "er_read_samples");
}
static int er_read_rng_impl(ErScene* s, uint32_t* dst) {
    if (!s) return fail(ER_ERR_INVALID_ARG, "er_read_rng: NULL scene");
    return read_back(s, s->d_rng.p, dst, (size_t)s->x_res * s->y_res * 4, "er_read_rng");
}

static int er_owned_count_impl(ErScene* s, uint32_t rank, uint64_t* out) {
    if (!s || !out) return fail(ER_ERR_INVALID_ARG, "er_owned_count: NULL argument");
    uint32_t world = s->begun ? s->params.world : 1;
    if (rank >= world) return fail(ER_ERR_INVALID_ARG, "er_owned_count: rank >= world");
    *out = (uint64_t)s->tiles_of(rank, world).size() * 64;
    return ER_OK;
}

static int er_pack_owned_impl(ErScene* s, int pass, void* dev_dst) {
    if (!s || !dev_dst) return fail(ER_ERR_INVALID_ARG, "er_pack_owned: NULL argument");
    if (pass < 0 || pass >= ER_PASS_COUNT) return fail(ER_ERR_INVALID_ARG, "er_pack_owned: pass out of range");
    std::lock_guard<std::mutex> lk(s->mtx);
    if (!s->begun) return fail(ER_ERR_STATE, "er_pack_owned: er_render_begin has not succeeded");
    HIP_TRY(hipSetDevice(s->device));
    er_launch_pack(s->dev, s->d_owned.p, s->dev.owned_tile_count, pass, dev_dst, s->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    return er_scene_stream_status(s, "er_pack_owned");
}

static int er_unpack_owned_impl(ErScene* s, int pass, uint32_t src_rank, const void* dev_src) {
    if (!s || !dev_src) return fail(ER_ERR_INVALID_ARG, "er_unpack_owned: NULL argument");
    if (pass < 0 || pass >= ER_PASS_COUNT) return fail(ER_ERR_INVALID_ARG, "er_unpack_owned: pass out of range");
    std::lock_guard<std::mutex> lk(s->mtx);
    if (!s->begun) return fail(ER_ERR_STATE, "er_unpack_owned: er_render_begin has not succeeded");
    if (src_rank >= s->params.world) return fail(ER_ERR_INVALID_ARG, "er_unpack_owned: src_rank >= world");
    HIP_TRY(hipSetDevice(s->device));
    auto it = s->d_rank_tiles.find(src_rank);
    if (it == s->d_rank_tiles.end()) {
        std::vector<uint32_t> t = s->tiles_of(src_rank, s->params.world);
        ScopedDevBuf<uint32_t> b;
        int rc = upload(b, t.data(), t.size(), s->stream);
        if (rc != ER_OK) return rc;
        HIP_TRY(hipStreamSynchronize(s->stream));   // t goes out of scope
        it = s->d_rank_tiles.emplace(src_rank, DevBuf<uint32_t>(b)).first;
        b.p = nullptr;                               // now owned by the scene
    }
    er_launch_unpack(s->dev, it->second.p, (uint32_t)it->second.n, pass, dev_src, s->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (src_rank != s->params.rank) s->unpacked[pass].insert(src_rank);
    return ER_OK;
}

static int er_get_counters_impl(ErScene* s, ErCounters* out) {
    if (!s || !out) return fail(ER_ERR_INVALID_ARG, "er_get_counters: NULL argument");
    DevCounters c;
    int rc = read_back(s, s->d_counters.p, &c, sizeof(c), "er_get_counters");
    if (rc != ER_OK) return rc;
    out->paths = c.paths; out->bounce_samples = c.bounce_samples; out->rays = c.rays; out->node_visits = c.node_visits;
    out->tri_tests = c.tri_tests; out->shaded_hits = c.shaded_hits; out->texel_fetches = c.texel_fetches;
    out->hdri_samples = c.hdri_samples;
    out->trace_wave_steps = c.trace_wave_steps; out->trace_busy_lanes = c.trace_busy_lanes;
    out->trace_node_lanes = c.trace_node_lanes; out->trace_tri_lanes = c.trace_tri_lanes;
    return ER_OK;
}

static int er_get_profile_impl(ErScene* s, ErProfile* out) {
    if (!s || !out) return fail(ER_ERR_INVALID_ARG, "er_get_profile: NULL argument");
    if (!s->begun) return fail(ER_ERR_STATE, "er_get_profile: er_render_begin has not succeeded");
    *out = s->profile;
    return ER_OK;
}

static int er_accel_info_impl(ErScene* s, ErAccelInfo* out) {
    if (!s || !out) return fail(ER_ERR_INVALID_ARG, "er_accel_info: NULL argument");
    if (!s->begun) return fail(ER_ERR_STATE, "er_accel_info: er_render_begin has not succeeded");
    *out = s->accel;
    return ER_OK;
}

}  // extern "C"

// ---- the exported entry points: every body above runs inside guarded() (no exception crosses the C ABI) ----
extern "C" {
int er_device_info(int index, ErDeviceInfo* out) { return guarded("er_device_info", [&]() -> int { return er_device_info_impl(index, out); }); }
int er_device_find(const char* selector) { return guarded("er_device_find", [&]() -> int { return er_device_find_impl(selector); }); }
int er_scene_create(const ErSceneDesc* d, ErScene** out) { return guarded("er_scene_create", [&]() -> int { return er_scene_create_impl(d, out); }); }
int er_render_begin(ErScene* s, const ErRenderParams* p) { return guarded("er_render_begin", [&]() -> int { return er_render_begin_impl(s, p); }); }
int er_render_samples_async(ErScene* s, uint32_t n) { return guarded("er_render_samples_async", [&]() -> int { return er_render_samples_async_impl(s, n); }); }
int er_wait(ErScene* s, float* elapsed_ms) { return guarded("er_wait", [&]() -> int { return er_wait_impl(s, elapsed_ms); }); }
int er_render_samples(ErScene* s, uint32_t n) { return guarded("er_render_samples", [&]() -> int { return er_render_samples_impl(s, n); }); }
int er_samples_done(ErScene* s, uint32_t* out) { return guarded("er_samples_done", [&]() -> int { return er_samples_done_impl(s, out); }); }
int er_read_pass(ErScene* s, int pass, float* dst) { return guarded("er_read_pass", [&]() -> int { return er_read_pass_impl(s, pass, dst); }); }
int er_state_size(ErScene* s, uint64_t* bytes) { return guarded("er_state_size", [&]() -> int { return er_state_size_impl(s, bytes); }); }
int er_state_export(ErScene* s, void* dst, uint64_t bytes) { return guarded("er_state_export", [&]() -> int { return er_state_export_impl(s, dst, bytes); }); }
int er_state_import(ErScene* s, const void* src, uint64_t bytes) { return guarded("er_state_import", [&]() -> int { return er_state_import_impl(s, src, bytes); }); }
int er_denoise(ErScene* s, uint32_t levels, float colour_sigma) { return guarded("er_denoise", [&]() -> int { return er_denoise_impl(s, levels, colour_sigma); }); }
int er_read_samples(ErScene* s, uint32_t* dst) { return guarded("er_read_samples", [&]() -> int { return er_read_samples_impl(s, dst); }); }
int er_read_rng(ErScene* s, uint32_t* dst) { return guarded("er_read_rng", [&]() -> int { return er_read_rng_impl(s, dst); }); }
int er_owned_count(ErScene* s, uint32_t rank, uint64_t* out) { return guarded("er_owned_count", [&]() -> int { return er_owned_count_impl(s, rank, out); }); }
int er_pack_owned(ErScene* s, int pass, void* dev_dst) { return guarded("er_pack_owned", [&]() -> int { return er_pack_owned_impl(s, pass, dev_dst); }); }
int er_unpack_owned(ErScene* s, int pass, uint32_t src_rank, const void* dev_src) { return guarded("er_unpack_owned", [&]() -> int { return er_unpack_owned_impl(s, pass, src_rank, dev_src); }); }
int er_get_counters(ErScene* s, ErCounters* out) { return guarded("er_get_counters", [&]() -> int { return er_get_counters_impl(s, out); }); }
int er_get_profile(ErScene* s, ErProfile* out) { return guarded("er_get_profile", [&]() -> int { return er_get_profile_impl(s, out); }); }
int er_accel_info(ErScene* s, ErAccelInfo* out) { return guarded("er_accel_info", [&]() -> int { return er_accel_info_impl(s, out); }); }
}  // extern "C"
