/* eleven_hip.h -- C ABI of the MI355X path-tracing core (libeleven_hip.so).
 *
 * This is the drop-in boundary for ElevenRender's per-sample hot path.  It replaces
 * the SYCL seam of the reference between RenderingManager and the device runtime:
 *
 *   reference interface (file:line under the reference tree)        replaced by
 *   --------------------------------------------------------------  ----------------------
 *   dev_Scene::dev_Scene(Scene*)            src/kernel.cpp:244-266   er_scene_create
 *   copy_scene(dev_Scene*,dev_Scene*,queue) src/SYCLCopy.cpp:3-104   er_scene_create + er_render_begin
 *   renderSetup(q,scene,dev,spp,bs)         src/kernel.cpp:651-678   er_render_begin
 *   kernel_render_enqueue(q,spp,bs,...)     src/kernel.cpp:680-706   er_render_samples
 *   renderingKernel(dev_Scene*,idx,samples) src/kernel.cpp:477-646   (the HIP kernels behind er_render_samples)
 *   RenderingManager::get_pass(string)      src/Managers.cpp:287-302 er_read_pass
 *   RenderingManager::get_render_info()     src/Managers.cpp:211-232 er_samples_done
 *   is_compatible(sycl::device&)            src/kernel.cpp:708-720   er_device_info().compatible
 *   get_sycl_info device enumeration        src/CommandManager.cpp:303-362  er_device_count / er_device_info
 *   NameSelector "name|platform"            src/Managers.cpp:191-208 er_device_find
 *
 * Conventions: plain C, PODs and raw pointers only; every entry point returns
 * ER_OK (0) or a negative ErStatus and never throws; er_last_error() gives the
 * text for the calling thread.  All host pointers in ErSceneDesc stay owned by the
 * caller and are copied during er_scene_create.  The library fails loudly
 * (ER_ERR_NO_DEVICE) when no gfx950 device / HIP runtime is usable: there is no
 * CPU fallback behind this ABI.
 */
#ifndef ELEVEN_HIP_H
#define ELEVEN_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: ErCounters grew the four trace_* fields and ErProfile the empty-launch fields (round 2); a host built against the
 * round-1 header must not pass the version check, since the library fills both caller-allocated structs completely. */
#define ER_ABI_VERSION 2

typedef enum ErStatus {
    ER_OK = 0,
    ER_ERR_INVALID_ARG = -1,
    ER_ERR_NO_DEVICE = -2,
    ER_ERR_HIP = -3,
    ER_ERR_STATE = -4,     /* call order violated (e.g. render before begin) */
    ER_ERR_OOM = -5
} ErStatus;

/* Pass planes, reference enum Passes (src/kernel.h:8). DENOISE is allocated and
 * initialised like the others but never written by the kernel (src/kernel.cpp:604). */
typedef enum ErPass { ER_PASS_BEAUTY = 0, ER_PASS_DENOISE = 1, ER_PASS_NORMAL = 2,
                      ER_PASS_TANGENT = 3, ER_PASS_BITANGENT = 4, ER_PASS_COUNT = 5 } ErPass;

typedef struct ErVec3 { float x, y, z; } ErVec3;

/* reference Camera (src/Camera.h:5-25); rotation in degrees, XYZ Euler. */
typedef struct ErCamera {
    float focal_length, sensor_width, sensor_height, aperture, focus_distance;
    ErVec3 rotation;
    int32_t bokeh;
    ErVec3 position;
} ErCamera;

/* reference Material without its strings (src/Material.h:20-47). Texture ids < 0 = constant. */
typedef struct ErMaterial {
    int32_t albedo_tex, emission_tex, roughness_tex, metallic_tex, normal_tex, opacity_tex,
            transmission_tex;
    int32_t albedo_shader_id;          /* -1 = none; see asl_shade, src/shader.cpp:6-10 */
    ErVec3 albedo, emission;
    float opacity, roughness, metallic, clearcoat_gloss, clearcoat, anisotropic, eta,
          transmission, specular, specular_tint, sheen_tint, subsurface, sheen, ax, ay;
} ErMaterial;

/* reference Texture (src/Texture.h:12-73): interleaved f32, row-major, `channels` per texel.
 * filter: 0 = NO_FILTER, 1 = BILINEAR. */
typedef struct ErTexture {
    int32_t width, height, channels, filter;
    const float* data;
} ErTexture;

/* reference HDRI (src/HDRI.h:9-42). cdf has width*height+1 entries; if NULL the library
 * builds it exactly as HDRI::generateCDF does (src/HDRI.cpp:62-83) and radiance_sum is ignored. */
typedef struct ErHdri {
    ErTexture texture;
    const float* cdf;
    float radiance_sum;
} ErHdri;

/* reference PointLight (src/PointLight.h:4-16). The reference never evaluates point
 * lights (src/kernel.cpp:269-301 has no caller); they are used only when
 * ErRenderParams.flags has ER_FLAG_POINT_LIGHTS (a build-defined extension). */
typedef struct ErPointLight { ErVec3 position, radiance; } ErPointLight;

/* Flattened reference Scene (src/Scene.h:24-73, Tri src/Tri.h:8-21). Triangle arrays are
 * indexed [tri][corner][component]. */
typedef struct ErSceneDesc {
    uint32_t tri_count;
    const float* vertices;       /* [tri_count][3][3] */
    const float* normals;        /* [tri_count][3][3] */
    const float* tangents;       /* [tri_count][3][3] */
    const float* uvs;            /* [tri_count][3][2] */
    const float* tangent_sign;   /* [tri_count] */
    const int32_t* material_id;  /* [tri_count] */
    uint32_t material_count;
    const ErMaterial* materials;
    uint32_t texture_count;
    const ErTexture* textures;
    ErHdri hdri;
    ErCamera camera;
    uint32_t point_light_count;
    const ErPointLight* point_lights;
    uint32_t x_res, y_res;
} ErSceneDesc;

#define ER_FLAG_POINT_LIGHTS 1u   /* extension, default off = reference behaviour: one point-light sample per opaque
                                     bounce after the author's sketch src/kernel.cpp:269-301 (rules: csrc/er_shade.h) */
#define ER_FLAG_MIS          128u /* extension, default off = reference behaviour (NEE and BRDF-sampled environment both
                                     counted in full, src/kernel.cpp:571-577): balance-heuristic weights per direction */
#define ER_FLAG_COUNTERS     2u   /* count node visits / triangle tests (slower kernel variant) */
/* Schedule selection (every schedule computes bit-identical results).  Default: the streaming schedule, at every frame size since
 * round 5 (C1 at 256 x 256: 1 186 Msamples/s against 849 for round 1's fused kernel, which was removed), except for a rank that owns
 * more pixels than the streaming schedule's pixel rings hold (8.4 M): the wavefront schedule.  The flags force one. */
#define ER_FLAG_MEGAKERNEL   4u   /* v0: one wave per 8x8 tile, wave-synchronous bounce loop over the binary BVH */
#define ER_FLAG_FUSED        16u  /* (round 1's lane-asynchronous single kernel, removed in round 5: accepted, means ER_FLAG_STREAM) */
#define ER_FLAG_WAVEFRONT    32u  /* one trace + one shade launch per bounce over compacted ray queues */
#define ER_FLAG_STREAM       256u /* one launch per call, one resident workgroup per CU: tracer waves and shader waves feed each
                                     other through rings in LDS (er_stream.hip) */
#define ER_FLAG_PROFILE      8u   /* bracket every trace / shade launch with HIP events (see er_get_profile) */
/* Acceleration-structure builder.  Default since round 5: the DEVICE build (top-down binned SAH on the GPU, the host builder's own
 * algorithm and tree: 65 ms instead of 0.43 s at 1 M triangles, 0.33 s instead of 5.0 s at 10 M) for scenes of at least
 * ER_GPU_BUILD_MIN_TRIS (20 000; environment variable of that name) triangles, the host build below that; a device build that
 * declines or fails falls back to the host build.  Images do not depend on the builder. */
#define ER_FLAG_GPU_BUILD    64u  /* force the device build whatever the triangle count (an error of it is then the call's error) */
#define ER_FLAG_HOST_BUILD   512u /* force the host build (er_bvh.cpp) */

/* reference RenderParameters (src/kernel.h:51-69) + what the MI355X build adds. */
typedef struct ErRenderParams {
    uint32_t sample_target;   /* informational, as in the reference */
    uint32_t block_size;      /* accepted for protocol compatibility; launch geometry is ours */
    uint32_t max_bounces;     /* 0 -> 5, the literal of src/kernel.cpp:508 */
    int32_t device;           /* HIP device ordinal */
    uint32_t rank, world;     /* pixel-tile shard: this process renders tiles t with t % world == rank; world 0 -> 1 */
    uint32_t flags;
} ErRenderParams;

typedef struct ErDeviceInfo {
    char name[256];
    char platform[64];        /* "AMD HIP" */
    uint64_t memory_bytes;
    uint32_t compute_units;
    int32_t compatible;       /* 1 iff gcnArchName starts with gfx950 */
    char arch[64];
} ErDeviceInfo;

/* Counters of the work done since er_render_begin (all ranks count only their own pixels). */
typedef struct ErCounters {
    uint64_t paths;            /* pixel-samples finished */
    uint64_t bounce_samples;   /* executed iterations of the bounce loop = the Msamples metric unit */
    uint64_t rays;             /* closest-hit + shadow traversals started */
    uint64_t node_visits;      /* only with ER_FLAG_COUNTERS */
    uint64_t tri_tests;        /* only with ER_FLAG_COUNTERS */
    uint64_t shaded_hits;      /* closest hits shaded */
    uint64_t texel_fetches;    /* only with ER_FLAG_COUNTERS */
    uint64_t hdri_samples;     /* CDF searches */
    /* lane occupancy of the wavefront traversal loop (only with ER_FLAG_COUNTERS): iterations of er_wf_trace's loop summed
     * over its waves, and how many of the 64 lanes held a ray / ran the node part / ran the triangle part in them */
    uint64_t trace_wave_steps, trace_busy_lanes, trace_node_lanes, trace_tri_lanes;
} ErCounters;

typedef struct ErScene ErScene;   /* opaque */

int er_abi_version(void);
const char* er_last_error(void);

int er_device_count(void);
int er_device_info(int index, ErDeviceInfo* out);
/* index of the device whose "name|platform" equals selector, or a negative status. */
int er_device_find(const char* selector);

int er_scene_create(const ErSceneDesc* desc, ErScene** out);
void er_scene_destroy(ErScene* scene);

/* Builds the acceleration structure, uploads, initialises passes/samples/RNG (setupKernel,
 * src/kernel.cpp:176-213, for EVERY pixel). May be called again to restart a render. */
int er_render_begin(ErScene* scene, const ErRenderParams* params);

/* Adds n samples to every owned pixel. Blocking. Equivalent to n launches of renderingKernel. */
int er_render_samples(ErScene* scene, uint32_t n);
/* Non-blocking form + explicit wait; elapsed_ms (may be NULL) = device time of everything
 * enqueued since the previous er_wait, measured with HIP events on the library's stream.
 * (Streaming schedule, the first call of a render with n >= 2: its first sample is a launch of its own that the
 * call waits for -- the library then decides, from the work counted in it, which screen regions go to which XCD --
 * before the other n - 1 samples are enqueued: that one call blocks for one sample pass.) */
int er_render_samples_async(ErScene* scene, uint32_t n);
int er_wait(ErScene* scene, float* elapsed_ms);

/* reference get_render_info semantic: dev_samples[0], i.e. samples added + 1. */
int er_samples_done(ErScene* scene, uint32_t* out);

/* Copies one pass plane (x_res*y_res*4 floats, RGBA, row-major, idx = y*x_res + x) to host memory.
 * With world > 1 only owned pixels are valid; others are left at the setup value (0,0,0,1). */
int er_read_pass(ErScene* scene, int pass, float* dst_rgba);
int er_read_samples(ErScene* scene, uint32_t* dst);

/* Checkpoint / resume (the reference keeps the progressive estimator only in device memory --
 * dev_passes + dev_samples + dev_randstate, src/kernel.h:44-46 -- and these three arrays ARE a complete resumable state).
 * er_state_size: bytes of one snapshot of this scene (5 planes of float4 + 2 u32 per pixel + a 64-byte header).
 * er_state_export: a sample-boundary snapshot into host memory (ordered after everything enqueued, like er_read_pass).
 * er_state_import: after er_render_begin on a scene of the same resolution, continue from a snapshot: the next
 * er_render_samples(n) gives exactly what n more samples would have given in the run that exported it (per-pixel RNG
 * streams; any schedule, any rank/world split -- a rank only ever touches the pixels it owns). */
int er_state_size(ErScene* scene, uint64_t* bytes);
int er_state_export(ErScene* scene, void* dst, uint64_t bytes);
int er_state_import(ErScene* scene, const void* src, uint64_t bytes);

/* Fills the DENOISE plane (which renderingKernel never writes, src/kernel.cpp:604) with an edge-avoiding a-trous
 * wavelet filter of the current BEAUTY plane, guided by colour and by the NORMAL plane: `levels` passes (1..8, 0 -> 5)
 * with stencil holes of 1, 2, 4, ... pixels; colour_sigma > 0 scales the colour edge-stop (0 -> 1).  Stands in for the
 * reference's host-side OIDN call behind `get_pass denoise` (src/Managers.cpp:319-343, CommandManager.cpp:265-274) --
 * a different filter: no parity claim.  On a frame sharded over several ranks (world > 1) it runs on the rank that BEAUTY
 * and NORMAL have been gathered to since the last sample (er_gather_pass, or er_unpack_owned of every other rank), as
 * the reference denoises where the whole pass is; elsewhere ER_ERR_STATE. */
int er_denoise(ErScene* scene, uint32_t levels, float colour_sigma);     /* x_res*y_res */
int er_read_rng(ErScene* scene, uint32_t* dst);         /* x_res*y_res */

/* Multi-GPU framebuffer combine helpers (the collective itself is done by the host with
 * RCCL; these move owned pixels between the full plane and a compact device buffer).
 * er_owned_count: number of pixels this rank owns.
 * er_pack_owned: dev_dst[owned][4] <- owned pixels of `pass`, in owned-pixel order (device pointer).
 * er_unpack_owned: scatters a compact buffer of rank `src_rank` into this scene's full plane. */
int er_owned_count(ErScene* scene, uint32_t rank, uint64_t* out);
int er_pack_owned(ErScene* scene, int pass, void* dev_dst);
int er_unpack_owned(ErScene* scene, int pass, uint32_t src_rank, const void* dev_src);

/* The combine itself, from the C++ side (north_star: "RCCL reduce over xGMI only for the final framebuffer accumulate";
 * reference hook: RenderingManager::get_pass, src/Managers.cpp:287-302).  One process per GPU:
 *   rank 0: er_comm_unique_id(id) -> the host hands the 128 bytes to the other ranks through any channel it has
 *           (the reference's host would use its TCP session; bench.py uses torch.distributed's broadcast)
 *   all:    er_comm_create(id, rank, world, device, &comm)        -- ncclCommInitRank; RCCL is dlopen'ed at this point
 *   all:    er_gather_pass(scene, pass, comm, root)               -- every rank packs the pixels it owns; the non-root
 *           ranks ncclSend, the root ncclRecv's inside ONE group (xGMI is point to point: the seven links carry the
 *           seven buffers side by side) and scatters them into its full plane.  One call per read-back, none per sample.
 *   all:    er_comm_destroy(comm)
 * The scene must have been begun with the communicator's rank / world. */
typedef struct ErComm ErComm;
#define ER_COMM_ID_BYTES 128
int er_comm_unique_id(uint8_t id[ER_COMM_ID_BYTES]);
int er_comm_create(const uint8_t id[ER_COMM_ID_BYTES], uint32_t rank, uint32_t world, int device, ErComm** out);
/* The same for the ranks of ONE process (a host that drives several GPUs itself, reference hook: RenderingManager with N
 * devices; also several ranks on one GPU): `world` communicators over an in-process transport -- a send parks a device copy,
 * the matching receive copies it device to device (peer copy over xGMI between two GPUs); no RCCL involved.  out[world].
 * er_gather_pass may be called by the ranks in any order, from one thread per rank or (non-roots first) from one thread. */
int er_comm_create_local(uint32_t world, ErComm** out);
void er_comm_destroy(ErComm* comm);
int er_gather_pass(ErScene* scene, int pass, ErComm* comm, uint32_t root);

int er_get_counters(ErScene* scene, ErCounters* out);

/* Per-kernel device time of the launches enqueued since the previous er_wait, measured with HIP events on
 * the library's stream (needs ER_FLAG_PROFILE; valid after er_wait). */
typedef struct ErProfile {
    float trace_ms, shade_ms;          /* summed over the launches that had rays to trace */
    uint32_t trace_launches, shade_launches;   /* ... and their number */
    uint32_t schedule;                 /* ER_FLAG_WAVEFRONT / ER_FLAG_FUSED / ER_FLAG_MEGAKERNEL actually in use; for the
                                          two single-kernel schedules trace_ms is that kernel's time and shade_ms is 0 */
    uint32_t concurrency;              /* wavefront schedule: number of slot pools whose launches run side by side on
                                          their own streams (their durations overlap, so trace_ms + shade_ms exceeds the
                                          elapsed time by up to this factor); 1 otherwise */
    uint32_t empty_launches;           /* wavefront schedule: trace + shade pairs that found empty queues (the host loop always
                                          enqueues n * (max_bounces + 1) iterations), and their summed duration */
    float empty_ms;
    uint64_t rays_logged;              /* wavefront schedule: rays the counted trace launches found in their queues */
} ErProfile;
int er_get_profile(ErScene* scene, ErProfile* out);

/* Measured HBM bandwidth of `device`, for the roofline's denominator ("the denominator used is the
 * measured peak from a device-to-device copy / triad kernel run in the same job"): a streaming copy and a streaming
 * read over `bytes` of device memory (0 -> 2 GiB, several times the 256 MB Infinity Cache), best of `iters` (0 -> 5)
 * timed with HIP events.  copy counts bytes read + bytes written. */
int er_measure_hbm_peak(int device, uint64_t bytes, uint32_t iters, float* copy_GBps, float* read_GBps);

/* Description of the built acceleration structure, for roofline accounting. */
typedef struct ErAccelInfo {
    uint32_t node_count, node_bytes;   /* wide nodes */
    uint32_t leaf_count, max_depth;
    uint32_t tri_record_bytes;         /* bytes fetched per triangle test */
    float build_ms, upload_ms;
    float lift_bound;                  /* global bound on |shadingPosition - geomPosition| */
    uint32_t builder;                  /* 0 = host binned-SAH build, 1 = device binned-SAH build */
} ErAccelInfo;
int er_accel_info(ErScene* scene, ErAccelInfo* out);

#ifdef __cplusplus
}
#endif
#endif /* ELEVEN_HIP_H */
