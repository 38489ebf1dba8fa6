/* eleven_hip_debug.h -- host-only inspection hooks of libeleven_hip.so (no GPU needed).
 * Not part of the drop-in boundary; used by the CPU test-suite to validate the library's own
 * acceleration-structure builder, which replaces Scene::buildBVH / BVH::build
 * (reference src/Scene.cpp:122-143, src/BVH.cpp:132-415). */
#ifndef ELEVEN_HIP_DEBUG_H
#define ELEVEN_HIP_DEBUG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ErBvhCheck {
    uint32_t node_count, leaf_count, max_depth, max_leaf_size;
    uint32_t tris_in_leaves;       /* must equal tri_count */
    uint32_t duplicate_tris;       /* triangles referenced more than once: must be 0 */
    uint32_t uncontained;          /* vertices outside their leaf's box, or child boxes outside the parent's: must be 0 */
    uint32_t unreachable_nodes;    /* must be 0 */
    float lift_bound;
    float build_ms;
    double sah_cost;               /* sum over inner nodes of child-area * child-count / root area */
} ErBvhCheck;

/* Builds the BVH for [tri_count][3][3] vertices/normals exactly as er_render_begin does and checks its invariants. */
int er_debug_bvh_check(const float* vertices, const float* normals, uint32_t tri_count, int threads, ErBvhCheck* out);

/* Runs the library's HDRI CDF search (elevenrender_amd/csrc/er_cdf.h, the replacement of the 21-level
 * HDRI::binarySearch, reference src/HDRI.cpp:85-98) on the host for `count` values. */
int er_debug_cdf_search(const float* cdf, int length, const float* values, int32_t* out, int count);

/* Closest hit of `n` arbitrary rays (origins/dirs [n][3]; directions taken as given) through the library's exact
 * traversal routine on the GPU, under the reference's metric (src/BVH.cpp:105-120): original triangle id (-1 = miss),
 * Hit.position and |Hit.position - origin|.  Needs a scene on which er_render_begin has succeeded.  Used by the
 * function-level parity test against the oracle's throwRay. */
struct ErScene;
int er_debug_closest_hit(struct ErScene* scene, const float* origins, const float* dirs, uint32_t n, int32_t* tri_ids, float* positions, float* distances);

/* The same query through the PRODUCTION traversal -- the 8-wide interval traversal of er_wf_trace / the streaming tracer waves
 * (csrc/er_trav.h: trav_choose / trav_fetch / trav_apply) followed by resolve_closest / resolve_shadow -- for `n`
 * arbitrary rays (directions taken as given).
 *   self_slots == NULL: closest-hit queries.  tri_ids = original triangle id (-1 miss), slots = the library's internal
 *     triangle handle (feed it back as self_slots), positions = Hit.position, distances = |Hit.position - origin|
 *     (inf on a miss), info = 0 one surviving candidate, 1 two survivors decided by the exact metric, 2 more than two:
 *     exact re-trace.
 *   self_slots != NULL: shadow queries, the reference's "occluded iff the closest hit is a triangle other than the one
 *     the ray leaves" (src/kernel.cpp:555-562) in the form the kernels use: occluded iff some triangle other than
 *     self_slots[i] (-1: none is exempt) is hit nearer than limits[i].  tri_ids = 1 occluded / 0 not; info = 0 / 1
 *     decided by the t-intervals, 2 exact metric of the candidates, 3 exact re-trace. */
int er_debug_trace_rays(struct ErScene* scene, const float* origins, const float* dirs, uint32_t n, const int32_t* self_slots,
                        const float* limits, int32_t* tri_ids, int32_t* slots, float* positions, float* distances, int32_t* info);

/* One record per executed bounce-loop iteration (reference src/kernel.cpp:508-592); same layout as the oracle's
 * OracleTraceRec (oracle/er_oracle.h). */
typedef struct ErTraceRec {
    int32_t bounce;
    int32_t tri;            /* ORIGINAL triangle id of the closest hit, -1 = miss */
    int32_t shadow_tri;     /* original id of the HDRI shadow ray's closest hit, -1 = none / not traced */
    int32_t opaque;         /* 1 if the opacity test passed */
    float position[3];      /* Hit.position */
    float wi[3];            /* next ray direction */
    float light[3];         /* accumulated radiance after this iteration */
    float reduction[3];     /* throughput after this iteration */
    int32_t shadow_occ;     /* HDRI shadow query: 1 occluded, 0 visible, -1 not traced */
    int32_t light_occ;      /* point-light query (ER_FLAG_POINT_LIGHTS): 1 occluded, 0 visible, -1 none */
} ErTraceRec;

/* Runs ONE more sample of pixel idx -- bounce_step (csrc/er_shade.h) over the production traversal, every query traced
 * at once -- and records up to max_recs iterations.  The pixel's planes, sample count and RNG advance exactly as by
 * er_render_samples(scene, 1) restricted to that pixel.  *count = records written. */
int er_debug_trace_pixel(struct ErScene* scene, uint32_t idx, ErTraceRec* recs, int max_recs, int* count);

/* The DEVICE functions of the path, one call per item, for known-answer tests against the oracle's function-level entry
 * points (SURVEY.md section 4 level 1: a1 RNG, a3 camera, a5 Tri::hit record, a6-a8 textures and mappings, a9 HDRI search /
 * pdf, a10-a12 Disney sample / eval / pdf, the transcendentals).  `in` holds n items of in_stride floats, `out` receives n
 * items of out_stride floats (integers travel as float BITS).  Needs a scene on which er_render_begin has succeeded (the
 * camera, textures, HDRI and triangles are the scene's).  Kinds and layouts:
 *   ER_FN_RNG            in [bits(pixel idx)]                          out [16 x next(), 16 x bits(state)]        (32)
 *   ER_FN_CAMERA_RAY     in [x, y, r1, r2, r3, r4, r5]                 out [origin(3), dir(3)]                    (6)
 *   ER_FN_TRI_HIT        in [bits(original tri id), origin(3), dir(3)] out [valid, position(3), normal(3), gnormal(3),
 *                                                                           tangent(3), bitangent(3), tu, tv]     (18)
 *   ER_FN_DISNEY_EVAL    in [hd(20), V(3), N(3), L(3)]                 out [rgb]                                  (3)
 *   ER_FN_DISNEY_PDF     in [hd(20), V(3), N(3), L(3)]                 out [pdf]                                  (1)
 *   ER_FN_DISNEY_SAMPLE  in [hd(20), V(3), N(3), r1, r2, r3]           out [dir(3)]                               (3)
 *     hd = metallic, roughness, clearcoatGloss, clearcoat, anisotropic, transmission, specular, specularTint, sheenTint,
 *          subsurface, sheen, albedo(3), tangent(3), bitangent(3)   -- the oracle's layout (oracle/er_oracle.h)
 *   ER_FN_SPHERICAL      in [p(3)]                                     out [u, v]                                 (2)
 *   ER_FN_REV_SPHERICAL  in [u, v]                                     out [p(3)]                                 (3)
 *   ER_FN_TEXTURE        in [bits(texture id, -1 = the HDRI), u, v, filtered(0/1)]   out [rgb]                    (3)
 *                        (ER_ERR_STATE for a texture the library keeps compacted on the device -- one channel, possibly to the
 *                        power 2.2, er_render_begin -- since a fetch from that entry is not a fetch from the scene's texture;
 *                        begin the render with ER_TEX_COMPACT=0 in the environment to evaluate such a texture)
 *   ER_FN_HDRI_SEARCH    in [value]                                    out [bits(index)]                          (1)
 *   ER_FN_HDRI_PDF       in [bits(x), bits(y)]                         out [pdf]                                  (1)
 *   ER_FN_MATH           in [bits(op: 0 sin 1 cos 2 acos 3 log 4 pow 5 atan2), x, y]   out [value]                (1) */
enum { ER_FN_RNG = 0, ER_FN_CAMERA_RAY, ER_FN_TRI_HIT, ER_FN_DISNEY_EVAL, ER_FN_DISNEY_PDF, ER_FN_DISNEY_SAMPLE, ER_FN_SPHERICAL,
       ER_FN_REV_SPHERICAL, ER_FN_TEXTURE, ER_FN_HDRI_SEARCH, ER_FN_HDRI_PDF, ER_FN_MATH, ER_FN_COUNT };
int er_debug_eval(struct ErScene* scene, int kind, const float* in, uint32_t n, uint32_t in_stride, float* out, uint32_t out_stride);

/* Loopback transport for er_gather_pass: `world` communicators that live in ONE process and move the packed buffers
 * through device-to-device copies, so that pack -> exchange -> unpack can be driven on a one-GPU box without RCCL (which
 * refuses two ranks on one GPU).  out[world].  Call er_gather_pass for the non-root ranks first, then for the root. */
struct ErComm;
int er_debug_comm_create_local(uint32_t world, struct ErComm** out);

/* What the streaming schedule ran the last completed call with, and what it read from it (er_api.cpp er_stream_adapt): for bench.py's
 * `projected` block and the tests.  Valid after er_wait / a read-back; all zero for the other schedules. */
typedef struct ErStreamInfo {
    uint32_t waves, tracers;       /* waves per workgroup (16 or 12) and how many of them trace */
    uint32_t large_regions;        /* 1: the deal of 16 x 16-tile screen regions per XCD is in use; 0: the default 8 x 8 one */
    uint32_t deal_pending;         /* 1: the deal is still to be decided (no call has completed) */
    uint32_t launches;             /* kernel launches of this render so far */
    uint32_t pixels_per_cu;        /* owned pixels per workgroup (tiles x 64 / workgroups) */
    double lanes_busy;             /* share of the tracer waves' lanes that held a ray, last completed launch */
    double launch_ms;              /* device time of that launch */
    double cost_spread;            /* (max - min) / mean of the XCDs' counted work under the large deal; < 0: not decided */
    uint64_t spec_started, spec_right, spec_wrong;   /* speculative samples (small shares) started / whose guessed RNG state was right / wrong, completed launches of this render */
    uint32_t form;                 /* the kernel's form this share is launched in: 0 plain (whole frames), 1 pixels that are behind keep their slots, 2 that and speculative samples */
    uint32_t reserved;
} ErStreamInfo;
int er_debug_stream_info(struct ErScene* s, ErStreamInfo* out);

/* The buffers er_gather_pass keeps on a scene (round 5: allocated by the first gather, reused by every later one): the receive
 * buffer for `peer`'s pixels on the root (*in_ptr, *in_bytes; NULL / 0 before the first gather or for peer == own rank) and the
 * packed send buffer of a non-root rank (*mine_ptr, *mine_bytes).  Device pointers, for identity comparison only. */
int er_debug_gather_buffers(struct ErScene* s, uint32_t peer, void** in_ptr, uint64_t* in_bytes, void** mine_ptr, uint64_t* mine_bytes);

/* `bytes` of a known pattern from this rank to ITSELF through the communicator's transport table -- group start, send(self),
 * recv(self), group end on a stream of the library's own -- and compared after the stream has drained: the one exchange RCCL
 * allows on a one-GPU box (it refuses two ranks on one device), so that ncclSend / ncclRecv of the dlopen'd library have carried
 * bytes from the C++ side before the first multi-GPU run.  *ms = wall time of the exchange.  ER_ERR_STATE if the bytes differ. */
int er_debug_comm_loopback(struct ErComm* c, uint64_t bytes, double* ms);

/* The streaming schedule's deal of a rank's owned 8 x 8 tiles (`owned`: tile indices ty * tiles_x + tx) to `blocks` workgroups, as
 * er_render_begin makes it (host code, no device needed): out[b + k * blocks] = the k-th tile of workgroup b or 0xFFFFFFFF, for
 * k < *most; workgroups b and b + 8 share an XCD.  edge = side of a super-tile in tiles, 0 = the library's default.  out_cap >= blocks * *most
 * (a first call with out_cap = 0 fails with ER_ERR_INVALID_ARG after setting *most). */
int er_debug_stream_deal(const uint32_t* owned, uint32_t count, uint32_t tiles_x, uint32_t blocks, int xcd_aware, uint32_t edge, uint32_t* out, uint32_t out_cap,
                         uint32_t* most);

/* Test hook for the out-of-memory path of the boundary: while `bytes` is non-zero, any single large host allocation the
 * library announces (scene copy in er_scene_create, build staging in er_render_begin) larger than `bytes` fails as
 * std::bad_alloc would, which the entry point must turn into ER_ERR_OOM (no exception crosses the C ABI).  0 = off. */
void er_debug_set_host_alloc_limit(uint64_t bytes);

/* Test hook for er_render_begin's handling of a device BVH build that does not deliver: 0 = off; 1 = every device build fails as a
 * fault inside the builder would (default builder: one line on stderr and the host builder takes over; ER_FLAG_GPU_BUILD: ER_ERR_HIP);
 * 2 = as running out of device memory would (default builder: silent host fallback; ER_FLAG_GPU_BUILD: ER_ERR_OOM). */
void er_debug_set_gpu_build_failure(int kind);

#ifdef __cplusplus
}
#endif
#endif
