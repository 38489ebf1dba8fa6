/* er_oracle.h -- C entry points of the CPU oracle (liberoracle.so).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (elevenrender_amd/, include/) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load it -- as the checker / CPU baseline, never as the thing shipped.
 *
 * PARITY UNPINNED: the reference (101001000/ElevenRender) has no tests, golden vectors or
 * fixtures for this path, and its sources cannot be compiled in this image (they need a
 * SYCL toolchain + Boost; stand-in headers are not allowed).  This oracle is therefore a
 * line-by-line restatement of the reference algorithm from reading its source, pinned only
 * by published known-answer values of the public algorithms it uses (Jenkins OAAT,
 * Marsaglia xorshift32) and by analytic properties; see DESIGN.md "Oracle".
 */
#ifndef ER_ORACLE_H
#define ER_ORACLE_H
#include "../include/eleven_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleOpts {
    int32_t math_mode;    /* 0 = glibc libm (what a CPU build of the reference gets), 1 = er_math.h */
    int32_t max_bounces;  /* 0 -> 5 (src/kernel.cpp:508) */
    int32_t traversal;    /* 0 = reference BVH (fixed depth 18, src/BVH.cpp), 1 = brute force over all tris */
    int32_t threads;      /* worker threads over pixel rows; <=0 -> 1 */
    uint32_t flags;       /* ER_FLAG_POINT_LIGHTS | ER_FLAG_MIS: the two build-defined extensions, mirrored operation for
                             operation from elevenrender_amd/csrc/er_shade.h (the reference defines no result for either) */
} OracleOpts;

typedef struct OracleCounters {
    uint64_t paths, bounce_samples, rays;
    uint64_t node_visits;   /* inner-node iterations of BVH::transverse (two box tests each) */
    uint64_t tri_tests, tri_hits, shaded_hits, texel_fetches, hdri_samples;
} OracleCounters;

/* one record per executed bounce-loop iteration */
typedef struct OracleTraceRec {
    int32_t bounce;
    int32_t tri;            /* ORIGINAL triangle id of the closest hit, -1 = miss */
    int32_t shadow_tri;     /* original id of the shadow ray's closest hit, -1 = none / not traced */
    int32_t opaque;         /* 1 if the opacity test passed */
    float position[3];
    float wi[3];            /* next ray direction */
    float light[3];
    float reduction[3];
    int32_t shadow_occ;     /* HDRI shadow query: 1 if shadowHit.valid && shadowHit.triIdx != hitdata.triIdx, 0 if not, -1 not traced */
    int32_t light_occ;      /* point-light query (extension): 1 occluded, 0 visible, -1 none */
} OracleTraceRec;

typedef struct Oracle Oracle;

Oracle* oracle_create(const ErSceneDesc* desc, const OracleOpts* opts);
void oracle_destroy(Oracle* o);
double oracle_build_seconds(const Oracle* o);

/* n_samples calls of renderingKernel for every pixel idx in [idx0, idx1) (idx1 = 0 -> all). */
void oracle_render(Oracle* o, uint32_t n_samples, uint32_t idx0, uint32_t idx1);
void oracle_read_pass(const Oracle* o, int pass, float* dst_rgba);
void oracle_read_samples(const Oracle* o, uint32_t* dst);
void oracle_read_rng(const Oracle* o, uint32_t* dst);
void oracle_counters(const Oracle* o, OracleCounters* out);

/* Runs ONE more sample of pixel idx and records up to max_recs bounce records; returns count. */
int oracle_trace_pixel(Oracle* o, uint32_t idx, OracleTraceRec* recs, int max_recs);

/* closest hit of the reference traversal for arbitrary rays: out_tri = original tri id or -1,
 * out_pos[3] = Hit.position */
void oracle_closest_hit(Oracle* o, const float* origins, const float* dirs, int n,
                        int32_t* out_tri, float* out_pos);

/* ---- function-level entry points (for known-answer tests) ---- */
uint32_t oracle_jenkins_oaat_u32(uint32_t seed);
uint32_t oracle_jenkins_oaat_bytes(const uint8_t* key, size_t len);   /* the published general form */
void oracle_rng_stream(uint32_t pixel_idx, int n, uint32_t* states, float* values);
void oracle_xorshift32(uint32_t* state);
void oracle_camera_ray(const ErCamera* cam, uint32_t x_res, uint32_t y_res, int x, int y,
                       const float r[5], int math_mode, float out_origin[3], float out_dir[3]);
/* tri = 9 verts, 9 normals, 9 tangents, 6 uvs, sign; returns hit flag and fills
 * out[0..2]=position, [3..5]=normal, [6..8]=gnormal, [9..11]=tangent, [12..14]=bitangent, [15]=tu, [16]=tv */
int oracle_tri_hit(const float* verts, const float* normals, const float* tangents, const float* uvs,
                   float tangent_sign, const float origin[3], const float dir[3], float out[17]);
int oracle_box_hit(const float origin[3], const float dir[3], const float b1[3], const float b2[3]);
/* hd[]: metallic, roughness, clearcoatGloss, clearcoat, anisotropic, transmission, specular,
 * specularTint, sheenTint, subsurface, sheen, albedo[3], tangent[3], bitangent[3]  (20 floats) */
void oracle_disney_eval(const float hd[20], const float V[3], const float N[3], const float L[3],
                        int math_mode, float out[3]);
float oracle_disney_pdf(const float hd[20], const float V[3], const float N[3], const float L[3], int math_mode);
void oracle_disney_sample(const float hd[20], const float V[3], const float N[3], float r1, float r2, float r3,
                          int math_mode, float out[3]);
void oracle_spherical_mapping(const float p[3], int math_mode, float* u, float* v);
void oracle_reverse_spherical_mapping(float u, float v, int math_mode, float out[3]);
void oracle_texture_fetch(const ErTexture* tex, float u, float v, int filtered, float out[3]);
void oracle_hdri_cdf(const ErTexture* tex, float* cdf, float* radiance_sum);
int oracle_hdri_binary_search(const float* cdf, float value, int length);
float oracle_hdri_pdf(const ErTexture* tex, float radiance_sum, int x, int y, int math_mode);
/* math: kind 0 sin,1 cos,2 acos,3 log,4 pow(x,y),5 atan2(x=y_arg,y=x_arg) ; mode as above */
void oracle_math(int kind, int math_mode, const float* x, const float* y, float* out, int n);

#ifdef __cplusplus
}
#endif
#endif
