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

#ifdef __cplusplus
}
#endif
#endif
