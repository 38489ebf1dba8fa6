// er_bvh.h -- acceleration structure of the MI355X build (host-side builder + device layout).
//
// The reference builds a fixed-depth-18 complete binary tree (src/BVH.cpp:132-415) and
// traverses it without ordering or pruning (src/BVH.cpp:63-103).  The contract of this
// path is only "same nearest hit under the reference's metric" (SURVEY.md section 8 a4/a5), so
// this build ships its own structure:
//
//   * binary BVH, binned SAH, variable depth (bounded by ER_BVH_MAX_DEPTH so the per-lane
//     LDS traversal stack can never overflow), leaves of <= ER_BVH_LEAF_MAX triangles;
//   * 64-byte nodes holding BOTH children's boxes, so one aligned 64 B read decides two
//     box tests (the reference copies two 44 B nodes per step);
//   * triangles re-ordered into leaf order ("slots"): a 48 B intersection record (3 vertices
//     + the original id) and a 112 B attribute record fetched only for hits.
#pragma once
#include <stdint.h>
#include <vector>

#define ER_BVH_MAX_DEPTH 32   // also the LDS stack depth per lane
#define ER_BVH_LEAF_MAX 4
#define ER_BVH_NO_CHILD 0x7fffffff

// child reference: >= 0 inner node index; < 0 leaf: ~((first_slot << 3) | (count-1)); ER_BVH_NO_CHILD = empty
struct ErNode {             // 64 bytes, 64-byte aligned
    float lo0[3], hi0[3];   // child 0 box
    float lo1[3], hi1[3];   // child 1 box
    int32_t c0, c1;
    int32_t pad[2];
};
static_assert(sizeof(ErNode) == 64, "node must be 64 bytes");

struct ErTriIsect {         // 48 bytes: what a triangle test reads
    float v0[3]; int32_t tri_id;
    float v1[3]; int32_t material;
    float v2[3]; float sign;
};
static_assert(sizeof(ErTriIsect) == 48, "isect record must be 48 bytes");

struct ErTriAttr {          // 112 bytes: read for candidates (normals) and for the shaded hit (all)
    float n[3][3];          // 36
    float t[3][3];          // 36
    float uv[3][2];         // 24
    float pad[4];           // -> 112
};
static_assert(sizeof(ErTriAttr) == 112, "attr record must be 112 bytes");

struct ErBvhBuild {
    std::vector<ErNode> nodes;
    std::vector<uint32_t> slot_to_tri;   // leaf order -> original triangle id
    uint32_t leaf_count = 0;
    uint32_t max_depth = 0;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};   // scene bounds
    float lift_bound = 0;                // max over tris of the bound on |shadingPosition - geomPosition|
    double build_ms = 0;
};

// vertices/normals: [tri][3][3].  threads <= 0 -> hardware concurrency.
void er_build_bvh(const float* vertices, const float* normals, uint32_t tri_count, int threads, ErBvhBuild* out);
