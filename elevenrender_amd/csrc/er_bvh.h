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

#define ER_BVH_MAX_DEPTH 64   // also the stack depth per lane of the binary-tree routines (32 until round 5: the device builder's trees of 10 M triangles are deeper)
// Stack levels of the WIDE traversal (one entry per level of the 8-wide tree at most): the first WF_LDS_STACK (er_trav.h) of them in LDS, the others in the HBM spill
// area, which is sized for all of them.  The wide tree is far shallower than the binary one it is collapsed from (C2: 9 levels, C4: 11; the binary
// routines' own stacks stay ER_BVH_MAX_DEPTH deep, ER_STACK in er_device.h); er_render_begin refuses a scene whose wide tree is deeper.
#ifndef ER_STACK8
#define ER_STACK8 32
#endif
// uint2 entries of ONE wave's spill region where a wave may play either role (streaming schedule, debug hooks): the wide traversal's levels or the
// exact binary re-trace's int stack (ER_BVH_MAX_DEPTH levels x 64 lanes, two ints per entry)
#define ER_SPILL_PER_WAVE ((ER_STACK8 * 64) > (ER_BVH_MAX_DEPTH * 32) ? (ER_STACK8 * 64) : (ER_BVH_MAX_DEPTH * 32))

#ifndef ER_BVH_LEAF_MAX
#define ER_BVH_LEAF_MAX 2   // measured on C2: 2 -> 838, 3 -> 792, 4 -> 764 Msamples/s (fewer triangle fetches per ray)
#endif
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
    float v1[3]; float lift;     // bound on |shadingPosition - geomPosition| for this triangle (src/Tri.h:106-117)
    float v2[3]; float sign;
};
static_assert(sizeof(ErTriIsect) == 48, "isect record must be 48 bytes");

#ifndef ER_ATTR_PIECES
#define ER_ATTR_PIECES 7    // 16-byte pieces per attribute record: 7 = 112 bytes packed, 8 = one record per 128-byte line
#endif
struct ErTriAttr {          // 112 bytes (+ padding to ER_ATTR_PIECES x 16): read for candidates (normals) and for the shaded hit (all)
    float n[3][3];          // 36
    float t[3][3];          // 36
    float uv[3][2];         // 24
    int32_t material;       // 4
    float pad[3 + 4 * (ER_ATTR_PIECES - 7)];           // -> 112 / 128
};
static_assert(sizeof(ErTriAttr) == 16 * ER_ATTR_PIECES, "attr record must be ER_ATTR_PIECES pieces");

// 8-wide compressed node (layout after Ylitie, Karras, Laine 2017, "Efficient incoherent ray traversal on
// GPUs through compressed wide BVHs"): child boxes are 8-bit offsets from `p` in units of 2^e per axis,
// quantised OUTWARD (decoded box always contains the float box), inner children are consecutive nodes from
// child_base, leaf children's triangles are consecutive slots from tri_base.  Children sit in slots whose
// 3-bit index encodes their position relative to the node centre, so `slot ^ octant` orders a ray's visits.
struct ErNode8 {            // 80 bytes = five 16-byte loads
    float p[3];
    uint8_t e[3];           // biased exponents: scale = 2^(e - 127) as an IEEE bit pattern (e << 23)
    uint8_t imask;          // bit s: slot s holds an inner node
    uint32_t child_base;
    uint32_t tri_base;
    uint32_t tri_present;   // bit 2*s + j: slot s is a leaf with more than j triangles (leaves hold <= 2)
    uint32_t reserved;      // the node's triangles are stored compactly in bit order from tri_base
    uint8_t qlo[3][8];
    uint8_t qhi[3][8];
};
static_assert(sizeof(ErNode8) == 80, "wide node must be 80 bytes");
// Stride of the wide nodes in the traversal buffer, in 16-byte pieces: 5 = packed (80 B; every second node straddles two
// 128-byte cache lines), 8 = one node per 128-byte line.
#ifndef ER_NODE8_PIECES
#define ER_NODE8_PIECES 5
#endif
static_assert(ER_BVH_LEAF_MAX <= 2, "ErNode8::tri_present has two bits per child slot");

struct ErBvhBuild {
    std::vector<ErNode8> nodes8;         // 8-wide compressed tree over the same triangle order
    uint32_t max_depth8 = 0;
    std::vector<ErNode> nodes;
    std::vector<uint32_t> slot_to_tri;   // leaf order -> original triangle id
    uint32_t leaf_count = 0;
    uint32_t max_depth = 0;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};   // scene bounds
    float lift_bound = 0;                // max over tris of the bound on |shadingPosition - geomPosition|
    std::vector<float> tri_lift;         // the same bound per ORIGINAL triangle id
    double build_ms = 0;
};

// collapse of ErBvhBuild::nodes (root = node 0) into nodes8; re-orders slot_to_tri and the binary tree's leaf
// references into the wide tree's leaf order.  Called by er_build_bvh; public for the device builder.
void er_collapse_bvh8(ErBvhBuild* out);

// vertices/normals: [tri][3][3].  threads <= 0 -> hardware concurrency.
void er_build_bvh(const float* vertices, const float* normals, uint32_t tri_count, int threads, ErBvhBuild* out);
