// er_debug.h -- launch wrappers of the inspection / measurement kernels (er_debug.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/eleven_hip_debug.h"
#include "er_bvh.h"

struct DevScene;

// n arbitrary rays through the PRODUCTION traversal (er_trav.h step loop + resolve_closest / resolve_shadow).
// self == nullptr: closest-hit queries; else shadow queries (occluded iff a triangle other than slot self[i] is hit
// nearer than limit[i]).  spill: blocks x ER_STACK8 x 64 uint2 (er_trav.h).
void er_launch_debug_trace(const DevScene& S, const float* o, const float* d, uint32_t n, const int32_t* self, const float* limit,
                           int32_t* tri, int32_t* slot, float* pos, float* dist, int32_t* info, void* spill, hipStream_t stream);
// ONE more sample of pixel idx, one record per executed bounce-loop iteration.  spill: ER_DEBUG_PIXEL_SCRATCH uint2.
// (derived from the depth bounds: the wide traversal's spill levels occupy entries [0, ER_STACK8 x 64), the exact routine's int stack
// -- ER_BVH_MAX_DEPTH levels x 64 ints = half as many uint2 -- follows them; er_debug.hip asserts the two agree)
#define ER_DEBUG_PIXEL_SCRATCH ((ER_STACK8 + ER_BVH_MAX_DEPTH / 2) * 64)
void er_launch_debug_pixel(const DevScene& S, uint32_t idx, ErTraceRec* recs, int max_recs, int* count, void* spill, hipStream_t stream);
// device functions of the path, one item per thread (ER_FN_* of include/eleven_hip_debug.h)
void er_launch_debug_eval(const DevScene& S, int kind, const float* in, uint32_t n, uint32_t in_stride, float* out, uint32_t out_stride, hipStream_t stream);
// streaming kernels for the measured HBM peak: copy (dst = src) and triad-like read-modify-write over n float4.
void er_launch_hbm_copy(const float4* src, float4* dst, size_t n, hipStream_t stream);
void er_launch_hbm_read(const float4* src, float* sink, size_t n, hipStream_t stream);
