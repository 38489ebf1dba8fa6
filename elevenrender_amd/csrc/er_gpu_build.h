// er_gpu_build.h -- device-side BVH builder (er_gpu_build.hip), SURVEY.md 8(f) rank 1.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <atomic>
#include <string>

#include "er_bvh.h"

#ifndef ER_GPU_BUILD_MIN_TRIS
#define ER_GPU_BUILD_MIN_TRIS 20000u   // scenes of fewer triangles take the host build by default (a few ms there; the device build is a few ms of launches whatever the size)
#endif

// Whole structure on the device: binary tree, SAH-optimal collapse into 8-wide nodes, final slot order and the
// triangle records, in the buffers er_render_begin hands to the kernels.  Nothing but a few counters comes back.
// Returns 0 on success; > 0 = the device builder declines (too few triangles, tree deeper than the traversal stack
// bound) and the caller should use er_build_bvh; -2 = out of device memory, -1 = any other HIP error or a guard of the builder
// itself (a fault: the caller says so).  `err` gets the reason.
struct ErGpuSceneArrays {        // host arrays of the scene, per original triangle
    const float* vertices;       // [n][3][3]
    const float* normals;        // [n][3][3]
    const float* tangents;       // [n][3][3]
    const float* uvs;            // [n][3][2]
    const float* tangent_sign;   // [n]
    const int32_t* material_id;  // [n]
};
struct ErGpuBvhDevice {
    float4* nodes = nullptr;     // binary tree, nodes_f4 float4 (ownership passes to the caller: hipFree)
    float4* geom = nullptr;      // wide nodes + padding (n8_pieces float4) followed by (n + 1) triangle records
    float4* attr = nullptr;      // attribute records, attr_f4 float4
    size_t nodes_f4 = 0, geom_f4 = 0, attr_f4 = 0, n8_pieces = 0;
    uint32_t nodes8_count = 0, max_depth8 = 0, max_depth2 = 0, leaf_count = 0;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, lift_bound = 0;
    double build_ms = 0;
};
// Test hook (include/eleven_hip_debug.h er_debug_set_gpu_build_failure): 0 = none; 1 = device builds fail as a fault inside the builder
// would; 2 = as a device out-of-memory would.  (Until round 6 an environment variable read on the production path.)
inline std::atomic<int> er_debug_gpu_build_failure{0};
int er_gpu_build_device(const ErGpuSceneArrays& arrays, uint32_t n, int device, ErGpuBvhDevice* out, std::string& err);
