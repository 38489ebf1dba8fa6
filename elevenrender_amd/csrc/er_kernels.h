// er_kernels.h -- host-callable launch wrappers of the gfx950 kernels (er_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

struct DevScene;

void er_launch_setup(const DevScene& S, hipStream_t stream);
// hipFuncGetAttributes on one kernel of every translation unit: hipSuccess iff the loaded library carries code this
// device can run (*which names the first kernel that failed).  Called by er_render_begin before the first launch.
hipError_t er_probe_kernels(const char** which);
hipError_t er_probe_wavefront(const char** which);
hipError_t er_probe_gpu_build(const char** which);
void er_launch_atrous(const float4* src, int src_stride, const float4* normal, int normal_stride, float4* dst, int w, int h, int step, float kc, hipStream_t stream);
void er_launch_plane(const DevScene& S, int pass, float4* dst, hipStream_t stream);      // DevScene::passes -> one contiguous plane
void er_launch_debug_hit(const DevScene& S, const float* o, const float* d, uint32_t n, int32_t* tri, float* pos, float* dist, hipStream_t stream);
void er_launch_render(const DevScene& S, uint32_t n_samples, bool count, hipStream_t stream);
void er_launch_pack(const DevScene& S, const uint32_t* tiles, uint32_t ntiles, int pass, void* dst, hipStream_t stream);
void er_launch_unpack(const DevScene& S, const uint32_t* tiles, uint32_t ntiles, int pass, const void* src, hipStream_t stream);
