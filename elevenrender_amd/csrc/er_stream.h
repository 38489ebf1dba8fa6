// er_stream.h -- launch wrapper of the CU-resident streaming schedule (er_stream.hip, ER_FLAG_STREAM).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

struct DevScene;
struct WfState;

#ifndef ER_STREAM_SLOTS
#define ER_STREAM_SLOTS 2048u     // slots (pixels in flight) per workgroup; one workgroup of 16 waves per CU
#endif

// W: the wavefront schedule's slot records with blocks * ER_STREAM_SLOTS slots (W.slots; shadow records doubled with the
// point-light extension), W.spill: 16 * ER_BVH_MAX_DEPTH * 64 entries per workgroup.  pix: one word per slot (its pixel),
// ticket: one word (zeroed by the launch), status: one word, 0 unless a wave's watchdog fired.
void er_launch_stream(const DevScene& S, const WfState& W, uint32_t* pix, uint32_t* ticket, uint32_t* status, uint32_t n_samples, bool count,
                      uint32_t blocks, uint32_t tracers, hipStream_t stream);
hipError_t er_probe_stream(const char** which);
