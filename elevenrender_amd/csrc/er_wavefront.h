// er_wavefront.h -- per-slot path state and queues of the wavefront schedule (er_wavefront.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

struct DevScene;

#define ER_WF_FINALIZE_ONLY 0x80000000u   // queue entry flag: no ray this iteration, only finalise the path

// indices into WfState::counts.  Every counter sits on its own 128-byte line: they are hammered by
// device-scope atomics from every wave, and words that share a line serialise on one L2 channel.
#define WF_LINE 32
#define WF_NC 0                   // [2] closest-queue lengths, by parity (WF_NC, WF_NC + WF_LINE)
#define WF_NS (2 * WF_LINE)       // [2] shadow-queue lengths, by parity
#define WF_TS (4 * WF_LINE)       // shade ticket
#define WF_TT (5 * WF_LINE)       // [8] trace tickets, one per XCD range (WF_TT + x * WF_LINE)
#define WF_COUNTS (13 * WF_LINE)
#define WF_PAR(p) ((p) * WF_LINE)

// One slot per owned pixel lane (owned_tile_count * 64).  All records are 16 bytes so a lane moves
// its state with dwordx4 accesses.
struct WfState {
    float4* ray_o;     // closest-hit ray origin (xyz)
    float4* ray_d;     // closest-hit ray direction (xyz), w = brdfpdf of the last opaque bounce (ER_FLAG_MIS; < 0 none)
    int* hit;          // triangle slot of the closest hit, -1 = miss (written by trace)
    int* hit2;         // second surviving candidate (exact metric decides in shade), -1 none, -2 = re-trace exactly
    float4* light;     // xyz = accumulated radiance of the current path, w = bits(RNG state)
    float4* reduc;     // xyz = path throughput, w = bits(bounce | pending-shadow flag)
    uint32_t* left;    // samples still to finish for this pixel in this call (incl. the current one)
    float4* aov_n;     // first-bounce normal / tangent / bitangent (src/kernel.cpp:581-585)
    float4* aov_t;
    float4* aov_b;
    float4* sh_o;      // shadow ray origin (xyz), w = bits(triangle slot the ray leaves)
    float4* sh_d;      // shadow ray direction (xyz), w = distance at which the ray re-hits that triangle (inf if not)
    float4* c_vis;     // contribution to add if the shadow ray is unoccluded
    float4* c_occ;     // ... if it is occluded
    int* occluded;     // written by trace: 0 no, 1 yes, 2 = decide among occ_a/occ_b by exact metric, 3 = re-trace exactly
    int* occ_a;
    int* occ_b;
    uint32_t* q[2];    // closest queues (slot | flags), ping-pong by iteration parity
    uint32_t* qs[2];   // shadow queues (slot)
    uint32_t* counts;  // WF_COUNTS words
    uint2* spill;      // per persistent trace wave: ER_STACK8 x 64 stack entries beyond the LDS levels (er_trav.h)
    int* shade_stack;  // per shade wave: ER_BVH_MAX_DEPTH x 64 ints, the stack of the rare exact re-trace (in LDS until round 5: 16 KB per wave at depth 64)
    uint32_t slots;    // owned_tile_count * 64.  With ER_FLAG_POINT_LIGHTS the shadow records (sh_o, sh_d, c_vis, c_occ,
                       // occluded, occ_a, occ_b) have 2 * slots entries: [slot] = the HDRI query, [slot + slots] = the
                       // point-light query of the same bounce; shadow-queue entries are these indices
    uint32_t pool, pools;   // this state drives the owned tiles t with t % pools == pool (see er_api.cpp: slot pools)
};

void er_launch_wf_begin(const DevScene& S, const WfState& W, uint32_t n_samples, hipStream_t stream);
// ray_log (may be NULL): the launch stores the number of rays it found in its queues there (ER_FLAG_PROFILE)
void er_launch_wf_trace(const DevScene& S, const WfState& W, uint32_t parity, bool count, uint32_t blocks, uint32_t* ray_log, hipStream_t stream);
void er_launch_wf_shade(const DevScene& S, const WfState& W, uint32_t parity, bool count, uint32_t blocks, hipStream_t stream);
