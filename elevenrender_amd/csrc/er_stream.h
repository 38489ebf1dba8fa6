// er_stream.h -- launch wrapper of the CU-resident streaming schedule (er_stream.hip, ER_FLAG_STREAM).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

struct DevScene;
struct WfState;

#ifndef ER_STREAM_SLOTS
#define ER_STREAM_SLOTS 1024u     // slots (pixels in flight) per workgroup; one workgroup of 16 waves per CU.  Measured on C2 with the
                                  // top of the tree in LDS: 768 -> 1060, 1024 -> 1330, 1536 -> 1311, 2048 -> 1286 Msamples/s (the LDS the
                                  // rings of more slots take is worth more as tree levels)
#endif

#ifndef ER_STREAM_SUPER_TILE_DEFAULT
#define ER_STREAM_SUPER_TILE_DEFAULT 8u   // side of the screen regions dealt whole to one XCD, in tiles: the deal that spreads a frame's cost evenly
#endif
#ifndef ER_STREAM_SUPER_TILE_LARGE
#define ER_STREAM_SUPER_TILE_LARGE 16u    // the deal a render moves to after its first call if the frame's cost is even: an XCD's 512 tiles in flight are two compact regions
                                          // instead of eight, +1.6 % on C2, +5 % on C4, +2.5 % on C5 (frames of even cost) -- and -8 ... -18 % on frames whose cost is uneven
                                          // (the soup seen from far away / off to one side: fewer, larger regions per XCD sample the cost too coarsely), so the library
                                          // takes it only if the XCDs' shares of the work COUNTED during the first call (path lengths per tile) are within
                                          // ER_STREAM_COST_SPREAD_MAX of each other under it (er_api.cpp er_stream_adapt; profiles/r04_sweep_super_tile.log, r05_deal_by_counted_work.log)
#endif
#ifndef ER_STREAM_COST_SPREAD_MAX
#define ER_STREAM_COST_SPREAD_MAX 0.04    // (max - min) / mean of the eight XCDs' summed path lengths under the large deal: 0.011 C4, 0.019 Cornell at 1080p, 0.021 C2 (large regions +0.6 ... +4 %); 0.125 / 0.40 on the soup off to one side / from far away (large regions -8 % / -18 %)
#endif
#ifndef ER_STREAM_ADAPT_MIN_MS
#define ER_STREAM_ADAPT_MIN_MS 4.0       // a launch shorter than this (device time) is no reading of the tracer lanes' occupancy: start-up and tail dominate it
#endif
#ifndef ER_STREAM_SMALL_SHARE
#define ER_STREAM_SMALL_SHARE 1152u  // owned pixels per CU up to which a workgroup runs as 12 waves of 168 registers (9 tracers + 3 shaders) instead of 16 of 128
#endif
#ifndef ER_STREAM_TEN_TRACERS_SHARE
#define ER_STREAM_TEN_TRACERS_SHARE 704u   // ... of which 10 trace above this many owned pixels per CU, 9 up to it
#endif
#ifndef ER_STREAM_SPEC_SHARE
#define ER_STREAM_SPEC_SHARE 2304u   // owned pixels per CU up to which the kernel's form with speculative samples is launched (`spec`; the 12-wave form always is)
#endif
#ifndef ER_STREAM_SPEC_MIN_TRIS
#define ER_STREAM_SPEC_MIN_TRIS 1000u   // scenes of fewer triangles never start speculative samples (er_stream.hip ST_DRAWS_MASK; C1: -9 % with them)
#endif
#ifndef ER_STREAM_SPEC_LONG_DEFAULT
#define ER_STREAM_SPEC_LONG_DEFAULT 12  // sixteenths of max_bounces: pixels whose paths are longer than that on average start a speculative successor with every sample
#endif
#ifndef ER_STREAM_KEEP_SHARE
#define ER_STREAM_KEEP_SHARE 6000u   // owned pixels per CU up to which the kernel's form is launched in which a pixel that is behind its workgroup's most advanced one keeps
                                     // its slot (er_stream.hip s_front): C2's 1/2 and 1/3 shares 3 ... 5 % faster, a 200 000-triangle soup at 1280 x 720 (3 600 pixels per
                                     // CU) 4.6 %; the whole C2 frame (8 100) +- 0.5 %: it stays in the plain form, the code of round 5
#endif
#ifndef ER_STREAM_SPEC_KEEP_DEFAULT
#define ER_STREAM_SPEC_KEEP_DEFAULT 2   // 1 + the samples a pixel may be behind its workgroup's most advanced one before it goes on in the slot it has (0 = off)
#endif
#define ER_STREAM_MAX_RING 32768u  // cells of a workgroup's pixel ring at most (one "entry read" bit per cell in LDS): a rank may own
                                  // up to 256 x 32768 = 8.4 M pixels under this schedule (a 4K frame), beyond that er_render_begin takes the wavefront one

// waves: 16 (1024 threads, 128 registers per wave) or 12 (768 threads, 168 registers) per workgroup, tracers of them trace.
// records: slots * er_stream_record_bytes(lights) bytes (slots = blocks * ER_STREAM_SLOTS; lights: the scene uses the point-light
// extension, whose queries take a third line per slot); spill: er_stream_spill_entries(blocks) uint2 entries; ring:
// blocks * ring_cap uint2 entries (the workgroups' pixel rings; ring_cap = a power of two >= 64 * er_stream_deal_tiles(...) and
// <= ER_STREAM_MAX_RING); status: 23 words at an address that is 4 (mod 8): [0] 0 unless a wave's watchdog or a ring guard fired,
// [1..2] iterations of all tracer waves' loops and [3..4] the lanes that held a ray in them, both added up as 64-bit counts by the
// launch (the caller zeroes them before it); [5..6] the earliest start of a workgroup (caller: all ones) and [7 + 2 x ..] the latest end of
// a wave of XCD x = workgroup index % 8 (caller: zero), wall_clock64() ticks.
// S_dev: a device copy of S (the kernel reads the scene descriptor from constant memory, not from its arguments).
void er_launch_stream(const DevScene& S, const DevScene* S_dev, void* records, uint32_t slots, bool lights, void* spill, const uint32_t* deal, uint32_t deal_count, void* ring,
                      uint32_t ring_cap, uint32_t* status, uint32_t n_samples, bool count, uint32_t blocks, uint32_t tracers, uint32_t waves, bool spec, bool keep, hipStream_t stream);
// the deal of the owned tiles to the workgroups (device copy of `out` = `deal` above, deal_count = out.size()); returns the most tiles of one workgroup
// edge: side of a super-tile in 8 x 8 tiles; 0 = ER_STREAM_SUPER_TILE from the environment, else ER_STREAM_SUPER_TILE_DEFAULT
uint32_t er_stream_deal_tiles(const uint32_t* owned, uint32_t count, uint32_t tiles_x, uint32_t blocks, bool xcd_aware, std::vector<uint32_t>& out, uint32_t edge = 0);
uint32_t er_stream_record_bytes(bool lights);
size_t er_stream_spill_entries(uint32_t blocks);
hipError_t er_probe_stream(const char** which);
