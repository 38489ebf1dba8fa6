// er_stream.hip -- CU-resident streaming schedule of the per-sample path for gfx950 (ER_FLAG_STREAM).
//
// Same arithmetic per pixel as renderingKernel (reference src/kernel.cpp:477-646) and as the other schedules of this
// library; the per-slot records are those of the wavefront schedule (er_wavefront.h) and the traversal step and the bounce
// step are the shared ones (er_trav.h, er_bounce.inc).  What changes is, once more, only the schedule:
//
//   * ONE launch per call; one workgroup of 16 waves per CU, resident for the whole call.  The first `tracers` waves only
//     trace, the others only shade (the split is a launch argument);
//   * a workgroup owns a fixed share of the pixels (the owned tiles, dealt round-robin to the workgroups) and ER_STREAM_SLOTS
//     slots.  A slot runs ONE sample of a pixel, puts the pixel back at the tail of the workgroup's pixel ring (HBM; entry =
//     pixel, samples left) and takes the pixel at its head: a pixel's samples are one RNG stream and must run one after
//     the other, but with single samples as the unit all of a workgroup's pixels advance side by side and finish
//     together (with whole pixels as the unit the call ended in a long tail of last pixels).  Slot state lives in HBM
//     (one record per slot, StState below); slots, ring and the planes of the workgroup's pixels are only ever touched
//     by waves of that workgroup -- i.e. of one CU, which share the vector L1 -- so workgroup-scope release/acquire (a wait
//     for the wave's own stores) is all the ordering needed;
//   * shader waves and tracer waves feed each other through three rings in LDS: rays to trace (closest-hit and shadow queries),
//     slots to shade, and slots whose sample is to be FINISHED.  A per-slot word in LDS holds the number of rays of the slot
//     still in flight and what the tracers learnt about them; the tracer that finishes the last one appends the slot to the
//     shade ring -- or, when the path has left the scene and no shadow verdict is ambiguous, straight to the finish ring.
//     A shader wave runs either a SHADING step (the bounce, for a batch of slots whose ray hit something) or a FINISHING step
//     (sky look-up of escaped paths, accumulation into the planes, pixel exchange, next camera ray) for a batch of its own:
//     one path in 4.5 steps ends, and run inline that code kept the whole wave busy for a fifth of its lanes;
//   * wave-level ring work is batched on the tracer side too: a finished ray stays in its lane's registers until the wave's
//     next ring visit (every `refill_min` idle lanes), where publishing results and taking new rays share the cost;
//   * every ring is the bounded queue of er_ring.h: cells carry (lap, full, payload), producers and consumers CHECK the cell
//     they were given and nobody clears anything, so no entry can be lost, duplicated or read from the wrong lap whatever the
//     interleaving -- no timing assumption (the host-thread model of the same functions, tests/native/ring_model.cpp, runs
//     them with tiny capacities under ThreadSanitizer);
//   * there is no launch boundary between bounces and therefore no tail in which a few long rays hold a launch open.
//
// A tracer lane owns a ray from the ring to its last step (state in registers, er_wf_trace's loop).  Three other tracers were
// built and measured in round 3 -- rays as contexts in LDS served by stateless waves through shared node / triangle queues, by
// waves with private lists, and two contexts per lane -- all bit-exact, all slower (profiles/r03_experiment_*).
//
// Every wave leaves its loop when the workgroup's last slot has retired (s_ctl[C_DONE]); a wave that sees no progress for
// ~0.2 s, or whose ring wait outlasts ER_RING_GUARD polls, raises the status word and ends the workgroup (it cannot hang).
#include <algorithm>
#include <cstdlib>
#include <map>
#include <vector>
#include "er_device.h"
#include "er_kernels.h"
#include "er_wavefront.h"
#include "er_trav.h"
#include "er_shade.h"
#include "er_stream.h"
#include "er_ring.h"

using namespace erd;

// Diagnostic build (-DER_TIME_PROBE, tools/shader_sections.py): the shader waves stamp s_memtime at section boundaries and lane 0 adds
// the cycles since the previous stamp to one of ten per-workgroup LDS sums, flushed into the (otherwise meaningless in this build)
// event counters at the end.  Section n = the code between stamp n and the next stamp that executes.
#ifdef ER_TIME_PROBE
#define ER_TPS(n)                                                                                       \
    {                                                                                                    \
        const unsigned long long tp_now = __builtin_amdgcn_s_memtime();                                  \
        if (tp_sec < 10u && (threadIdx.x & 63) == 0) atomicAdd(&s_tp[tp_sec], (unsigned)((tp_now - tp_last) >> 4)); \
        tp_last = tp_now;                                                                                \
        tp_sec = (n);                                                                                    \
    }
#undef ER_TP
#define ER_TP(n) ER_TPS(n)
#else
#define ER_TPS(n) ((void)0)
#endif

// Diagnostic build (-DER_STAGE_PROBE, tools/stage_probe.py): where does a bounce of a path spend its time?  Every closest-hit ray is
// stamped (wall_clock64, 10-ns ticks) when a shader wave queues it (A), when a tracer lane takes it (B), when its traversal ends (C),
// when the tracer publishes it (D), when a shader wave starts the slot's next step (E) and when that step has queued the next ray (F);
// the sums of B-A, C-B, D-C, E-D, F-E and the number of rays come back in the event counters (meaningless otherwise in this build).
#ifdef ER_STAGE_PROBE
#define ER_SP(x) x
__device__ __forceinline__ uint32_t sp_now() { return (uint32_t)wall_clock64(); }
__device__ __forceinline__ void sp_add(unsigned long long* c, uint32_t d) { atomicAdd(c, (unsigned long long)d); }
#else
#define ER_SP(x)
#endif

// Diagnostic build (-DER_TRACER_PROBE, tools/tracer_probe.py): where does an iteration of a tracer wave spend its time?  Every tracer wave
// stamps s_memtime at the boundaries of its loop's parts and adds the cycles since the previous stamp to one of seven wave-uniform sums
// (ring visit: publish + refill / choose / loads issued + wait for the triangle pieces / triangle block / wait for the node pieces + the
// LDS copy / node block / idle polls); the sums, the iterations, the lanes that held a ray and the numbers of publishes and refills come
// back in the event counters (meaningless otherwise in this build).
#ifdef ER_TRACER_PROBE
#define ER_TRP(i)                                                           \
    {                                                                       \
        const unsigned long long trp_now = __builtin_amdgcn_s_memtime();    \
        trp[i] += (uint32_t)(trp_now - trp_last);                           \
        trp_last = trp_now;                                                 \
    }
#else
#define ER_TRP(i) ((void)0)
#endif

namespace {

#define ST_SLOT_BITS 11          // ring payloads: local slot (11 bits) | kind or flag (2 bits) = ER_RING_PAYLOAD_BITS
static_assert(ER_STREAM_SLOTS <= (1u << ST_SLOT_BITS) && ST_SLOT_BITS + 2 <= ER_RING_PAYLOAD_BITS, "a local slot must fit the ring payload");
#define ST_SLOT_MASK ((1u << ST_SLOT_BITS) - 1u)
// ray-ring entry = local slot | kind << 11: 0 closest hit, 1 HDRI shadow query, 2 point-light query;
// shade-ring entry = local slot | fin << 11 (fin: the slot is only finalised, ER_WF_FINALIZE_ONLY);
// finish-ring entry = local slot | esc << 11 (esc: put there by a tracer -- the shading step's part for a ray that left the scene is still to do)
#define ST_FIN 0x100u            // the same flag in s_wait
// flags the TRACERS add to a slot's s_wait word while its rays finish (the tracer that takes the count to zero sees them all):
#define ST_ESC 0x200u            //   the closest-hit ray found no candidate at all: the path has left the scene
#define ST_AMB1 0x400u           //   the HDRI shadow query ended ambiguous (the shader must resolve it by exact distances)
#define ST_AMB2 0x800u           //   ... the point-light query
// ring capacities (log2).  With the checked cells a full ring only makes its producers wait (shader waves for the tracers
// to drain the ray ring -- which they do whatever the shaders are doing -- never the other way round: the shade ring holds
// a slot at most once, so ER_STREAM_SLOTS cells can never be full), so capacities are a tuning matter, not a safety margin.
#define ST_RQ_LOG2 (ER_STREAM_SLOTS > 1024u ? 13u : 12u)
#define ST_SQ_LOG2 (ER_STREAM_SLOTS > 1024u ? 11u : 10u)
static_assert((1u << ST_SQ_LOG2) >= ER_STREAM_SLOTS, "the shade ring must hold every slot once");
// a wave's reservation (<= 3 x 64 entries) must fit the ring several times over (a reservation longer than the ring would wait for
// readers of its own unpublished entries), and the camera rays of all slots go in before the waves start
static_assert((1u << ST_RQ_LOG2) >= ER_STREAM_SLOTS && (1u << ST_RQ_LOG2) >= 4u * 192u, "ray ring too small");
// ... and every ray the slots can have in flight at once -- a closest-hit ray and an HDRI shadow query per slot, a point-light query
// as well with the extension (asserted for that case: one ring size serves both) -- so that the ray ring is never full either and NO
// producer of this kernel ever waits for a reader (the model, tests/native/ring_model.cpp, runs far below this on purpose)
static_assert((1u << ST_RQ_LOG2) >= 3u * ER_STREAM_SLOTS, "the ray ring must hold three rays per slot");
// Static issue priority (s_setprio once, before the loop: arbitration between the waves of a SIMD is by priority, then age).  A
// shading step is ~6 000 vector instructions on a SIMD it shares with two or three tracer waves and was 138 k cycles long
// (profiles/r03_shader_sections_c2.log); with the shader waves at priority 1 five of them fed eleven tracer waves (four feed twelve
// since finished and escaped paths have batches of their own: 1 557 -> 1 627 Msamples/s, profiles/r03_ab_escaped_paths_to_finish_ring.log):
// C2 1 367 -> 1 396, C5 983 -> 1 037, 4K 1 417 -> 1 435, 720p 1 283 -> 1 276 Msamples/s (profiles/r03_ab_wave_priority_*.log; at 10 + 6
// the same priority LOSES 6 %: the tracers then wait for issue slots).  Tracer waves at priority 1 instead: +1 %.
#ifndef ER_STREAM_SPLIT_WAIT
#define ER_STREAM_SPLIT_WAIT 1   // 1: the tracer's triangle block runs while the node pieces are still arriving (er_trav.h trav_fetch_issue / trav_wait_*)
#endif
#ifndef ER_SHADER_PRIO
#define ER_SHADER_PRIO 1         // s_setprio of the shader waves / of the tracer waves for their whole loops (0 = leave it)
#endif
#ifndef ER_TRACER_PRIO
#define ER_TRACER_PRIO 0
#endif
// Waves per CU = a template argument of the kernel (its launch bound decides the register budget): 16 waves of 128 registers -- four per
// SIMD, the shader step spills 111 of them -- or 12 waves of 168 registers (34 spilled), which is faster when a workgroup owns so few
// pixels that the tracer lanes cannot be filled anyway (one GPU's eighth of a 1080p frame: 1.26 instead of 1.34 ms per pass,
// profiles/r04_sweep_sim_world8.log); er_api.cpp chooses by owned pixels per CU.
// north_star: "top BVH levels staged in LDS".  Every workgroup keeps the first wide nodes in LDS (the tree is stored
// breadth-first: 585 = levels 0-3, 47 KB) and the tracer lanes whose node is one of them read it with ds_read_b128 instead of
// five global loads: 9 of a ray's 21 node visits on C2.  Measured on C2 with the first tracer: 0 nodes 1240, 73 -> 1274,
// 256 -> 1286, 585 -> 1330, 800 -> 1331 Msamples/s (profiles/r02_ab_top_levels_in_lds.log).  Round 4: the 16 KB of LDS the
// backend no longer takes for itself (-fno-slp-vectorize) hold 315 nodes of level 4 as well: C2 +0.2 %, C4 +1.5 %
// (profiles/r04_ab_top_nodes_900.log); a deeper LDS stack instead (10 entries): -0.5 %.
#ifndef ER_STREAM_TOP_NODES
#define ER_STREAM_TOP_NODES 900
#endif
#ifndef ST_MAX_TRACERS
#define ST_MAX_TRACERS 13
#endif
enum { C_LIVE = 0, C_DONE, C_INIT, C_WORDS };
#ifndef ST_IDLE_SLEEP
#define ST_IDLE_SLEEP 16         // s_sleep argument (x 64 cycles) of a wave that found nothing to do (4 .. 48 measured: no difference)
#endif
#ifndef ST_BATCH_SLEEP
#define ST_BATCH_SLEEP 8         // ... of a wave waiting for a fuller batch
#endif
#define ST_WATCHDOG 300000u      // idle polls (>= 1000 cycles each) without any ring activity in the workgroup before a wave gives up
#define WF_PENDING_BIT 0x10000u  // per-slot flags in reduc.w, as in er_wavefront.hip: bounce (bits 0-15) | pending HDRI shadow query
#define WF_LPENDING_BIT 0x20000u //   | pending point-light query
// status word bits (er_wait turns any of them into ER_ERR_STATE): 1 tracer watchdog, 2 shader watchdog, 4 pixel ring, 8 shade
// ring, 16 ray ring, 32 context queues -- the last four mean a ring wait outlasted its guard, which no correct run can see
#define ST_ERR_PIXEL 4u
#define ST_ERR_SHADE 8u
#define ST_ERR_RAY 16u
#define ST_ERR_CTX 32u
// the pixel ring's "previous entry has been read" bits (er_ring.h): one per cell, so ring_cap <= ER_STREAM_MAX_RING (er_api.cpp checks)
#define ST_PXBITS_WORDS (ER_STREAM_MAX_RING / 32u)

__device__ __forceinline__ unsigned st_wave_sum(unsigned v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// wave-aggregated reservation on a plain LDS counter: returns this lane's rank-ordered value (valid only if `want`)
__device__ __forceinline__ uint32_t st_reserve(uint32_t* counter, bool want) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0) return 0;
    const unsigned lane = threadIdx.x & 63;
    const unsigned leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);      // (v_readlane: no LDS round trip, unlike __shfl = ds_bpermute)
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// a wave appends the payloads of its `want` lanes to a ring: one reservation, every lane puts its own cell (waiting, in
// theory, for the previous lap's reader), then the entries are published.  All lanes of the wave call.
template <uint32_t LOG2>
__device__ __forceinline__ void st_push(uint32_t* cells, uint32_t* ctl, bool want, uint32_t payload, uint32_t* status, uint32_t err) {
    const unsigned long long m = __ballot(want);
    if (m == 0) return;
    const unsigned lane = threadIdx.x & 63;
    const uint32_t n = (uint32_t)__popcll(m);
    uint32_t base = 0;
    if (lane == 0) base = er_ring_reserve(ctl, n);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
    if (want && !er_ring_put(cells, LOG2, base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), payload)) atomicOr(status, err);
    if (lane == 0) er_ring_publish(ctl, n);      // after the cells: the LDS operations of a wave execute in order
}

// a wave is granted min(want, published entries) positions of a ring (all lanes call; wave-uniform result)
__device__ __forceinline__ uint32_t st_take(uint32_t* ctl, uint32_t want, uint32_t& base, unsigned long long seen) {
    uint32_t granted = 0, hb = 0;
    if ((threadIdx.x & 63) == 0 && want > 0) granted = er_ring_grant(ctl, want, hb, seen);
    base = (uint32_t)__builtin_amdgcn_readlane((int)hb, 0);
    return (uint32_t)__builtin_amdgcn_readlane((int)granted, 0);
}

// The slot records of the wavefront schedule (er_wavefront.h: the same fields with the same meaning), laid out slot by slot.
//   * ONE base pointer: the kernel holds tracer and shader code side by side and both keep many uniform values in scalar
//     registers; twenty plane pointers are forty of them.  A field's address is base + a 32-bit byte offset;
//   * records, not planes: the shade ring delivers slots in the order their rays finish, so a wave's 64 slots are scattered
//     over the workgroup's slots; with one plane per field every 16-byte access pulled in the records of seven other
//     slots (L2 hit rate 47 %, 1.5x the fabric reads of the wavefront schedule); a slot's fields now share its own lines.
//     Line 0 holds everything the tracers read and write, line 1 the rest, line 2 the point-light query.
#define ST_STRIDE_PLAIN 256u
#define ST_STRIDE_LIGHTS 384u
struct StState {
    char* base;
    uint2* spill;
    uint32_t slots, stride;
    // a shadow record index >= slots addresses the point-light query of slot (index - slots), as in the wavefront schedule
    template <class T, uint32_t OFF, uint32_t LOFF>
    __device__ __forceinline__ T& fld2(uint32_t i) const {
        const bool l = i >= slots;
        return *(T*)(base + (size_t)((l ? i - slots : i) * stride + (l ? LOFF : OFF)));
    }
    template <class T, uint32_t OFF>
    __device__ __forceinline__ T& fld(uint32_t i) const { return *(T*)(base + (size_t)(i * stride + OFF)); }
    __device__ __forceinline__ float4& ray_o(uint32_t i) const { return fld<float4, 0>(i); }
    __device__ __forceinline__ float4& ray_d(uint32_t i) const { return fld<float4, 16>(i); }
    __device__ __forceinline__ float4& sh_o(uint32_t i) const { return fld2<float4, 32, 256>(i); }
    __device__ __forceinline__ float4& sh_d(uint32_t i) const { return fld2<float4, 48, 272>(i); }
    __device__ __forceinline__ int& hit(uint32_t i) const { return fld<int, 64>(i); }
    __device__ __forceinline__ int& hit2(uint32_t i) const { return fld<int, 68>(i); }
    __device__ __forceinline__ int& occluded(uint32_t i) const { return fld2<int, 72, 288>(i); }
    __device__ __forceinline__ int& occ_a(uint32_t i) const { return fld2<int, 76, 292>(i); }
    __device__ __forceinline__ int& occ_b(uint32_t i) const { return fld2<int, 80, 296>(i); }
    __device__ __forceinline__ uint32_t& left(uint32_t i) const { return fld<uint32_t, 84>(i); }
    __device__ __forceinline__ uint32_t& pix(uint32_t i) const { return fld<uint32_t, 88>(i); }
    __device__ __forceinline__ uint32_t& stamp(uint32_t i) const { return fld<uint32_t, 92>(i); }      // (ER_STAGE_PROBE only: 10-ns ticks)
    __device__ __forceinline__ uint32_t& stamp0(uint32_t i) const { return fld<uint32_t, 208>(i); }    // (... when the slot's sample began)
    __device__ __forceinline__ float4& light(uint32_t i) const { return fld<float4, 96>(i); }
    __device__ __forceinline__ float4& reduc(uint32_t i) const { return fld<float4, 112>(i); }
    __device__ __forceinline__ float4& aov_n(uint32_t i) const { return fld<float4, 128>(i); }
    __device__ __forceinline__ float4& aov_t(uint32_t i) const { return fld<float4, 144>(i); }
    __device__ __forceinline__ float4& aov_b(uint32_t i) const { return fld<float4, 160>(i); }
    __device__ __forceinline__ float4& c_vis(uint32_t i) const { return fld2<float4, 176, 304>(i); }
    __device__ __forceinline__ float4& c_occ(uint32_t i) const { return fld2<float4, 192, 320>(i); }
    // speculation (SPEC): the verdict word of a speculative slot (SP_*); the local slot + 1 of the speculative sample a slot's sample has started
    // (0: none); the RNG state a speculative sample assumed; the finish-ring payload a parked speculative slot arrived with; 1 if the

    __device__ __forceinline__ uint32_t& spec_word(uint32_t i) const { return fld<uint32_t, 212>(i); }
    __device__ __forceinline__ uint32_t& spec_link(uint32_t i) const { return fld<uint32_t, 216>(i); }
    __device__ __forceinline__ uint32_t& spec_start(uint32_t i) const { return fld<uint32_t, 220>(i); }
    __device__ __forceinline__ uint32_t& spec_entry(uint32_t i) const { return fld<uint32_t, 224>(i); }
    // (SPEC) what the sample in the slot has counted so far -- added to the event counters when it is ACCUMULATED, dropped when a wrong guess starts it
    // over, so that a discarded speculative sample is in no count (the metric's unit is an executed iteration of an accumulated sample):
    // bounce-loop iterations (bits 0-9) | shaded hits (10-19) | HDRI samples (20-29); rays queued; ER_FLAG_COUNTERS: texels, node visits, triangle tests
    __device__ __forceinline__ uint32_t& tally_a(uint32_t i) const { return fld<uint32_t, 232>(i); }
    __device__ __forceinline__ uint32_t& tally_rays(uint32_t i) const { return fld<uint32_t, 236>(i); }
    __device__ __forceinline__ uint32_t& tally_tex(uint32_t i) const { return fld<uint32_t, 240>(i); }
    __device__ __forceinline__ uint32_t& tally_nodes(uint32_t i) const { return fld<uint32_t, 244>(i); }
    __device__ __forceinline__ uint32_t& tally_tris(uint32_t i) const { return fld<uint32_t, 248>(i); }

    // the (origin, direction) pair a tracer reads for ray kind 0 / 1 / 2 of slot g: two adjacent 16-byte pieces
    __device__ __forceinline__ const float4* ray_pair(uint32_t g, uint32_t kind) const {
        return (const float4*)(base + (size_t)(g * stride + (kind == 0u ? 0u : (kind == 1u ? 32u : 256u))));
    }
};

// Pixel-ring cells carry the LAP of their position in the top byte of .y (1 .. 128; 0 = never written): a consumer waits for the
// entry of ITS lap.  Whether the previous lap's entry of a cell has been read is one bit per cell in LDS (er_bits_acquire /
// er_bits_release, er_ring.h): a producer that comes round to an unread cell waits for its reader instead of overwriting it.
#define ST_LAP_TAG(pos, cap) (((((pos) / (cap)) & 0x7Fu) + 1u) << 24)
#define ST_LEFT_MASK 0x00FFFFFFu
// Round 6, speculative sample pipelining (the 12-wave form of the kernel = small shares: SPEC).  A pixel's samples are ONE RNG stream
// (src/kernel.cpp:483-485, 645), so sample k + 1 can start only when sample k has drawn its last number -- unless that number of draws is
// known beforehand.  It is not, but it can be guessed: the pixels a small share waits for are those whose paths run their full length
// sample after sample, and a path of h opaque hits draws exactly 5 + 5 h numbers (6 h with the light extension).
// So when a slot starts a sample of a pixel whose last two accumulated samples drew equally many numbers (DevScene::px_draws) and a slot
// of the workgroup is free, the pixel's NEXT sample starts at once in that slot, from the state the current one will leave if it
// draws that many again.  Nothing of a speculative sample reaches the planes before the sample it follows has been accumulated AND has
// left exactly the state the guess assumed (a compare of two 32-bit states: xorshift32 walks one cycle through all non-zero states,
// so equal states mean equal streams from there on); after a wrong guess the pixel's next sample starts from the true state as it
// would have without the guess, and the speculative slot drops what it has.  The image cannot differ, and a dropped sample is in no
// count (StState::tally_*); what is traded is the work of a path traced for nothing against the chain of a pixel's samples, which
// is what a small share's launch lasts.
#define ST_DRAWS_MASK 0x7Fu          // DevScene::px_draws: the guessed draw count of the pixel's samples | confidence in it << ST_CONF_SHIFT (0 .. 7)
#define ST_CONF_SHIFT 8
#define ST_VOTE_SHIFT 11             // ... | votes (0 .. 7) for ...
#define ST_MAJ_SHIFT 16              // ... the draw count most of its samples had | ...
#define ST_LONG_SHIFT 23             // ... the mean length of its paths, iterations x 16 (0 .. 511)
enum { SP_NONE = 0, SP_PENDING = 1, SP_PARKED = 2, SP_VALID = 3, SP_INVALID = 4 };      // StState::spec_word of a speculative slot

// pixel k of workgroup b's share: entry b + (k / 64) * workgroups of the deal (er_stream_deal_tiles below), lane k % 64
// (false: no tile there, or outside the image)
__device__ __forceinline__ bool st_pixel_of(const DevScene& S, const uint32_t* deal, uint32_t deal_count, uint32_t b, uint32_t nb, uint32_t k, uint32_t& px,
                                            uint32_t& py) {
    const uint32_t t = b + (k >> 6) * nb, l = k & 63u;
    if (t >= deal_count) return false;
    const uint32_t tile = deal[t];
    if (tile == 0xFFFFFFFFu) return false;
    px = (tile % S.tiles_x) * ER_TILE + (l & 7u);
    py = (tile / S.tiles_x) * ER_TILE + (l >> 3);
    return px < S.x_res && py < S.y_res;
}

// A pixel travels through the slots and the pixel ring as px | py << 16 (both < 65 536: er_api.cpp checks), so that nobody divides by
// the run-time width to get its coordinates back; its index in the planes is py * x_res + px.
#define ST_PXY(px, py) ((uint32_t)(px) | ((uint32_t)(py) << 16))
__device__ __forceinline__ uint32_t st_pixel_index(const DevScene& S, uint32_t pxy) { return (pxy >> 16) * S.x_res + (pxy & 0xFFFFu); }

// first camera ray of a sample of pixel pxy in slot g (src/kernel.cpp:492-506); the pixel's RNG state comes from its plane
// (spec: the sample starts from `given` -- a guess of the state the pixel's sample in flight will leave -- instead of the pixel's plane; returns the start state)
template <bool SPECB>
__device__ __forceinline__ uint32_t st_begin_sample(const DevScene& S, const StState& W, uint32_t g, uint32_t pxy, uint32_t left, bool spec = false, uint32_t given = 0u) {
    const uint32_t px = pxy & 0xFFFFu, py = pxy >> 16;
    uint32_t rs = spec ? given : S.rng[py * S.x_res + px];
    const uint32_t rs0 = rs;
    float c1 = rng_next(rs), c2 = rng_next(rs), c3 = rng_next(rs), c4 = rng_next(rs), c5 = rng_next(rs);
    const CamTrig trig = {S.cam_cx, S.cam_sx, S.cam_cy, S.cam_sy, S.cam_cz, S.cam_sz};      // (evaluated once on the host: er_api.cpp)
    const Ray ray = camera_ray(S.cam, (int)px, (int)py, S.x_res, S.y_res, c1, c2, c3, c4, c5, S.cam_trig_valid ? &trig : nullptr);
    W.pix(g) = pxy;
    W.ray_o(g) = make_float4(ray.o.x, ray.o.y, ray.o.z, 0.0f);
    W.ray_d(g) = make_float4(ray.d.x, ray.d.y, ray.d.z, -1.0f);
    W.light(g) = make_float4(0.0f, 0.0f, 0.0f, __builtin_bit_cast(float, rs));
    W.reduc(g) = make_float4(1.0f, 1.0f, 1.0f, __builtin_bit_cast(float, 0u));
    W.aov_n(g) = make_float4(0, 0, 0, 0);
    W.aov_t(g) = make_float4(0, 0, 0, 0);
    W.aov_b(g) = make_float4(0, 0, 0, 0);
    W.left(g) = left;
    if (SPECB) {      // (the 16-wave form of the kernel writes none of this: it is the code it was)
        W.spec_word(g) = spec ? (uint32_t)SP_PENDING : (uint32_t)SP_NONE;
        W.spec_link(g) = 0u;
        W.spec_start(g) = given;
        W.tally_a(g) = 0u; W.tally_rays(g) = 1u;      // (the camera ray)
        W.tally_tex(g) = 0u; W.tally_nodes(g) = 0u; W.tally_tris(g) = 0u;
    }
    ER_SP(W.stamp(g) = sp_now(); W.stamp0(g) = W.stamp(g);)
    return rs0;
}

// what a finished traversal leaves in the slot's record (closest: winner + second candidate; shadow: verdict + candidates)
__device__ __forceinline__ void st_write_result(const StState& W, uint32_t rec, bool shadow, bool occluded, bool overflow, int s0, int s1) {
    if (shadow) {
        W.occluded(rec) = occluded ? 1 : (overflow ? 3 : (s0 >= 0 ? 2 : 0));     // a certain occluder ends the query at once
        W.occ_a(rec) = s0;
        W.occ_b(rec) = s1;
    } else {
        W.hit(rec) = s0 >= 0 ? s0 : s1;
        W.hit2(rec) = overflow ? -2 : ((s0 >= 0 && s1 >= 0) ? s1 : -1);
    }
}

}  // namespace

template <bool COUNT, bool EXT, uint32_t ST_THREADS, bool FUSE, int FORM>
__global__ __launch_bounds__(ST_THREADS) void er_stream_kernel(const DevScene __attribute__((address_space(4)))* Sp, StState W, const uint32_t* deal, uint32_t deal_count, uint2* ring_base, uint32_t ring_cap,
                                                          uint32_t* status, uint32_t n_samples, uint32_t tracers, uint32_t refill_min, uint32_t batch_min, uint32_t fin_min) {
    // The scene descriptor lives in constant memory and is read with scalar loads where it is used.  As a by-value kernel argument its
    // ~60 dwords stayed in SGPRs for the whole kernel and the shading step moved 585 scalar spills to and from VGPR lanes (round 4).
    const DevScene& S = *(const DevScene*)Sp;
    // when the launch began and when each XCD's last wave left (status[5..6], status[7 + 2 x ..]; 100 MHz): how evenly the deal spread
    // the frame's COST over the XCDs is something only the run can tell (er_api.cpp er_stream_adapt)
    if (threadIdx.x == 0) atomicMin((unsigned long long*)(status + 5), (unsigned long long)wall_clock64());
    constexpr uint32_t RQ_LOG2 = ST_RQ_LOG2, SLOTS = ER_STREAM_SLOTS, TOP_NODES = ER_STREAM_TOP_NODES;
    // The kernel's FORM is a template argument: 0 -- the instances that render whole frames, the code they were in round 5; 1 -- with the rule that
    // a pixel which is behind keeps its slot (s_front), for shares of a few pixels per slot; 2 -- that and speculative sample pipelining (comment at
    // ST_DRAWS_MASK), for shares in which slots fall free.  er_launch_stream picks by owned pixels per CU (er_api.cpp).
    constexpr bool SPEC = FORM == 2;
    constexpr bool KEEP = FORM >= 1;
    bool spec_on = false;
    uint32_t spec_need = 0;      // the confidence (1 .. 7) a pixel's guess needs for a speculative start; 0 = no speculation
    uint32_t spec_slack = 0;     // ... and the polls a shader wave must have waited for work before a step in which it may start speculative samples
    uint32_t spec_long = 0;      // ... and the mean path length (iterations x 16) from which a pixel's samples ALWAYS start a speculative successor (0 = never)
    uint32_t spec_keep = 0;      // ... and 1 + the samples a pixel may be behind the workgroup's most advanced one before it goes on in the slot it has instead of queueing (0 = never)
    if (SPEC) { spec_need = (fin_min >> 8) & 7u; spec_slack = (fin_min >> 12) & 0xFFu; spec_long = (fin_min >> 20) & 0x1FFu; spec_keep = fin_min >> 29; spec_on = spec_need != 0u && S.px_draws != nullptr; fin_min &= 0xFFu; }
    else if (KEEP) { spec_keep = fin_min >> 29; fin_min &= 0xFFu; }
    __shared__ uint32_t s_free[SPEC ? (1u << ST_SQ_LOG2) : 1u];      // ring of free slots (SPEC): slots whose pixel ring ran dry, and the slots beyond the share's pixels
    __shared__ __attribute__((aligned(8))) uint32_t s_free_ctl[ER_RING_WORDS];
    __shared__ uint32_t s_rq[1u << RQ_LOG2];
    __shared__ uint32_t s_sq[1u << ST_SQ_LOG2];
    __shared__ uint32_t s_fq[1u << ST_SQ_LOG2];      // finish ring: slots whose path is over and whose sample waits to be accumulated
    __shared__ uint32_t s_wait[SLOTS];
    __shared__ uint32_t s_pxbits[ST_PXBITS_WORDS];
    __shared__ __attribute__((aligned(8))) uint32_t s_rq_ctl[ER_RING_WORDS], s_sq_ctl[ER_RING_WORDS], s_px_ctl[ER_RING_WORDS], s_fq_ctl[ER_RING_WORDS];
    __shared__ uint32_t s_ctl[C_WORDS];
    __shared__ uint32_t s_front;         // the fewest samples any of the workgroup's pixels has left: the most advanced one
    __shared__ uint32_t s_spec[SPEC ? 3 : 1];       // (SPEC) speculative samples started / guesses right / wrong: counted in LDS (no registers), every wave flushes what it finds when it leaves
    __shared__ float4 s_top[TOP_NODES * 5];          // (the LDS copy stays compact -- five pieces per node -- whatever the stride in device memory)
#ifdef ER_TIME_PROBE
    __shared__ uint32_t s_tp[13];      // [1..9] cycles / 16 per section of the shader loop, [10] shader steps, [11] slots shaded, [12] cycles / 16 of tracer iterations
    if (threadIdx.x < 13) s_tp[threadIdx.x] = 0;
#endif
    for (uint32_t i = threadIdx.x; i < TOP_NODES * 5u; i += ST_THREADS)
        s_top[i] = i / 5u < S.node8_count ? S.nodes8[(i / 5u) * ER_NODE8_PIECES + i % 5u] : make_float4(0, 0, 0, 0);
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t g0 = blockIdx.x * SLOTS;          // this workgroup's first slot
    volatile er_lds_u32* v_ctl = (volatile er_lds_u32*)s_ctl;      // (explicitly LDS: see er_ring.h)
    const unsigned long long below = (1ull << lane) - 1ull;

    // ---- start: empty rings (every cell = lap 0, empty), then every slot takes a pixel and queues its first camera ray ----
    for (uint32_t i = threadIdx.x; i < (1u << RQ_LOG2); i += ST_THREADS) s_rq[i] = 0;
    for (uint32_t i = threadIdx.x; i < (1u << ST_SQ_LOG2); i += ST_THREADS) { s_sq[i] = 0; s_fq[i] = 0; }
    for (uint32_t i = threadIdx.x; i < ST_PXBITS_WORDS; i += ST_THREADS) s_pxbits[i] = 0;
    if (SPEC) for (uint32_t i = threadIdx.x; i < (1u << ST_SQ_LOG2); i += ST_THREADS) s_free[i] = 0;
    if (threadIdx.x < ER_RING_WORDS) { s_rq_ctl[threadIdx.x] = 0; s_sq_ctl[threadIdx.x] = 0; s_px_ctl[threadIdx.x] = 0; s_fq_ctl[threadIdx.x] = 0; }
    if (SPEC && threadIdx.x < ER_RING_WORDS) s_free_ctl[threadIdx.x] = 0;
    if (SPEC && threadIdx.x < 3u) s_spec[threadIdx.x] = 0;
    if (KEEP && threadIdx.x == 0) s_front = 0xFFFFFFFFu;
    if (threadIdx.x < C_WORDS) s_ctl[threadIdx.x] = 0;
    __syncthreads();
    // this workgroup's pixels in the order of its tiles: the first SLOTS valid ones start in the slots, the others
    // wait in the pixel ring (entry = pixel, samples left | lap tag; tag 0 marks a cell never written)
    uint2* ring = ring_base + (size_t)blockIdx.x * ring_cap;
    for (uint32_t s = threadIdx.x; s < SLOTS; s += ST_THREADS) s_wait[s] = 0;
    for (uint32_t k = threadIdx.x; k < ring_cap; k += ST_THREADS) ring[k] = make_uint2(0u, 0u);
    __syncthreads();
    for (uint32_t k0 = 0; k0 < ring_cap; k0 += ST_THREADS) {
        const uint32_t k = k0 + threadIdx.x;
        uint32_t px = 0, py = 0;
        const bool valid = n_samples > 0 && k < ring_cap && st_pixel_of(S, deal, deal_count, blockIdx.x, gridDim.x, k, px, py);
        const uint32_t v = st_reserve(&s_ctl[C_INIT], valid);      // (rank among the valid pixels; order does not matter)
        const bool to_slot = valid && v < SLOTS;
        const uint32_t idx = ST_PXY(px, py);
        if (to_slot) {
            st_begin_sample<SPEC>(S, W, g0 + v, idx, n_samples);
            s_wait[v] = 1u;
        } else if (valid) {
            const uint32_t pos = v - SLOTS;      // (< ring_cap: lap 0 of a ring nobody reads yet)
            atomicOr(&s_pxbits[pos >> 5], 1u << (pos & 31u));
            ring[pos] = make_uint2(idx, n_samples | ST_LAP_TAG(pos, ring_cap));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const unsigned long long m = __ballot(to_slot);
        if (m) {
            if (lane == (int)(__ffsll((long long)m) - 1)) atomicAdd(&s_ctl[C_LIVE], (uint32_t)__popcll(m));
            st_push<RQ_LOG2>(s_rq, s_rq_ctl, to_slot, v, status, ST_ERR_RAY);      // kind 0: the camera ray
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t nv = s_ctl[C_INIT];
        const uint32_t in_ring = nv > SLOTS ? nv - SLOTS : 0u;
        s_px_ctl[ER_RING_HEAD] = 0;
        s_px_ctl[ER_RING_TAIL] = in_ring;
        s_px_ctl[ER_RING_COUNT] = in_ring;
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_ctl[C_LIVE] == 0) s_ctl[C_DONE] = 1;
    if (SPEC && spec_on) {      // the slots no pixel started in are free from the beginning
        const uint32_t nv = s_ctl[C_INIT], started = nv < SLOTS ? nv : SLOTS;
        for (uint32_t s0 = 0; s0 < SLOTS; s0 += ST_THREADS) {
            const uint32_t sl = s0 + threadIdx.x;
            st_push<ST_SQ_LOG2>(s_free, s_free_ctl, sl < SLOTS && sl >= started && n_samples > 0, sl, status, ST_ERR_SHADE);
        }
    }

    unsigned c_rays = 0, c_nodes = 0, c_tris = 0;
    unsigned c_paths = 0, c_bounce = 0, c_shaded = 0, c_texels = 0, c_hdri = 0;
    unsigned c_wsteps = 0, c_busy = 0, c_nl = 0, c_tl = 0;

    {
        __shared__ uint2 s_stack[ST_MAX_TRACERS * WF_LDS_STACK * 64];
        __syncthreads();
        // from here on the waves run their own loops: NO workgroup barrier below this line
        if (wave < tracers) {
            // =========================== tracer: er_wf_trace's loop, fed from the ray ring ===========================
            uint2* spill = W.spill + (size_t)(blockIdx.x * 16u + wave) * ER_SPILL_PER_WAVE + lane;
            uint2* stack = s_stack + (size_t)wave * (WF_LDS_STACK * 64) + lane;
            Trav T;
            trav_begin(T, f3s(0), f3(0, 0, 1), false, -1, 0.0f);
            bool busy = false, done = false, done_occl = false;      // a ray in traversal / a finished ray whose result is not yet published
#if ER_TRACER_PRIO
            __builtin_amdgcn_s_setprio(ER_TRACER_PRIO);
#endif
            unsigned ray_n0 = 0, ray_t0 = 0;      // (SPEC && COUNT) the lane's visit / test counts when it took its ray
            uint32_t lsk = 0;      // the ray in hand: its ray-ring payload, local slot | kind << 11 (ONE register across the traversal; the
                                   // record index g0 + slot (+ W.slots for a point-light query) is recomputed where it is needed)
            uint32_t idle = 0, progress = 0;
            uint32_t a_iter = 0, a_busy = 0;      // iterations of this wave's loop and the lanes that held a ray in them (-> status[1..4])
            ER_SP(uint32_t spA = 0; uint32_t spB = 0; uint32_t spC = 0; uint32_t sAB = 0; uint32_t sBC = 0; uint32_t sCD = 0; uint32_t sN = 0;)      // (per lane, flushed once at the end)
#ifdef ER_TRACER_PROBE
            uint32_t trp[7] = {0, 0, 0, 0, 0, 0, 0}, trp_pub = 0, trp_take = 0;
            unsigned long long trp_last = __builtin_amdgcn_s_memtime();
#endif
            while (true) {
                ER_MARK("tracer_loop_top");
                // Every refill_min idle lanes the wave does its ring work in one go: FIRST the finished rays of the idle lanes are
                // published (results out, then the slot's in-flight count; the tracer that takes it to zero hands the slot to the
                // shaders), THEN the idle lanes take rays from the ring.  Between two such visits a finished lane just sits idle with
                // its result in registers: publishing per iteration cost the whole wave ~200 instructions on 86 % of its iterations
                // (some lane of 64 nearly always finishes), for the two or three lanes concerned.
                const unsigned long long bm0 = __ballot(busy);
                unsigned long long rq_peek = 0;
                const bool visit = 64u - (unsigned)__popcll(bm0) >= refill_min || bm0 == 0;
                if (visit && __ballot(done)) {
                    ER_MARK("tracer_publish");
#ifdef ER_TRACER_PROBE
                    trp_pub++;
#endif
                    const uint32_t ls = lsk & ST_SLOT_MASK, kind = lsk >> ST_SLOT_BITS;
                    if (done) st_write_result(W, g0 + ls + (kind == 2u ? W.slots : 0u), T.shadow, done_occl, T.overflow, T.s0, T.s1);
                    if (SPEC && COUNT && done) {      // this ray's visits and tests belong to its sample (counted if and when that is accumulated)
                        atomicAdd(&W.tally_nodes(g0 + ls), c_nodes - ray_n0);
                        atomicAdd(&W.tally_tris(g0 + ls), c_tris - ray_t0);
                        c_nodes = ray_n0; c_tris = ray_t0;
                    }
                    ER_SP(if (done && kind == 0u) { const uint32_t spD = sp_now(); sAB += spB - spA; sBC += spC - spB; sCD += spD - spC; sN++; W.stamp(g0 + ls) = spD; })
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    bool last = false;
                    uint32_t fin = 0;
                    if (done) {
                        const bool esc = !T.shadow && !T.overflow && T.s0 < 0 && T.s1 < 0;
                        const bool amb = T.shadow && !done_occl && (T.overflow || T.s0 >= 0);       // (st_write_result's verdicts 2 and 3)
                        const uint32_t add = (esc ? ST_ESC : 0u) + (amb ? (kind == 1u ? ST_AMB1 : ST_AMB2) : 0u) - 1u;
                        fin = atomicAdd(&s_wait[ls], add) + add;
                        last = (fin & 0xFFu) == 0u;
                    }
                    // A slot whose path has just left the scene, with every pending shadow verdict certain, has nothing for a shading step
                    // to do but look up the sky: it goes straight to the finish ring (flagged), which does that for full batches of such
                    // slots.  Everything else goes to the shade ring.
                    const bool escaped = last && (fin & ST_ESC) != 0u && (fin & (ST_AMB1 | ST_AMB2)) == 0u;
                    st_push<ST_SQ_LOG2>(s_sq, s_sq_ctl, last && !escaped, ls | ((fin & ST_FIN) ? (1u << ST_SLOT_BITS) : 0u), status, ST_ERR_SHADE);
                    st_push<ST_SQ_LOG2>(s_fq, s_fq_ctl, escaped, ls | (1u << ST_SLOT_BITS), status, ST_ERR_SHADE);
                    done = false;
                    ER_MARK("tracer_publish_end");
                }
                if (visit && er_ring_peek_count(rq_peek = er_ring_peek(s_rq_ctl)) > 0) {
                    uint32_t hb = 0;
                    const uint32_t granted = st_take(s_rq_ctl, 64u - (uint32_t)__popcll(bm0), hb, rq_peek);
                    const bool take = !busy && (uint32_t)__popcll(~bm0 & below) < granted;
#ifdef ER_TRACER_PROBE
                    if (granted > 0) trp_take++;
#endif
                    if (granted > 0) {
                        uint32_t e = 0;
                        bool got = false;
                        if (take) {
                            got = er_ring_get(s_rq, RQ_LOG2, hb + (uint32_t)__popcll(~bm0 & below), e);
                            if (!got) atomicOr(status, ST_ERR_RAY);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        if (got) {
                            lsk = e;
                            const uint32_t kind = e >> ST_SLOT_BITS;
                            const uint32_t rec = g0 + (e & ST_SLOT_MASK) + (kind == 2u ? W.slots : 0u);
                            const bool shadow = kind != 0u;
                            const float4 ro = shadow ? W.sh_o(rec) : W.ray_o(rec);
                            const float4 rd = shadow ? W.sh_d(rec) : W.ray_d(rec);
                            trav_begin(T, f3(ro.x, ro.y, ro.z), f3(rd.x, rd.y, rd.z), shadow, shadow ? __builtin_bit_cast(int, ro.w) : -1,
                                       shadow ? rd.w : __builtin_inff());
                            if (!SPEC) c_rays++;      // (SPEC: counted where the ray is queued, into its sample's tally)
                            if (SPEC && COUNT) { ray_n0 = c_nodes; ray_t0 = c_tris; }
                            busy = true;
                            ER_SP(if (kind == 0u) { spA = W.stamp(rec); spB = sp_now(); })
                        }
                    }
                }
                ER_MARK("tracer_refill_end");
                const unsigned long long bm = __ballot(busy);
                if (bm == 0) {
                    if (v_ctl[C_DONE]) break;
                    __builtin_amdgcn_s_sleep(ST_IDLE_SLEEP);
                    ER_TRP(6);
                    const uint32_t pr = er_ring_load(&s_rq_ctl[ER_RING_TAIL]) + er_ring_load(&s_sq_ctl[ER_RING_TAIL]) + er_ring_load(&s_fq_ctl[ER_RING_TAIL]);
                    if (pr != progress) { progress = pr; idle = 0; }
                    if (++idle > ST_WATCHDOG) {
                        if (lane == 0) { atomicOr(status, 1u); s_ctl[C_DONE] = 1; }
                        break;
                    }
                    continue;
                }
                idle = 0;
                a_iter++;                                   // (wave-uniform: two scalar operations per iteration)
                a_busy += (uint32_t)__popcll(bm);
                ER_TRP(0);
#ifdef ER_TIME_PROBE
                const unsigned long long tr0 = __builtin_amdgcn_s_memtime();
#endif
                ER_MARK("tracer_choose");
                bool finished = false, do_step = false;
                TravStep st;
                st.node = false; st.tri = false; st.two = false; st.tslot = 0; st.noff = 0; st.toff = 0;
                if (busy) {
                    if (S.node_count != 0) do_step = trav_choose(T, S, stack, spill, st);
                    finished = !do_step;
                }
                if (COUNT) {
                    c_wsteps++;
                    c_busy += (unsigned)__popcll(bm);
                    c_nl += (unsigned)__popcll(__ballot(st.node));
                    c_tl += (unsigned)__popcll(__ballot(st.tri));
                }
                TravData D;
                ER_MARK("tracer_fetch");
                ER_TRP(1);
                const bool top = st.node && st.noff < (uint32_t)(TOP_NODES * ER_NODE8_PIECES);
#if ER_STREAM_SPLIT_WAIT
                bool occl = false;
                {
                    TravStep sg = st;
                    if (top) sg.node = false;          // (these lanes' node loads go to the shared dummy address)
                    TravRaw R;
                    trav_fetch_issue(S, sg, R);
                    trav_wait_tri(R, D);
                    ER_TRP(2);
                    ER_MARK("tracer_apply");
                    // the triangle block while the node pieces are still on their way
                    const bool do_tri = busy && do_step;
                    TravStep stt = st;
                    stt.tri = st.tri && do_tri;
                    occl = trav_apply_tri<COUNT>(T, S, stt, D, c_tris);
                    ER_TRP(3);
                    trav_wait_node(R, D);
                    if (top) trav_node_from_lds(D, s_top, st.noff);
                    ER_TRP(4);
                    TravStep sn = st;
                    sn.node = st.node && do_tri && !occl;
                    trav_apply_node<COUNT>(T, S, sn, D, c_nodes);
                }
                if (busy) {
                    if (do_step && occl) finished = true;
                    if (finished) { busy = false; done = true; done_occl = occl; ER_SP(spC = sp_now();) }
                }
#else
                {
                    TravStep sg = st;
                    if (top) sg.node = false;          // (these lanes' node loads go to the shared dummy address)
                    trav_fetch(S, sg, D);
                    if (top) trav_node_from_lds(D, s_top, st.noff);
                }
                ER_MARK("tracer_apply");
                bool occl = false;
                if (busy) {
                    if (do_step && trav_apply<COUNT>(T, S, st, D, c_nodes, c_tris)) { occl = true; finished = true; }
                    if (finished) { busy = false; done = true; done_occl = occl; }
                }
#endif
#ifdef ER_TIME_PROBE
                if (lane == 0) atomicAdd(&s_tp[12], (unsigned)((__builtin_amdgcn_s_memtime() - tr0) >> 4));
#endif
                ER_MARK("tracer_iter_end");
                ER_TRP(5);
            }
            ER_MARK("tracer_loop_end");
#ifdef ER_TRACER_PROBE
            const unsigned trp_rays = st_wave_sum(c_rays);
            if (lane == 0) {
                unsigned long long* c = (unsigned long long*)&S.counters->node_visits;      // node_visits ... trace_tri_lanes: nine 64-bit sums
                for (int i = 0; i < 7; i++) atomicAdd(&c[i], (unsigned long long)trp[i]);
                atomicAdd(&c[7], (unsigned long long)trp_pub);
                atomicAdd(&c[8], (unsigned long long)trp_take);
                atomicAdd(&S.counters->paths, (unsigned long long)a_iter);
                atomicAdd(&S.counters->bounce_samples, (unsigned long long)a_busy);
                atomicAdd(&S.counters->rays, (unsigned long long)trp_rays);
            }
#endif
            ER_SP({ unsigned long long* spc = (unsigned long long*)&S.counters->node_visits; sp_add(spc + 0, sAB); sp_add(spc + 1, sBC); sp_add(spc + 2, sCD); sp_add(spc + 5, sN); })
            // how full the tracer lanes were: the host reads it after the call and moves one wave between the two roles for the next
            // call when the tracers starve or the shaders idle (er_api.cpp, er_stream_adapt)
            if (lane == 0) {
                atomicAdd((unsigned long long*)(status + 1), (unsigned long long)a_iter);
                atomicAdd((unsigned long long*)(status + 3), (unsigned long long)a_busy);
            }
        }
    }
    if (wave >= tracers) {
        // =========================== shader: er_wf_shade's step, fed from the shade ring ===========================
        // exact re-trace (rare): its stack in HBM (the tracer waves use the first `tracers` areas of the workgroup, the shader waves the others)
        int* stack = (int*)(W.spill + (size_t)(blockIdx.x * 16u + wave) * ER_SPILL_PER_WAVE) + lane;
        bool have = false;
        uint32_t e = 0;
        uint32_t idle = 0, spins = 0, progress = 0;
        uint32_t slack = 0;
        ER_SP(uint32_t sDE = 0; uint32_t sEF = 0; uint32_t sGE = 0; uint32_t sNG = 0; uint32_t sSS = 0; uint32_t sNS = 0;)
#if ER_SHADER_PRIO
        __builtin_amdgcn_s_setprio(ER_SHADER_PRIO);      // (static priority for the whole loop: issue arbitration is by priority, then age)
#endif
#ifdef ER_TIME_PROBE
        unsigned long long tp_last = __builtin_amdgcn_s_memtime();
        uint32_t tp_sec = 99u;      // (nothing is charged until the first stamp)
#endif
        while (true) {
            ER_MARK("shader_loop_top");
            const unsigned long long sq_peek = er_ring_peek(s_sq_ctl), fq_peek = er_ring_peek(s_fq_ctl);
            const uint32_t avail = er_ring_peek_count(sq_peek), favail = er_ring_peek_count(fq_peek);
            if (avail == 0 && favail == 0) {
                if (v_ctl[C_DONE]) break;
                __builtin_amdgcn_s_sleep(ST_IDLE_SLEEP);
                const uint32_t pr = er_ring_load(&s_rq_ctl[ER_RING_TAIL]) + er_ring_load(&s_sq_ctl[ER_RING_TAIL]) + er_ring_load(&s_fq_ctl[ER_RING_TAIL]);
                if (pr != progress) { progress = pr; idle = 0; }
                if (SPEC) slack++;
                if (++idle > ST_WATCHDOG) {
                    if (lane == 0) { atomicOr(status, 2u); s_ctl[C_DONE] = 1; }
                    break;
                }
                continue;
            }
            idle = 0;
            // Two kinds of step.  A SHADING step runs the bounce for a batch of slots whose rays are traced; a slot whose path ends
            // there is not finished on the spot but handed to the finish ring.  A FINISHING step accumulates the samples of a batch of
            // such slots, puts their pixels back and starts the next samples.  One path in 4.5 steps ends, so finishing inline ran
            // the ~1 400 instructions of accumulate + pixel ring + camera ray in nearly every shading step for a fifth of its lanes;
            // as batches of its own it runs full.  A full finish batch goes first (its slots hold no ray in flight); a partial one
            // runs when the shade ring has too little for a batch.
            // (batch_min carries the patience in its upper bits: polls of ST_BATCH_SLEEP x 64 cycles a wave waits for a full batch before it
            // takes a partial one; a partial finishing batch after a third of that)
            const uint32_t bmin = batch_min & 0xFFu, patience = batch_min >> 8;
            bool fin_mode = favail >= fin_min;
            if (!fin_mode && avail < bmin) {
                if (favail > 0 && (avail == 0 || spins >= patience / 3u)) fin_mode = true;
                else if (spins < patience) {     // a fuller batch costs the same instructions: wait a little for one
                    spins++;
                    if (SPEC) slack++;
                    __builtin_amdgcn_s_sleep(ST_BATCH_SLEEP);
                    continue;
                }
            }
            spins = 0;
            const uint32_t waited = slack;      // (SPEC) polls this wave spent waiting for work since its last step: the shader waves' slack
            slack = 0;
            ER_MARK("shader_take");
            ER_TPS(0);
            uint32_t hb = 0;
            const uint32_t granted = fin_mode ? st_take(s_fq_ctl, 64u, hb, fq_peek) : st_take(s_sq_ctl, 64u, hb, sq_peek);
            if (granted == 0) continue;          // another wave was quicker
            have = (uint32_t)lane < granted;
            if (have && !(fin_mode ? er_ring_get(s_fq, ST_SQ_LOG2, hb + (uint32_t)lane, e) : er_ring_get(s_sq, ST_SQ_LOG2, hb + (uint32_t)lane, e))) { have = false; atomicOr(status, ST_ERR_SHADE); }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            bool push_closest = false, push_shadow = false, push_light = false, retire = false;
            const uint32_t ls = e & ST_SLOT_MASK;
            const uint32_t slot = g0 + ls;
            bool want_pixel = false, to_finish = false;
            uint32_t rs = 0, left_after = 0, done_idx = 0;
            // speculation (SPEC): this lane's sample continues in a speculative slot / wakes a parked speculative slot with this finish-ring entry /
            // has started a speculative sample in local slot ls_spec whose camera ray is to be queued
            bool has_spec = false, wake = false, push_spec = false, keep_own = false;
            uint32_t wake_entry = 0, ls_spec = 0;
            ER_MARK("shader_step");
            ER_TPS(1);
            ER_SP(uint32_t spE = 0;)
            ER_SP(if (have) { spE = sp_now(); if (fin_mode ? (e >> ST_SLOT_BITS) != 0 : (e >> ST_SLOT_BITS) == 0) sDE += spE - W.stamp(slot);
                              else if (fin_mode) { sGE += spE - W.stamp(slot); sNG++; } })
            bool parked = false, discard = false;
            if (SPEC && spec_on && fin_mode && have) {
                // a SPECULATIVE sample reaches its finishing step: nothing of it may reach the planes before the sample it follows has been
                // accumulated and has left the state this one assumed.  Verdict there: commit, or start over from the true state.  Not yet:
                // the slot parks (one compare-and-swap against the committing slot's exchange: exactly one of the two moves it on)
                uint32_t sw = __hip_atomic_load(&W.spec_word(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (sw != (uint32_t)SP_NONE) {
                    if (sw == (uint32_t)SP_PENDING) {
                        W.spec_entry(slot) = e;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        sw = atomicCAS(&W.spec_word(slot), (uint32_t)SP_PENDING, (uint32_t)SP_PARKED);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    parked = sw == (uint32_t)SP_PENDING;
                    discard = sw == (uint32_t)SP_INVALID;
                    if (!parked) W.spec_word(slot) = (uint32_t)SP_NONE;      // (from here on an ordinary sample: committed below, or started over)
                }
            }
            // A speculative sample whose guess was wrong is dropped where it stands -- at its next shading step or at its finishing step -- and its
            // slot falls free: the pixel's sample has been started over, from the true state, by the slot that wrote the verdict (so a wrong
            // guess costs the work of a path, never time in the chain of the pixel's samples)
            // (the verdict word is requested here, with the step's other loads, and looked at when the step is over: a dropped sample's step runs like any
            // other and is thrown away whole)
            uint32_t cancel_w = (uint32_t)SP_NONE;
            if (SPEC && spec_on && have && !fin_mode) cancel_w = __hip_atomic_load(&W.spec_word(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (SPEC && discard) retire = true;
            bool cancel = false;
            // (SPEC) what this step counts goes to the slot's tally, not to the lane's accumulators -- it is counted when the sample is accumulated: inside
            // the step the counters' names mean step-local ones (no value to save and put back across the step, no read of the tally: one atomic add at its end)
            unsigned &o_bounce = c_bounce, &o_shaded = c_shaded, &o_hdri = c_hdri, &o_texels = c_texels, &o_nodes = c_nodes, &o_tris = c_tris;
            if (have && !fin_mode) {
                unsigned t_bounce = 0, t_shaded = 0, t_hdri = 0, t_texels = 0, t_nodes = 0, t_tris = 0;
                unsigned &c_bounce = SPEC ? t_bounce : o_bounce, &c_shaded = SPEC ? t_shaded : o_shaded, &c_hdri = SPEC ? t_hdri : o_hdri;
                unsigned &c_texels = SPEC ? t_texels : o_texels, &c_nodes = SPEC ? t_nodes : o_nodes, &c_tris = SPEC ? t_tris : o_tris;
                (void)c_texels;
                const bool fin_only = (e >> ST_SLOT_BITS) != 0;
                float4 L4 = W.light(slot), R4 = W.reduc(slot);
                F3 light = f3(L4.x, L4.y, L4.z), reduction = f3(R4.x, R4.y, R4.z);
                rs = __builtin_bit_cast(uint32_t, L4.w);
                uint32_t packed = __builtin_bit_cast(uint32_t, R4.w);
                uint32_t bounce = packed & 0xFFFFu;
                if (packed & WF_PENDING_BIT) {   // resolve the previous bounce's shadow query
                    int occ = W.occluded(slot);
                    if (occ >= 2) {
                        float4 so = W.sh_o(slot), sd = W.sh_d(slot);
                        Ray sr;
                        sr.o = f3(so.x, so.y, so.z);
                        sr.d = f3(sd.x, sd.y, sd.z);
                        occ = resolve_shadow<COUNT>(S, stack, sr, __builtin_bit_cast(int, so.w), sd.w, occ, W.occ_a(slot), W.occ_b(slot), c_nodes, c_tris) ? 1 : 0;
                    }
                    float4 c = occ ? W.c_occ(slot) : W.c_vis(slot);
                    light = light + f3(c.x, c.y, c.z);
                }
                if (EXT && (packed & WF_LPENDING_BIT)) {   // ... then its point-light query (second half of the shadow records)
                    const uint32_t q = slot + W.slots;
                    int occ = W.occluded(q);
                    if (occ >= 2) {
                        float4 so = W.sh_o(q), sd = W.sh_d(q);
                        Ray sr;
                        sr.o = f3(so.x, so.y, so.z);
                        sr.d = f3(sd.x, sd.y, sd.z);
                        occ = resolve_shadow<COUNT>(S, stack, sr, __builtin_bit_cast(int, so.w), sd.w, occ, W.occ_a(q), W.occ_b(q), c_nodes, c_tris) ? 1 : 0;
                    }
                    float4 c = occ ? W.c_occ(q) : W.c_vis(q);
                    light = light + f3(c.x, c.y, c.z);
                }
                bool pending = false, lpending = false;
                bool done = fin_only;
                Ray ray;
                ray.o = f3s(0);
                ray.d = f3(0, 0, 1);
                float prev_pdf = -1.0f;
                if (!fin_only) {
                    float4 o = W.ray_o(slot), d = W.ray_d(slot);
                    ray.o = f3(o.x, o.y, o.z);
                    ray.d = f3(d.x, d.y, d.z);
                    int hslot = resolve_closest<COUNT>(S, stack, ray, W.hit(slot), W.hit2(slot), c_nodes, c_tris);
                    c_bounce++;
                    if (EXT) prev_pdf = d.w;
#define ER_BOUNCE_HDRI_QUERY(sr, self_slot, d_self, cv, co)                                                     \
    W.sh_o(slot) = make_float4((sr).o.x, (sr).o.y, (sr).o.z, __builtin_bit_cast(float, (int)(self_slot)));      \
    W.sh_d(slot) = make_float4((sr).d.x, (sr).d.y, (sr).d.z, (d_self));                                          \
    W.c_vis(slot) = make_float4((cv).x, (cv).y, (cv).z, 0.0f);                                                   \
    W.c_occ(slot) = make_float4((co).x, (co).y, (co).z, 0.0f)
#define ER_BOUNCE_LIGHT_QUERY(lr, limit, lv, lo)                                                                \
    {                                                                                                            \
        const uint32_t lq = slot + W.slots;                                                                      \
        const F3 lv_ = (lv), lo_ = (lo);                                                                         \
        W.sh_o(lq) = make_float4((lr).o.x, (lr).o.y, (lr).o.z, __builtin_bit_cast(float, -1));                   \
        W.sh_d(lq) = make_float4((lr).d.x, (lr).d.y, (lr).d.z, (limit));                                         \
        W.c_vis(lq) = make_float4(lv_.x, lv_.y, lv_.z, 0.0f);                                                    \
        W.c_occ(lq) = make_float4(lo_.x, lo_.y, lo_.z, 0.0f);                                                    \
    }
#define ER_BOUNCE_FIRST_HIT(n, t, b)                                                                            \
    W.aov_n(slot) = make_float4((n).x, (n).y, (n).z, 0.0f);                                                      \
    W.aov_t(slot) = make_float4((t).x, (t).y, (t).z, 0.0f);                                                      \
    W.aov_b(slot) = make_float4((b).x, (b).y, (b).z, 0.0f)
#define ER_BOUNCE_FUSE FUSE
#include "er_bounce.inc"
#undef ER_BOUNCE_HDRI_QUERY
#undef ER_BOUNCE_LIGHT_QUERY
#undef ER_BOUNCE_FIRST_HIT
                }
                bool alive = true, fin_next = false;
                if (done && (pending || lpending)) {
                    fin_next = true;          // the path is over but a shadow query is in flight: come back once, without a ray
                } else if (done) {
                    // the path is over and nothing is pending: its sample goes to the finish ring (light and RNG state through the record)
                    W.light(slot) = make_float4(light.x, light.y, light.z, __builtin_bit_cast(float, rs));
                    alive = false;
                    to_finish = true;
                    ER_SP(W.stamp(slot) = sp_now();)      // G: handed to the finish ring by a shading step
                } else {
                    push_closest = true;
                }
                push_shadow = pending;
                push_light = EXT && lpending;
                if (alive) {
                    if (!fin_next) {
                        W.ray_o(slot) = make_float4(ray.o.x, ray.o.y, ray.o.z, 0.0f);
                        W.ray_d(slot) = make_float4(ray.d.x, ray.d.y, ray.d.z, EXT ? prev_pdf : -1.0f);
                    }
                    W.light(slot) = make_float4(light.x, light.y, light.z, __builtin_bit_cast(float, rs));
                    W.reduc(slot) = make_float4(reduction.x, reduction.y, reduction.z,
                                                __builtin_bit_cast(float, bounce | (pending ? WF_PENDING_BIT : 0u) | ((EXT && lpending) ? WF_LPENDING_BIT : 0u)));
                    s_wait[ls] = (push_closest ? 1u : 0u) + (push_shadow ? 1u : 0u) + (push_light ? 1u : 0u) + (fin_next ? ST_FIN : 0u);
                }
                if (SPEC) {
                    cancel = cancel_w == (uint32_t)SP_INVALID;
                    if (!cancel) {
                        // (tally_a | tally_rays << 32 in one add; the fields of the low word never carry into the high one: <= 1 000 iterations)
                        const unsigned long long add = (unsigned long long)(t_bounce | (t_shaded << 10) | (t_hdri << 20)) |
                                                       ((unsigned long long)((push_closest ? 1u : 0u) + (push_shadow ? 1u : 0u) + (push_light ? 1u : 0u)) << 32);
                        __hip_atomic_fetch_add((unsigned long long*)&W.tally_a(slot), add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (COUNT) {
                            W.tally_tex(slot) += t_texels;
                            atomicAdd(&W.tally_nodes(slot), t_nodes);      // (the tracers add to these two as well)
                            atomicAdd(&W.tally_tris(slot), t_tris);
                        }
                    } else {
                        push_closest = false; push_shadow = false; push_light = false; to_finish = false;
                        retire = true;
                    }
                }
            }
            if (!fin_mode) {
                if (__ballot(to_finish)) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the record first, then the entry
                    st_push<ST_SQ_LOG2>(s_fq, s_fq_ctl, to_finish, ls, status, ST_ERR_SHADE);
                }
            } else if (have && !parked && !discard) {
                ER_TPS(6);
                const uint32_t pxy = W.pix(slot);
                const uint32_t idx = st_pixel_index(S, pxy);
                const float4 L4 = W.light(slot);
                F3 light = f3(L4.x, L4.y, L4.z);
                rs = __builtin_bit_cast(uint32_t, L4.w);
                const float4 R4 = W.reduc(slot);
                const uint32_t packed = __builtin_bit_cast(uint32_t, R4.w);
                // while the host has not yet decided which deal of tiles the frame gets, the path's length goes to its tile's sum (a count of
                // work, not a time: the decision is the same on every run of the same frame; one fire-and-forget atomic per finished sample)
                if (S.tile_cost) __hip_atomic_fetch_add(&S.tile_cost[(pxy >> 19) * S.tiles_x + ((pxy & 0xFFFFu) >> 3)], (packed & 0xFFFFu) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((e >> ST_SLOT_BITS) != 0) {
                    // straight from the tracers: the path's last ray left the scene.  What the shading step does for such a slot --
                    // the previous bounce's shadow verdicts (certain ones: the tracers checked), then the miss branch of the bounce.
                    F3 reduction = f3(R4.x, R4.y, R4.z);
                    uint32_t bounce = packed & 0xFFFFu;
                    if (packed & WF_PENDING_BIT) {
                        const float4 c = W.occluded(slot) ? W.c_occ(slot) : W.c_vis(slot);
                        light = light + f3(c.x, c.y, c.z);
                    }
                    if (EXT && (packed & WF_LPENDING_BIT)) {
                        const uint32_t q = slot + W.slots;
                        const float4 c = W.occluded(q) ? W.c_occ(q) : W.c_vis(q);
                        light = light + f3(c.x, c.y, c.z);
                    }
                    const float4 d = W.ray_d(slot);
                    Ray ray;
                    ray.o = f3s(0);
                    ray.d = f3(d.x, d.y, d.z);
                    float prev_pdf = EXT ? d.w : -1.0f;
                    const int hslot = -1;
                    bool done = false, pending = false, lpending = false;
                    c_bounce++;
// (the hooks belong to the hit branch, which `hslot = -1` compiles out)
#define ER_BOUNCE_HDRI_QUERY(sr, self_slot, d_self, cv, co) ((void)(sr), (void)(self_slot), (void)(d_self), (void)(cv), (void)(co))
#define ER_BOUNCE_LIGHT_QUERY(lr, limit, lv, lo) ((void)(lr), (void)(limit), (void)(lv), (void)(lo))
#define ER_BOUNCE_FIRST_HIT(n, t, b) ((void)(n), (void)(t), (void)(b))
#include "er_bounce.inc"
#undef ER_BOUNCE_HDRI_QUERY
#undef ER_BOUNCE_LIGHT_QUERY
#undef ER_BOUNCE_FIRST_HIT
                    (void)bounce; (void)done; (void)pending; (void)lpending; (void)prev_pdf;
                }
                // src/kernel.cpp:597-645
                const float4 an = W.aov_n(slot), at = W.aov_t(slot), ab = W.aov_b(slot);
                const uint32_t sa = S.samples[idx];
                const uint32_t sa2 = accumulate_sample(S, idx, sa, light, f3(an.x, an.y, an.z), f3(at.x, at.y, at.z), f3(ab.x, ab.y, ab.z));
                if (sa2 != sa) S.samples[idx] = sa2;
                S.rng[idx] = rs;
                c_paths++;
                if (SPEC) {      // the accumulated sample's own counts (this step's part -- a path that left the scene -- was counted above, directly)
                    const uint32_t ta = W.tally_a(slot);
                    c_bounce += ta & 0x3FFu; c_shaded += (ta >> 10) & 0x3FFu; c_hdri += (ta >> 20) & 0x3FFu;
                    c_rays += W.tally_rays(slot);
                    if (COUNT) { c_texels += W.tally_tex(slot); c_nodes += W.tally_nodes(slot); c_tris += W.tally_tris(slot); }
                }
                if (SPEC && spec_on) {
                    {   // the pixel's next guess: what this sample drew -- 5 for the camera ray, then per iteration that hit: 1 (opacity) + 4 for an opaque
                        // one (HDRI cell, three for the BRDF sample), 5 with the light extension (src/kernel.cpp:492-493, 538-545; er_bounce.inc)
                        const uint32_t ta = W.tally_a(slot);
                        const uint32_t per_opaque = (EXT && (S.ext_flags & ER_FLAG_POINT_LIGHTS) != 0 && S.light_count > 0) ? 5u : 4u;
                        uint32_t nd = 5u + ((ta >> 10) & 0x3FFu) + per_opaque * ((ta >> 20) & 0x3FFu);
                        nd = nd > ST_DRAWS_MASK ? 0u : nd;      // (longer than the field: no guess for this pixel)
                        // a saturating counter per pixel, as a branch predictor keeps one per branch: the guess stays while it is mostly right
                        const uint32_t oldh = S.px_draws[idx];
                        uint32_t cand = oldh & ST_DRAWS_MASK, conf = (oldh >> ST_CONF_SHIFT) & 7u;
                        if (nd != 0u && nd == cand) conf = conf < 7u ? conf + 1u : 7u;
                        else if (conf >= 2u) conf -= 2u;
                        else { cand = nd; conf = nd != 0u ? 1u : 0u; }
                        // ... and for the pixels a small share's launch ends on -- those whose paths are long, ST_LONG_SHIFT -- the count MOST of its
                        // samples drew (a majority vote: + 1 / - 1, replaced at zero) and the mean length of its paths (iterations x 16, an
                        // exponential average over about eight samples)
                        uint32_t maj = (oldh >> ST_MAJ_SHIFT) & ST_DRAWS_MASK, votes = (oldh >> ST_VOTE_SHIFT) & 7u, mean16 = oldh >> ST_LONG_SHIFT;
                        if (nd != 0u && nd == maj) votes = votes < 7u ? votes + 1u : 7u;
                        else if (votes > 0u) votes--;
                        else { maj = nd; votes = nd != 0u ? 1u : 0u; }
                        const uint32_t its16 = ((ta & 0x3FFu) + ((e >> ST_SLOT_BITS) != 0 ? 1u : 0u)) << 4;
                        mean16 = (uint32_t)((int)mean16 + (((int)(its16 > 511u ? 511u : its16) - (int)mean16) >> 3));
                        S.px_draws[idx] = cand | (conf << ST_CONF_SHIFT) | (votes << ST_VOTE_SHIFT) | (maj << ST_MAJ_SHIFT) | (mean16 << ST_LONG_SHIFT);
                    }
                    const uint32_t link = W.spec_link(slot);
                    if (link) {
                        // this sample started a speculative successor: its guess was right iff this sample left exactly the state it assumed
                        const uint32_t f = g0 + link - 1u;
                        const bool ok = W.spec_start(f) == rs;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the planes and the RNG state first, then the verdict
                        const uint32_t old = atomicExch(&W.spec_word(f), ok ? (uint32_t)SP_VALID : (uint32_t)SP_INVALID);
                        atomicAdd(&s_spec[ok ? 1 : 2], 1u);      // (statistics: guesses right / wrong)
                        if (old == (uint32_t)SP_PARKED) {      // it has finished its path and waits: this lane puts it back on the finish ring
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                            wake = true;
                            wake_entry = W.spec_entry(f);
                        }
                        has_spec = ok;      // right: the pixel goes on in that slot, no entry in the pixel ring.  Wrong: the pixel goes back to the ring as after any
                                            // sample (this slot, or the next free one, starts its next sample from the true state); the other slot drops what it has
                    }
                }
                ER_SP(sSS += sp_now() - W.stamp0(slot); sNS++;)
                // the sample is done: the pixel goes back to the ring (below, as a wave) and the slot takes the next one
                left_after = W.left(slot) - 1;
                done_idx = pxy;
                want_pixel = true;
                // (the pixel ring is first in, first out: every pixel waits equally long for its next turn, so the cheap pixels of a share of a few pixels
                // per slot take more turns per millisecond than the expensive ones, finish early, and leave the launch to the chains of the expensive
                // ones.  A pixel that is BEHIND the workgroup's most advanced one goes on in the slot it has: all pixels then advance sample by sample
                // together, and the expensive ones never wait)
                if (KEEP && spec_keep != 0u) {
                    const uint32_t front = atomicMin(&s_front, left_after);
                    keep_own = left_after > 0u && !has_spec && left_after >= front + spec_keep;
                }
            }
            ER_MARK("shader_pixel_ring");
            ER_TPS(7);
            // finished samples: pixel back to the tail of the pixel ring (unless that was its last sample), next pixel from the head
            if (__ballot(want_pixel)) {
                const bool back = want_pixel && left_after > 0 && !has_spec && !(KEEP && keep_own);
                const unsigned long long mb = __ballot(back);
                if (mb) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the pixel's planes and RNG state first, then its entry
                    uint32_t base = 0;
                    if (lane == 0) base = er_ring_reserve(s_px_ctl, (uint32_t)__popcll(mb));
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
                    if (back) {
                        const uint32_t pos = base + (uint32_t)__popcll(mb & below), cell = pos & (ring_cap - 1u);
                        if (!er_bits_acquire(s_pxbits, cell)) atomicOr(status, ST_ERR_PIXEL);      // (waits while the previous lap's entry is unread)
                        ring[cell] = make_uint2(done_idx, left_after | ST_LAP_TAG(pos, ring_cap));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) er_ring_publish(s_px_ctl, (uint32_t)__popcll(mb));     // counted only once written
                }
                // take up to `want` entries (exact count: a slot that has just put its pixel back always finds an entry unless
                // another slot has taken it)
                bool spawn_want = false;
                uint32_t spawn_rs = 0, spawn_pxy = 0, spawn_left = 0, spawn_draws = 0;
                const bool want_take = want_pixel && !(KEEP && keep_own);
                const unsigned long long mw = __ballot(want_take);
                uint32_t hb2 = 0;
                const uint32_t granted2 = st_take(s_px_ctl, (uint32_t)__popcll(mw), hb2, er_ring_peek(s_px_ctl));
                const uint32_t rank = (uint32_t)__popcll(mw & below);
                if (want_take && rank < granted2) {
                    const uint32_t ppos = hb2 + rank, pcell = ppos & (ring_cap - 1u);
                    const unsigned long long* cell = (const unsigned long long*)(ring + pcell);
                    const uint32_t want_tag = ST_LAP_TAG(ppos, ring_cap);
                    // the cell's writer may still be on its way (positions are handed out before they are written): wait for THIS
                    // lap's entry; pixel and count come in one 8-byte load
                    unsigned long long w = 0;
                    uint32_t y = 0, guard = 0;
                    while (true) {
                        w = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        y = (uint32_t)(w >> 32);
                        if ((y & ~ST_LEFT_MASK) == want_tag) break;
                        if (++guard >= ER_RING_GUARD) { y = 0; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    er_bits_release(s_pxbits, pcell);      // read: the next lap's writer may have the cell
                    const uint32_t nidx = (uint32_t)w;
                    y &= ST_LEFT_MASK;
                    if (y != 0) {
                        spawn_rs = st_begin_sample<SPEC>(S, W, slot, nidx, y);
                        if (SPEC && spec_on && y > 1u) {
                            const uint32_t h = S.px_draws[st_pixel_index(S, nidx)];
                            const bool longp = spec_long != 0u && (h >> ST_LONG_SHIFT) >= spec_long;
                            const bool sure = ((h >> ST_CONF_SHIFT) & 7u) >= spec_need;
                            spawn_draws = (longp && !sure) ? ((h >> ST_MAJ_SHIFT) & ST_DRAWS_MASK) : (h & ST_DRAWS_MASK);
                            // (... and only from a shader wave that has been WAITING for work: where the shader waves never wait another sample in
                            // flight adds to their queue, not to the pixel's progress -- C1's 12-triangle box is bound by its three shader waves and
                            // ran 9 % slower with every pixel two samples deep)
                            spawn_want = spawn_draws != 0u && (longp || sure) && waited >= spec_slack;
                        }
                        spawn_pxy = nidx; spawn_left = y - 1u;
                        s_wait[ls] = 1u;
                        push_closest = true;
                    } else {
                        atomicOr(status, ST_ERR_PIXEL);    // (cannot happen: a granted entry was never written)
                        retire = true;
                    }
                } else if (want_take) {
                    retire = true;      // nothing left in the ring: the pixels still unfinished are all in flight in other slots
                }
                if (KEEP && keep_own) {
                    spawn_rs = st_begin_sample<SPEC>(S, W, slot, done_idx, left_after);
                    if (SPEC && spec_on && left_after > 1u) {
                        const uint32_t h = S.px_draws[st_pixel_index(S, done_idx)];
                        const bool longp = spec_long != 0u && (h >> ST_LONG_SHIFT) >= spec_long;
                        const bool sure = ((h >> ST_CONF_SHIFT) & 7u) >= spec_need;
                        spawn_draws = (longp && !sure) ? ((h >> ST_MAJ_SHIFT) & ST_DRAWS_MASK) : (h & ST_DRAWS_MASK);
                        spawn_want = spawn_draws != 0u && (longp || sure) && waited >= spec_slack;
                    }
                    spawn_pxy = done_idx; spawn_left = left_after - 1u;
                    s_wait[ls] = 1u;
                    push_closest = true;
                }
                if (SPEC && spec_on && __ballot(spawn_want)) {
                    // the samples just begun whose pixels' last samples drew equally many numbers: where a slot is free, the pixel's next sample starts
                    // in it at once, from the state this one leaves if it draws as many numbers as the pixel's last one did
                    const unsigned long long msp = __ballot(spawn_want);
                    uint32_t hbf = 0;
                    const uint32_t gf = st_take(s_free_ctl, (uint32_t)__popcll(msp), hbf, er_ring_peek(s_free_ctl));
                    const uint32_t rk = (uint32_t)__popcll(msp & below);
                    uint32_t fslot = 0;
                    bool got_free = false;
                    if (spawn_want && rk < gf) {
                        got_free = er_ring_get(s_free, ST_SQ_LOG2, hbf + rk, fslot);
                        if (!got_free) atomicOr(status, ST_ERR_SHADE);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if (got_free) {
                        uint32_t sp = spawn_rs;
                        for (uint32_t k = 0; k < spawn_draws; k++) { sp ^= sp << 13; sp ^= sp >> 17; sp ^= sp << 5; }      // (rng_next's step, src/kernel.cpp:42-47)
                        st_begin_sample<SPEC>(S, W, g0 + fslot, spawn_pxy, spawn_left, true, sp);
                        W.spec_link(slot) = fslot + 1u;
                        s_wait[fslot] = 1u;
                        push_spec = true;
                        ls_spec = fslot;
                    }
                    const unsigned long long mgot = __ballot(got_free);
                    if (mgot && lane == (int)(__ffsll((long long)mgot) - 1)) {
                        atomicAdd(&s_ctl[C_LIVE], (uint32_t)__popcll(mgot));      // before any slot of this step retires
                        atomicAdd(&s_spec[0], (uint32_t)__popcll(mgot));          // (statistics: speculative samples started)
                    }
                }
            }
            if (SPEC && spec_on && __ballot(wake)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                st_push<ST_SQ_LOG2>(s_fq, s_fq_ctl, wake, wake_entry, status, ST_ERR_SHADE);      // the parked speculative slots whose verdict this step has written
            }
            ER_MARK("shader_publish");
            ER_TPS(8);
            // the slot's records are written: publish its rays
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            {
                const unsigned long long mc = __ballot(push_closest), ms = __ballot(push_shadow), ml = EXT ? __ballot(push_light) : 0ull;
                const unsigned long long mx = SPEC ? __ballot(push_spec) : 0ull;
                const unsigned nc = (unsigned)__popcll(mc), ns = (unsigned)__popcll(ms), nl = (unsigned)__popcll(ml) + (unsigned)__popcll(mx);
                ER_SP(if (push_closest) { const uint32_t spF = sp_now(); sEF += spF - spE; W.stamp(slot) = spF; })
                if (nc + ns + nl) {
                    uint32_t base = 0;
                    if (lane == 0) base = er_ring_reserve(s_rq_ctl, nc + ns + nl);
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
                    bool ok = true;
                    if (push_closest) ok = er_ring_put(s_rq, RQ_LOG2, base + (uint32_t)__popcll(mc & below), ls) && ok;
                    if (push_shadow) ok = er_ring_put(s_rq, RQ_LOG2, base + nc + (uint32_t)__popcll(ms & below), ls | (1u << ST_SLOT_BITS)) && ok;
                    if (EXT && push_light) ok = er_ring_put(s_rq, RQ_LOG2, base + nc + ns + (uint32_t)__popcll(ml & below), ls | (2u << ST_SLOT_BITS)) && ok;
                    if (SPEC && push_spec) ok = er_ring_put(s_rq, RQ_LOG2, base + nc + ns + (uint32_t)__popcll(ml) + (uint32_t)__popcll(mx & below), ls_spec) && ok;
                    if (!ok) atomicOr(status, ST_ERR_RAY);
                    if (lane == 0) er_ring_publish(s_rq_ctl, nc + ns + nl);      // (after the cells: LDS is in order per wave)
                }
                const unsigned long long mr = __ballot(retire);
                if (SPEC && spec_on && mr) st_push<ST_SQ_LOG2>(s_free, s_free_ctl, retire, ls, status, ST_ERR_SHADE);      // a slot without a pixel is free for a speculative sample
                if (mr) {
                    const uint32_t nr = (uint32_t)__popcll(mr);
                    if (lane == 0) {
                        const uint32_t oldl = atomicSub(&s_ctl[C_LIVE], nr);
                        if (oldl == nr) s_ctl[C_DONE] = 1;     // the workgroup's last slot has retired
                    }
                }
            }
            have = false;
            ER_TPS(9);
#ifdef ER_TIME_PROBE
            if (lane == 0) { atomicAdd(&s_tp[10], 1u); atomicAdd(&s_tp[11], granted); }
#endif
        }
        ER_MARK("shader_loop_end");
        ER_SP({ unsigned long long* spc = (unsigned long long*)&S.counters->node_visits; sp_add(spc + 3, sDE); sp_add(spc + 4, sEF); sp_add(spc + 6, sGE); sp_add(spc + 7, sNG); sp_add(spc + 8, sSS); atomicAdd(&S.counters->paths, (unsigned long long)sNS); })
    }
    ER_MARK("epilogue");
#if defined(ER_STAGE_PROBE) || defined(ER_TRACER_PROBE)
    return;
#endif
#ifdef ER_TIME_PROBE
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* c = (unsigned long long*)&S.counters->node_visits;      // node_visits, tri_tests, shaded_hits, texel_fetches, hdri_samples, trace_* x 4
        for (int i = 1; i < 10; i++) atomicAdd(&c[i - 1], (unsigned long long)s_tp[i]);
        atomicAdd(&S.counters->paths, (unsigned long long)s_tp[10]);
        atomicAdd(&S.counters->bounce_samples, (unsigned long long)s_tp[11]);
        atomicAdd(&S.counters->rays, (unsigned long long)s_tp[12]);
    }
    return;
#endif
    if (SPEC) {
        if (lane == 0) {
            const unsigned q0 = atomicExch(&s_spec[0], 0u), q1 = atomicExch(&s_spec[1], 0u), q2 = atomicExch(&s_spec[2], 0u);
            if (q0 | q1 | q2) { atomicAdd(status + 23, q0); atomicAdd(status + 24, q1); atomicAdd(status + 25, q2); }
        }
    }
    unsigned t0 = st_wave_sum(c_paths), t1 = st_wave_sum(c_bounce), t2 = st_wave_sum(c_rays), t3 = st_wave_sum(c_shaded), t4 = st_wave_sum(c_hdri);
    unsigned t5 = 0, t6 = 0, t7 = 0;
    if (COUNT) { t5 = st_wave_sum(c_nodes); t6 = st_wave_sum(c_tris); t7 = st_wave_sum(c_texels); }
    if (lane == 0) {
        atomicMax((unsigned long long*)(status + 7 + 2 * (blockIdx.x & 7u)), (unsigned long long)wall_clock64());
        if (t0) atomicAdd(&S.counters->paths, (unsigned long long)t0);
        if (t1) atomicAdd(&S.counters->bounce_samples, (unsigned long long)t1);
        if (t2) atomicAdd(&S.counters->rays, (unsigned long long)t2);
        if (t3) atomicAdd(&S.counters->shaded_hits, (unsigned long long)t3);
        if (t4) atomicAdd(&S.counters->hdri_samples, (unsigned long long)t4);
        if (COUNT) {
            atomicAdd(&S.counters->node_visits, (unsigned long long)t5);
            atomicAdd(&S.counters->tri_tests, (unsigned long long)t6);
            atomicAdd(&S.counters->texel_fetches, (unsigned long long)t7);
            atomicAdd(&S.counters->trace_wave_steps, (unsigned long long)c_wsteps);
            atomicAdd(&S.counters->trace_busy_lanes, (unsigned long long)c_busy);
            atomicAdd(&S.counters->trace_node_lanes, (unsigned long long)c_nl);
            atomicAdd(&S.counters->trace_tri_lanes, (unsigned long long)c_tl);
        }
    }
}

hipError_t er_probe_stream(const char** which) {
    hipFuncAttributes a;
    *which = "er_stream_kernel";
    hipError_t e = hipFuncGetAttributes(&a, (const void*)er_stream_kernel<false, false, 1024u, false, 0>);
    if (e == hipSuccess) e = hipFuncGetAttributes(&a, (const void*)er_stream_kernel<false, false, 768u, true, 0>);
    return e != hipSuccess ? e : hipFuncGetAttributes(&a, (const void*)er_stream_kernel<false, false, 768u, true, 2>);
}

// Which workgroup renders which tiles.  Workgroups b and b + 8 run on the same XCD and share its 4 MB L2 (observed dispatch
// order, MI355X_MICROARCH.md; used for speed only -- any deal gives the same pixels), so the frame is cut into super-tiles of
// edge x edge tiles (8: 64 x 64 pixels; er_render_begin also makes the deal of 16, er_stream.h), every super-tile goes to ONE XCD (the
// one with the fewest tiles so far), the XCDs are then levelled tile by tile, and inside an XCD the
// tiles are dealt round-robin to its workgroups: the camera rays and first bounces that an L2 serves then come from a few
// compact screen regions instead of from every eighth tile of the whole frame.  out[b + k * blocks] = the k-th tile of
// workgroup b, 0xFFFFFFFF = none; returns the largest number of tiles any workgroup got.
uint32_t er_stream_deal_tiles(const uint32_t* owned, uint32_t count, uint32_t tiles_x, uint32_t blocks, bool xcd_aware, std::vector<uint32_t>& out, uint32_t edge) {
    if (edge == 0u) {
        const char* e = getenv("ER_STREAM_SUPER_TILE");      // (A/B knob)
        const int v = e ? atoi(e) : 0;
        edge = v >= 1 ? (uint32_t)v : ER_STREAM_SUPER_TILE_DEFAULT;
    }
    const uint32_t X = (xcd_aware && blocks % 8u == 0u) ? 8u : 1u, per = blocks / X, S8 = edge;
    const uint32_t super_x = (tiles_x + S8 - 1u) / S8;
    std::map<uint32_t, std::vector<uint32_t>> by_super;      // row-major super-tile order; tiles inside keep their row-major order
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t tx = owned[i] % tiles_x, ty = owned[i] / tiles_x;
        by_super[X == 1u ? 0u : (ty / S8) * super_x + tx / S8].push_back(owned[i]);
    }
    std::vector<std::vector<uint32_t>> seq(X);
    for (auto& kv : by_super) {
        uint32_t best = 0;
        for (uint32_t x = 1; x < X; x++) if (seq[x].size() < seq[best].size()) best = x;
        seq[best].insert(seq[best].end(), kv.second.begin(), kv.second.end());
    }
    // Whole super-tiles leave the XCDs up to one super-tile apart (1.6 % of an XCD's share of a 1080p frame at the default edge, 6 % at
    // 16) and a launch lasts as long as its fullest XCD: level them tile by tile -- the tail of the fullest XCD's last super-tile goes
    // to the emptiest one -- until no two differ by more than a tile (tests/test_abi_cpu.py; within noise on the soup frames,
    // profiles/r04_sweep_super_tile.log: what a larger super-tile loses on a frame of uneven cost is the CONTENT of its XCDs' shares).
    const char* lv = getenv("ER_STREAM_LEVEL_XCDS");      // (A/B knob)
    for (; !(lv && atoi(lv) == 0);) {
        uint32_t hi = 0, lo = 0;
        for (uint32_t x = 1; x < X; x++) {
            if (seq[x].size() > seq[hi].size()) hi = x;
            if (seq[x].size() < seq[lo].size()) lo = x;
        }
        const size_t diff = seq[hi].size() - seq[lo].size();
        if (diff <= 1u) break;
        const size_t n = diff / 2u;
        seq[lo].insert(seq[lo].end(), seq[hi].end() - (ptrdiff_t)n, seq[hi].end());
        seq[hi].resize(seq[hi].size() - n);
    }
    uint32_t maxk = 0;
    for (uint32_t x = 0; x < X; x++) maxk = std::max<uint32_t>(maxk, (uint32_t)((seq[x].size() + per - 1u) / per));
    out.assign((size_t)blocks * maxk, 0xFFFFFFFFu);
    for (uint32_t x = 0; x < X; x++)
        for (size_t sidx = 0; sidx < seq[x].size(); sidx++) {
            const uint32_t j = (uint32_t)(sidx % per), k = (uint32_t)(sidx / per), b = j * X + x;      // b % X == x: the XCD
            out[(size_t)b + (size_t)k * blocks] = seq[x][sidx];
        }
    return maxk;
}

void er_launch_stream(const DevScene& S, const DevScene* S_dev, void* records, uint32_t slots, bool lights, void* spill, const uint32_t* deal, uint32_t deal_count, void* ring,
                      uint32_t ring_cap, uint32_t* status, uint32_t n_samples, bool count, uint32_t blocks, uint32_t tracers, uint32_t waves, bool spec, bool keep, hipStream_t stream) {
    static const uint32_t refill_min = [] {
        const char* e = getenv("ER_STREAM_REFILL_MIN");
        int v = e ? atoi(e) : 12;
        return (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
    }();
    static const uint32_t batch_min = [] {
        const char* e = getenv("ER_STREAM_BATCH_MIN");
        int v = e ? atoi(e) : 64;      // (five shader waves: a step that is not full is capacity lost)
        const char* p = getenv("ER_STREAM_BATCH_SPINS");      // patience, in polls (knob)
        // (round 5, profiles/r05_sweep_batch_patience.log: 24 polls until now.  Waiting longer for fuller batches loses at every size --
        // 96 polls: -2.5 % at a 1/8 share of the C2 frame, -22 % on C1; 400: -18 % / -63 % -- and waiting less costs nothing at the large
        // sizes and wins on small frames: 3 polls +14 % on C1, +-0.3 % on the whole C2 frame and its 1/4 and 1/8 shares)
        int sp = p ? atoi(p) : 3;
        return (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v)) | ((uint32_t)(sp < 0 ? 0 : (sp > 100000 ? 100000 : sp)) << 8);
    }();
    static const uint32_t fin_min = [] {
        const char* e = getenv("ER_STREAM_FIN_MIN");
        int v = e ? atoi(e) : 64;      // (a finishing step runs as soon as it is full)
        return (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
    }();
    // speculative sample pipelining (12-wave form).  ER_STREAM_SPEC: 0 = off, 1 .. 7 = the confidence a pixel's guessed draw count needs (A/B knob)
    static const uint32_t spec_flag = [] {
        const char* e = getenv("ER_STREAM_SPEC");
        const char* b = getenv("ER_STREAM_SPEC_SLACK");      // polls a shader wave must have waited before a step that starts speculative samples (knob; 0 = no condition)
        int v = e ? atoi(e) : 2, bl = b ? atoi(b) : 1;
        return ((uint32_t)(v < 0 ? 0 : (v > 7 ? 7 : v)) << 8) | ((uint32_t)(bl < 0 ? 0 : (bl > 255 ? 255 : bl)) << 12);
    }();
    // ER_STREAM_SPEC_LONG: sixteenths of max_bounces; a pixel whose paths are longer than that on average starts a speculative successor with every sample (0 = off)
    static const uint32_t spec_long16 = [] {
        const char* e = getenv("ER_STREAM_SPEC_LONG");
        int v = e ? atoi(e) : ER_STREAM_SPEC_LONG_DEFAULT;
        return (uint32_t)(v < 0 ? 0 : (v > 16 ? 16 : v));
    }();
    // ER_STREAM_SPEC_KEEP: 1 + the samples a pixel may be behind before it keeps its slot (0 = off, 1 .. 7)
    static const uint32_t spec_keep = [] {
        const char* e = getenv("ER_STREAM_SPEC_KEEP");
        int v = e ? atoi(e) : ER_STREAM_SPEC_KEEP_DEFAULT;
        return (uint32_t)(v < 0 ? 0 : (v > 7 ? 7 : v));
    }();
    if (S.owned_tile_count == 0 || n_samples == 0) return;
    // (a slot's tally keeps its iterations in ten bits; and a scene of a few triangles -- C1's 12-triangle box, 256 pixels per CU -- runs 5-9 % SLOWER two
    // samples deep, right guesses and all: its rays are three traversal steps long and there is nothing to overlap, profiles/r06_ab_speculative_samples.log)
    // (iterations x 16; the field holds up to 511: longer paths, no such pixels.  12-wave form only: 1/16 share of the C2 frame - 5 %, 1/8 - 0 ... 2 %; the 16-wave form's 1/4 share + 0.6 %)
    const uint32_t long_thr = waves == 12u ? S.max_bounces * spec_long16 : 0u;
    const uint32_t spec_now = (S.max_bounces <= 1000u && S.tri_count >= ER_STREAM_SPEC_MIN_TRIS) ? (spec_flag | ((long_thr > 511u ? 0u : long_thr) << 20) | (((waves == 16u && keep) ? spec_keep : 0u) << 29)) : 0u;      // (the 16-wave form, whose pixels outnumber its slots: 1/4 share - 3.7 ... 5 %, 1/6 - 1 %; the 12-wave form's 1/8 ... 1/16 shares +- 0 ... + 1 %)
    // (the kernel's last argument, one word: bits 0-7 the finishing batch's minimum; forms 1 and 2: bits 29-31 the keep rule's slack + 1; form 2: bits 8-10 the
    // confidence a guess needs, 12-19 the idle polls a shader wave must have behind it before it starts speculative samples, 20-28 the long pixels' mean path x 16)
    const uint32_t keep_plain = (keep && waves == 16u) ? (spec_keep << 29) : 0u;
    if (tracers > ST_MAX_TRACERS) tracers = ST_MAX_TRACERS;      // (the LDS traversal stacks are sized for that many; at least 3 shader waves stay)
    waves = waves == 12u ? 12u : 16u;
    if (tracers > waves - 1u) tracers = waves - 1u;      // at least one shader wave
    if (tracers < 1u) tracers = 1u;
    const bool ext = er_ext_active(S);
    // (FUSE: an instance without the fused-texel path for scenes in which no material is fused, er_device.h generate_hit_data)
    const bool fuse = S.fused_any != 0u;
    auto pick = [&](auto with, auto without) { return fuse ? with : without; };
    // instances: counters x extensions x fused textures, in five forms: 16 waves plain / with the keep rule / with that and speculative samples,
    // 12 waves plain / with speculative samples (the plain forms are the code of round 5: whole frames, and the scenes that start no speculative samples)
#define ST_PICK(THREADS, SPECV)                                                                                                                                  \
    (count ? (ext ? pick(er_stream_kernel<true, true, THREADS, true, SPECV>, er_stream_kernel<true, true, THREADS, false, SPECV>)                                 \
                  : pick(er_stream_kernel<true, false, THREADS, true, SPECV>, er_stream_kernel<true, false, THREADS, false, SPECV>))                               \
           : (ext ? pick(er_stream_kernel<false, true, THREADS, true, SPECV>, er_stream_kernel<false, true, THREADS, false, SPECV>)                               \
                  : pick(er_stream_kernel<false, false, THREADS, true, SPECV>, er_stream_kernel<false, false, THREADS, false, SPECV>)))
    auto k16 = ST_PICK(1024u, 0);
    auto k16k = ST_PICK(1024u, 1);
    auto k16s = ST_PICK(1024u, 2);
    auto k12s = ST_PICK(768u, 2);
    auto k12 = ST_PICK(768u, 0);
#undef ST_PICK
    StState st;
    st.base = (char*)records;
    st.spill = (uint2*)spill;
    st.slots = slots;
    st.stride = er_stream_record_bytes(lights);
    const DevScene __attribute__((address_space(4)))* dS = (const DevScene __attribute__((address_space(4)))*)S_dev;
    if (waves == 12u && spec && spec_now) hipLaunchKernelGGL(k12s, dim3(blocks), dim3(768), 0, stream, dS, st, deal, deal_count, (uint2*)ring, ring_cap, status, n_samples, tracers, refill_min, batch_min, fin_min | spec_now);
    else if (waves == 12u) hipLaunchKernelGGL(k12, dim3(blocks), dim3(768), 0, stream, dS, st, deal, deal_count, (uint2*)ring, ring_cap, status, n_samples, tracers, refill_min, batch_min, fin_min);
    else if (spec && spec_now) hipLaunchKernelGGL(k16s, dim3(blocks), dim3(1024), 0, stream, dS, st, deal, deal_count, (uint2*)ring, ring_cap, status, n_samples, tracers, refill_min, batch_min, fin_min | spec_now);
    else if (keep_plain) hipLaunchKernelGGL(k16k, dim3(blocks), dim3(1024), 0, stream, dS, st, deal, deal_count, (uint2*)ring, ring_cap, status, n_samples, tracers, refill_min, batch_min, fin_min | keep_plain);
    else hipLaunchKernelGGL(k16, dim3(blocks), dim3(1024), 0, stream, dS, st, deal, deal_count, (uint2*)ring, ring_cap, status, n_samples, tracers, refill_min, batch_min, fin_min);
}

uint32_t er_stream_record_bytes(bool lights) { return lights ? ST_STRIDE_LIGHTS : ST_STRIDE_PLAIN; }
// uint2 entries of the spill buffer: per workgroup 16 waves x ER_SPILL_PER_WAVE (er_trav.h): a tracer wave's stack levels beyond the LDS ones
// (ER_STACK8 levels of the wide tree), a shader wave's exact re-trace stack (ER_STACK ints per lane, two per entry): 16 KB per wave, 67 MB on 256 CUs
size_t er_stream_spill_entries(uint32_t blocks) { return (size_t)blocks * 16 * ER_SPILL_PER_WAVE; }
