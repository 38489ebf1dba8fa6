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
//     with the wavefront schedule's fields (one record per slot, StState below); slots, ring and the planes of the
//     workgroup's pixels are only ever touched
//     by waves of that workgroup -- i.e. of one CU, which share the vector L1 -- so workgroup-scope release/acquire (a wait
//     for the wave's own stores) is all the ordering needed;
//   * the two kinds of waves feed each other through two rings in LDS: rays to trace (closest-hit and shadow queries) and
//     slots to shade.  A per-slot counter in LDS holds the number of rays of the slot still in flight; the tracer that
//     finishes the last one appends the slot to the shade ring.  Producers write their cells and then add to the ring's
//     count of written entries; a consumer wave takes min(wanted, count) entries with one LDS atomic (st_take), so a ring
//     position is only ever held for an entry that exists, and no lane waits for another lane's ray;
//   * there is no launch boundary between bounces and therefore no tail in which a few long rays hold a launch open: a
//     tracer lane that finishes a ray takes the next one from the ring, whatever bounce or sample it belongs to.
//
// Every wave leaves its loop when the workgroup's last slot has retired (s_ctl[C_DONE]); a wave that sees no progress for
// ~0.2 s raises the status word and ends the workgroup (it cannot hang).
#include <cstdlib>
#include "er_device.h"
#include "er_kernels.h"
#include "er_wavefront.h"
#include "er_trav.h"
#include "er_shade.h"
#include "er_stream.h"

using namespace erd;

namespace {

// ring capacities (powers of two, with room to spare).  ray ring: >= 3 rays per slot (2 without the point-light extension);
// shade ring: >= one entry per slot
#define ST_POW2_GE(x) ((x) <= 1024u ? 1024u : (x) <= 2048u ? 2048u : (x) <= 4096u ? 4096u : (x) <= 8192u ? 8192u : 16384u)
#define ST_RQ_CAP_OF(ext) ST_POW2_GE(((ext) ? 3u : 2u) * ER_STREAM_SLOTS + 768u)
#define ST_SQ_CAP ST_POW2_GE(ER_STREAM_SLOTS + 1024u)
#define ST_KIND_SHIFT 13         // ray-ring entry = local slot | kind << 13: 0 closest hit, 1 HDRI shadow query, 2 point-light query
#define ST_FIN 0x100u            // s_wait flag: when its rays are done the slot is only finalised (ER_WF_FINALIZE_ONLY)
#define ST_SQ_FIN 0x10000u       // the same flag in a shade-ring entry
#ifndef ST_THREADS
#define ST_THREADS 1024          // 16 waves per CU: four per SIMD, 128 VGPRs each
#endif
// north_star: "top BVH levels staged in LDS".  Every workgroup keeps the first ER_STREAM_TOP_NODES wide nodes in LDS (the
// tree is stored breadth-first: 585 = levels 0-3, 47 KB) and the tracer lanes whose node is one of them read it with
// ds_read_b128 instead of five global loads: 9 of a ray's 21 node visits on C2.  Measured on C2 (1024 slots): 0 nodes 1240,
// 73 -> 1274, 256 -> 1286, 585 -> 1330, 800 -> 1331 Msamples/s (profiles/r02_ab_top_levels_in_lds.log).  The reads are
// issued AFTER the step's global loads have arrived, straight into the registers those would have filled: a first version
// that fetched them early into registers of their own spilled the tracer loop and ran 2.6x slower.
#ifndef ER_STREAM_TOP_NODES
#define ER_STREAM_TOP_NODES 585
#endif
#define ST_NONE 0xFFFFFFFFu
#define ST_MAX_TRACERS 12
enum { C_RQ_HEAD = 0, C_RQ_TAIL, C_RQ_COUNT, C_SQ_HEAD, C_SQ_TAIL, C_SQ_COUNT, C_LIVE, C_DONE, C_PX_HEAD, C_PX_TAIL, C_PX_COUNT, C_INIT, C_WORDS };
#ifndef ST_IDLE_SLEEP
#define ST_IDLE_SLEEP 16         // s_sleep argument (x 64 cycles) of a wave that found nothing to do (4 .. 48 measured: no difference)
#endif
#ifndef ST_BATCH_SLEEP
#define ST_BATCH_SLEEP 8         // ... of a shader wave waiting for a fuller batch
#endif
#define ST_WATCHDOG 300000u      // idle polls (>= 1000 cycles each) without any ring activity in the workgroup before a wave gives up
#define WF_PENDING_BIT 0x10000u  // per-slot flags in reduc.w, as in er_wavefront.hip: bounce (bits 0-15) | pending HDRI shadow query
#define WF_LPENDING_BIT 0x20000u //   | pending point-light query

__device__ __forceinline__ unsigned st_wave_sum(unsigned v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// wave-aggregated reservation of ring positions (LDS counter): returns this lane's position (valid only if `want`)
__device__ __forceinline__ uint32_t st_reserve(uint32_t* counter, bool want) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0) return 0;
    const unsigned lane = threadIdx.x & 63;
    const unsigned leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// The slot records of the wavefront schedule (er_wavefront.h: the same fields with the same meaning), laid out slot by slot.
//   * ONE base pointer: the kernel holds tracer and shader code side by side and both keep many uniform values in scalar
//     registers; twenty plane pointers are forty of them (a first build reloaded spilled scalars 85 times per iteration of
//     the tracer loop).  A field's address is base + a 32-bit byte offset (scalar-base addressing form);
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
    __device__ __forceinline__ float4& light(uint32_t i) const { return fld<float4, 96>(i); }
    __device__ __forceinline__ float4& reduc(uint32_t i) const { return fld<float4, 112>(i); }
    __device__ __forceinline__ float4& aov_n(uint32_t i) const { return fld<float4, 128>(i); }
    __device__ __forceinline__ float4& aov_t(uint32_t i) const { return fld<float4, 144>(i); }
    __device__ __forceinline__ float4& aov_b(uint32_t i) const { return fld<float4, 160>(i); }
    __device__ __forceinline__ float4& c_vis(uint32_t i) const { return fld2<float4, 176, 304>(i); }
    __device__ __forceinline__ float4& c_occ(uint32_t i) const { return fld2<float4, 192, 320>(i); }
};

// Taking entries from a ring (all lanes of the wave call; wave-uniform result).  `count` = entries that have been WRITTEN
// and not yet handed out; the wave takes min(want, count) of them -- `granted`, at positions base .. base + granted - 1 -- with
// a compare-and-swap, so the count is exact at every instant.  Two earlier protocols, both caught by tests:
//   * idle lanes reserved positions AHEAD of the producers and polled them: a lane whose wave then did not poll for a few
//     hundred microseconds had its cell overwritten after the ring wrapped, the ray was lost and the workgroup never
//     finished (the watchdog's first catch);
//   * subtract-then-restore (count may dip below zero while several waves ask at once): harmless for the LDS rings, but in
//     the pixel ring a slot that had just put its pixel back could be refused during another wave's dip, retire, and -- if
//     every other slot retired too -- leave the pixel in the ring: one pixel of a small frame one sample short, once in
//     ~50 runs.
__device__ __forceinline__ int st_take(uint32_t* count, uint32_t* head, int want, uint32_t& base) {
    int granted = 0;
    uint32_t hb = 0;
    if ((threadIdx.x & 63) == 0 && want > 0) {
        uint32_t seen = *(volatile uint32_t*)count;
        while (true) {
            granted = (int)seen < want ? (int)seen : want;
            if (granted <= 0) { granted = 0; break; }
            const uint32_t old = atomicCAS(count, seen, seen - (uint32_t)granted);
            if (old == seen) break;
            seen = old;
        }
        if (granted) hb = atomicAdd(head, (uint32_t)granted);
    }
    base = __shfl(hb, 0, 64);
    return __shfl(granted, 0, 64);
}

// Pixel-ring cells carry the LAP of their position in the top byte of .y (1 .. 128; 0 = never written): a consumer knows which
// lap it expects, so nobody has to clear a cell after reading it.  (A first version did clear cells, with a plain store that
// nothing ordered against the NEXT lap's producer: on a workgroup with 64 pixels -- a 64-cell ring that turns over in
// microseconds -- the late clear occasionally wiped a fresh entry or let a stale one be read twice, and one pixel of a 64x48
// test frame came out different in one run of many.)
#define ST_LAP_TAG(pos, cap) (((((pos) / (cap)) & 0x7Fu) + 1u) << 24)
#define ST_LEFT_MASK 0x00FFFFFFu

// pixel k of workgroup b's share: tile b + (k / 64) * workgroups of the owned tiles, lane k % 64 (false: outside the image)
__device__ __forceinline__ bool st_pixel_of(const DevScene& S, uint32_t b, uint32_t nb, uint32_t k, uint32_t& px, uint32_t& py) {
    const uint32_t t = b + (k >> 6) * nb, l = k & 63u;
    if (t >= S.owned_tile_count) return false;
    const uint32_t tile = S.owned_tiles[t];
    px = (tile % S.tiles_x) * ER_TILE + (l & 7u);
    py = (tile / S.tiles_x) * ER_TILE + (l >> 3);
    return px < S.x_res && py < S.y_res;
}

// first camera ray of a sample of pixel idx in slot g (src/kernel.cpp:492-506); the pixel's RNG state comes from its plane
__device__ __forceinline__ void st_begin_sample(const DevScene& S, const StState& W, uint32_t g, uint32_t idx, uint32_t left) {
    const uint32_t px = idx % S.x_res, py = idx / S.x_res;
    uint32_t rs = S.rng[idx];
    float c1 = rng_next(rs), c2 = rng_next(rs), c3 = rng_next(rs), c4 = rng_next(rs), c5 = rng_next(rs);
    const Ray ray = camera_ray(S.cam, (int)px, (int)py, S.x_res, S.y_res, c1, c2, c3, c4, c5);
    W.pix(g) = idx;
    W.ray_o(g) = make_float4(ray.o.x, ray.o.y, ray.o.z, 0.0f);
    W.ray_d(g) = make_float4(ray.d.x, ray.d.y, ray.d.z, -1.0f);
    W.light(g) = make_float4(0.0f, 0.0f, 0.0f, __builtin_bit_cast(float, rs));
    W.reduc(g) = make_float4(1.0f, 1.0f, 1.0f, __builtin_bit_cast(float, 0u));
    W.aov_n(g) = make_float4(0, 0, 0, 0);
    W.aov_t(g) = make_float4(0, 0, 0, 0);
    W.aov_b(g) = make_float4(0, 0, 0, 0);
    W.left(g) = left;
}

}  // namespace

template <bool COUNT, bool EXT>
__global__ __launch_bounds__(ST_THREADS) void er_stream_kernel(DevScene S, StState W, uint2* ring_base, uint32_t ring_cap, uint32_t* status,
                                                          uint32_t n_samples, uint32_t tracers, uint32_t refill_min, uint32_t batch_min) {
    __shared__ uint2 s_stack[ST_MAX_TRACERS * WF_LDS_STACK * 64];
    constexpr uint32_t ST_RQ_CAP = ST_RQ_CAP_OF(EXT);
    __shared__ uint32_t s_rq[ST_RQ_CAP];
    __shared__ uint32_t s_sq[ST_SQ_CAP];
    __shared__ uint32_t s_wait[ER_STREAM_SLOTS];
    __shared__ uint32_t s_ctl[C_WORDS];
#if ER_STREAM_TOP_NODES > 0
    __shared__ float4 s_top[ER_STREAM_TOP_NODES * ER_NODE8_PIECES];
    for (uint32_t i = threadIdx.x; i < ER_STREAM_TOP_NODES * ER_NODE8_PIECES; i += ST_THREADS)
        s_top[i] = i < S.node8_count * ER_NODE8_PIECES ? S.nodes8[i] : make_float4(0, 0, 0, 0);
#endif
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t g0 = blockIdx.x * ER_STREAM_SLOTS;          // this workgroup's first slot
    volatile uint32_t* v_rq = s_rq;
    volatile uint32_t* v_sq = s_sq;
    volatile uint32_t* v_ctl = s_ctl;
    const unsigned long long below = (1ull << lane) - 1ull;

    // ---- start: empty rings, then every slot takes a pixel and queues its first camera ray ----
    for (uint32_t i = threadIdx.x; i < ST_RQ_CAP; i += ST_THREADS) s_rq[i] = 0;
    for (uint32_t i = threadIdx.x; i < ST_SQ_CAP; i += ST_THREADS) s_sq[i] = 0;
    if (threadIdx.x < C_WORDS) s_ctl[threadIdx.x] = 0;
    __syncthreads();
    // this workgroup's pixels in the order of its tiles: the first ER_STREAM_SLOTS valid ones start in the slots, the others
    // wait in the pixel ring (entry = pixel, samples left; .y == 0 marks an empty cell)
    uint2* ring = ring_base + (size_t)blockIdx.x * ring_cap;
    for (uint32_t s = threadIdx.x; s < ER_STREAM_SLOTS; s += ST_THREADS) s_wait[s] = 0;
    for (uint32_t k = threadIdx.x; k < ring_cap; k += ST_THREADS) ring[k] = make_uint2(0u, 0u);
    __syncthreads();
    for (uint32_t k0 = 0; k0 < ring_cap; k0 += ST_THREADS) {
        const uint32_t k = k0 + threadIdx.x;
        uint32_t px = 0, py = 0;
        const bool valid = n_samples > 0 && k < ring_cap && st_pixel_of(S, blockIdx.x, gridDim.x, k, px, py);
        const uint32_t v = st_reserve(&s_ctl[C_INIT], valid);      // (rank among the valid pixels; order does not matter)
        const bool to_slot = valid && v < ER_STREAM_SLOTS;
        const uint32_t idx = py * S.x_res + px;
        if (to_slot) {
            st_begin_sample(S, W, g0 + v, idx, n_samples);
            s_wait[v] = 1u;
        } else if (valid) {
            ring[v - ER_STREAM_SLOTS] = make_uint2(idx, n_samples | ST_LAP_TAG(v - ER_STREAM_SLOTS, ring_cap));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const unsigned long long m = __ballot(to_slot);
        if (m) {
            if (lane == (int)(__ffsll((long long)m) - 1)) atomicAdd(&s_ctl[C_LIVE], (uint32_t)__popcll(m));
            const uint32_t pos = st_reserve(&s_ctl[C_RQ_TAIL], to_slot);
            if (to_slot) s_rq[pos & (ST_RQ_CAP - 1u)] = v + 1u;
            if (lane == 0) atomicAdd(&s_ctl[C_RQ_COUNT], (uint32_t)__popcll(m));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t nv = s_ctl[C_INIT];
        const uint32_t in_ring = nv > ER_STREAM_SLOTS ? nv - ER_STREAM_SLOTS : 0u;
        s_ctl[C_PX_HEAD] = 0;
        s_ctl[C_PX_TAIL] = in_ring;
        s_ctl[C_PX_COUNT] = in_ring;
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_ctl[C_LIVE] == 0) s_ctl[C_DONE] = 1;
    __syncthreads();
    // from here on the waves run their own loops: NO workgroup barrier below this line

    unsigned c_rays = 0, c_nodes = 0, c_tris = 0;
    unsigned c_paths = 0, c_bounce = 0, c_shaded = 0, c_texels = 0, c_hdri = 0;
    unsigned c_wsteps = 0, c_busy = 0, c_nl = 0, c_tl = 0;
    uint2* spill = W.spill + (size_t)(blockIdx.x * 16u + wave) * (ER_STACK * 64) + lane;

    if (wave < tracers) {
        // =========================== tracer: er_wf_trace's loop, fed from the ray ring ===========================
        uint2* stack = s_stack + (size_t)wave * (WF_LDS_STACK * 64) + lane;
        Trav T;
        trav_begin(T, f3s(0), f3(0, 0, 1), false, -1, 0.0f);
        bool busy = false;
        uint32_t ls = 0, kind = 0, rec = 0;   // the ray in hand: local slot, kind, index of its records (g, or g + W.slots)
        uint32_t idle = 0, progress = 0;
        while (true) {
            // idle lanes take rays from the ring; skipped while fewer than refill_min lanes are idle (it costs the whole wave ~40
            // instructions and, when rays are taken, a pair of dependent loads) or the ring has nothing written
            const unsigned long long bm0 = __ballot(busy);
            if ((64u - (unsigned)__popcll(bm0) >= refill_min || bm0 == 0) && (int)v_ctl[C_RQ_COUNT] > 0) {
                uint32_t hb = 0;
                const int granted = st_take(&s_ctl[C_RQ_COUNT], &s_ctl[C_RQ_HEAD], 64 - __popcll(bm0), hb);
                const bool take = !busy && __popcll(~bm0 & below) < granted;
                if (granted > 0) {
                    uint32_t v = 0;
                    if (take) {
                        volatile uint32_t* cell = v_rq + ((hb + (uint32_t)__popcll(~bm0 & below)) & (ST_RQ_CAP - 1u));
                        uint32_t guard = 0;
                        while ((v = *cell) == 0 && ++guard < (1u << 22)) __builtin_amdgcn_s_sleep(1);   // (its writer is on its way)
                        *cell = 0;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if (take && v != 0) {
                        const uint32_t e = v - 1u;
                        ls = e & ((1u << ST_KIND_SHIFT) - 1u);
                        kind = e >> ST_KIND_SHIFT;
                        rec = g0 + ls + (kind == 2u ? W.slots : 0u);
                        const bool shadow = kind != 0u;
                        const float4 ro = shadow ? W.sh_o(rec) : W.ray_o(rec);
                        const float4 rd = shadow ? W.sh_d(rec) : W.ray_d(rec);
                        trav_begin(T, f3(ro.x, ro.y, ro.z), f3(rd.x, rd.y, rd.z), shadow, shadow ? __builtin_bit_cast(int, ro.w) : -1,
                                   shadow ? rd.w : __builtin_inff());
                        c_rays++;
                        busy = true;
                    } else if (take) {
                        atomicOr(status, 16u);      // (cannot happen: a granted entry was never written)
                    }
                }
            }
            const unsigned long long bm = __ballot(busy);
            if (bm == 0) {
                if (v_ctl[C_DONE]) break;
                __builtin_amdgcn_s_sleep(ST_IDLE_SLEEP);
                const uint32_t pr = v_ctl[C_RQ_TAIL] + v_ctl[C_SQ_TAIL];
                if (pr != progress) { progress = pr; idle = 0; }
                if (++idle > ST_WATCHDOG) {
                    if (lane == 0) { atomicOr(status, 1u); s_ctl[C_DONE] = 1; }
                    break;
                }
                continue;
            }
            idle = 0;
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
#if ER_STREAM_TOP_NODES > 0
            {
                const bool top = st.node && st.noff < (uint32_t)(ER_STREAM_TOP_NODES * ER_NODE8_PIECES);
                TravStep sg = st;
                if (top) sg.node = false;          // (these lanes' node loads go to the shared dummy address)
                trav_fetch(S, sg, D);
                if (top) { const float4* q = s_top + st.noff; D.n0 = q[0]; D.n1 = q[1]; D.n2 = q[2]; D.n3 = q[3]; D.n4 = q[4]; }
            }
#else
            trav_fetch(S, st, D);
#endif
            if (busy) {
                if (do_step) {
                    if (trav_apply<COUNT>(T, S, st, D, c_nodes, c_tris)) {
                        W.occluded(rec) = 1;   // a certain occluder ends the shadow query
                        finished = true;
                    }
                } else if (T.shadow) {
                    W.occluded(rec) = T.overflow ? 3 : (T.s0 >= 0 ? 2 : 0);
                }
                if (finished) {
                    if (T.shadow) {
                        W.occ_a(rec) = T.s0;
                        W.occ_b(rec) = T.s1;
                    } else {
                        W.hit(rec) = T.s0 >= 0 ? T.s0 : T.s1;
                        W.hit2(rec) = T.overflow ? -2 : ((T.s0 >= 0 && T.s1 >= 0) ? T.s1 : -1);
                    }
                    busy = false;
                }
            }
            // results out, then the slot's in-flight count; the tracer that takes it to zero hands the slot to the shaders
            if (__ballot(finished)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                bool last = false;
                uint32_t old = 0;
                if (finished) {
                    old = atomicSub(&s_wait[ls], 1u);
                    last = (old & 0xFFu) == 1u;
                }
                if (__ballot(last)) {
                    const unsigned long long ml = __ballot(last);
                    const uint32_t pos = st_reserve(&s_ctl[C_SQ_TAIL], last);
                    if (last) s_sq[pos & (ST_SQ_CAP - 1u)] = (ls | ((old & ST_FIN) ? ST_SQ_FIN : 0u)) + 1u;
                    if (lane == 0) atomicAdd(&s_ctl[C_SQ_COUNT], (uint32_t)__popcll(ml));      // (after the cells: LDS is in order per wave)
                }
            }
        }
    } else {
        // =========================== shader: er_wf_shade's step, fed from the shade ring ===========================
        int* stack = (int*)(W.spill + (size_t)(blockIdx.x * 16u + wave) * (ER_STACK * 64)) + lane;   // exact re-trace (rare): HBM
        bool have = false;
        uint32_t e = 0;
        uint32_t idle = 0, spins = 0, progress = 0;
        while (true) {
            const int avail = (int)v_ctl[C_SQ_COUNT];
            if (avail <= 0) {
                if (v_ctl[C_DONE]) break;
                __builtin_amdgcn_s_sleep(ST_IDLE_SLEEP);
                const uint32_t pr = v_ctl[C_RQ_TAIL] + v_ctl[C_SQ_TAIL];
                if (pr != progress) { progress = pr; idle = 0; }
                if (++idle > ST_WATCHDOG) {
                    if (lane == 0) { atomicOr(status, 2u); s_ctl[C_DONE] = 1; }
                    break;
                }
                continue;
            }
            idle = 0;
            if (avail < (int)batch_min && spins < 24u) {     // a fuller batch costs the same instructions: wait a little for one
                spins++;
                __builtin_amdgcn_s_sleep(ST_BATCH_SLEEP);
                continue;
            }
            spins = 0;
            uint32_t hb = 0;
            const int granted = st_take(&s_ctl[C_SQ_COUNT], &s_ctl[C_SQ_HEAD], 64, hb);
            if (granted == 0) continue;          // another wave was quicker
            have = lane < granted;
            if (have) {
                volatile uint32_t* cell = v_sq + ((hb + (uint32_t)lane) & (ST_SQ_CAP - 1u));
                uint32_t v = 0, guard = 0;
                while ((v = *cell) == 0 && ++guard < (1u << 22)) __builtin_amdgcn_s_sleep(1);   // (its writer is on its way)
                *cell = 0;
                e = v - 1u;
                if (v == 0) { have = false; atomicOr(status, 8u); }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            bool push_closest = false, push_shadow = false, push_light = false, retire = false;
            const uint32_t ls = e & 0xFFFFu;
            const uint32_t slot = g0 + ls;
            bool want_pixel = false;
            uint32_t rs = 0, left_after = 0, done_idx = 0;
            if (have) {
                const bool fin_only = (e & ST_SQ_FIN) != 0;
                uint32_t idx = W.pix(slot);
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
#include "er_bounce.inc"
#undef ER_BOUNCE_HDRI_QUERY
#undef ER_BOUNCE_LIGHT_QUERY
#undef ER_BOUNCE_FIRST_HIT
                }
                bool alive = true, fin_next = false;
                if (done && (pending || lpending)) {
                    fin_next = true;          // the path is over but a shadow query is in flight: come back once, without a ray
                } else if (done) {
                    // src/kernel.cpp:597-645
                    float4 an = W.aov_n(slot), at = W.aov_t(slot), ab = W.aov_b(slot);
                    const uint32_t sa = S.samples[idx];
                    const uint32_t sa2 = accumulate_sample(S, idx, sa, light, f3(an.x, an.y, an.z), f3(at.x, at.y, at.z), f3(ab.x, ab.y, ab.z));
                    if (sa2 != sa) S.samples[idx] = sa2;
                    S.rng[idx] = rs;
                    c_paths++;
                    // the sample is done: the pixel goes back to the ring (below, as a wave) and the slot takes the next one
                    left_after = W.left(slot) - 1;
                    done_idx = idx;
                    alive = false;
                    want_pixel = true;
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
            }
            // finished samples: pixel back to the tail of the pixel ring (unless that was its last sample), next pixel from the head
            if (__ballot(want_pixel)) {
                const bool back = want_pixel && left_after > 0;
                const unsigned long long mb = __ballot(back);
                if (mb) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the pixel's planes and RNG state first, then its entry
                    const uint32_t pos = st_reserve(&s_ctl[C_PX_TAIL], back);
                    if (back) ring[pos & (ring_cap - 1u)] = make_uint2(done_idx, left_after | ST_LAP_TAG(pos, ring_cap));
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) atomicAdd(&s_ctl[C_PX_COUNT], (uint32_t)__popcll(mb));     // counted only once written
                }
                // take up to `want` entries (exact count: a slot that has just put its pixel back always finds an entry unless
                // another slot has taken it)
                const unsigned long long mw = __ballot(want_pixel);
                uint32_t hb = 0;
                const int granted = st_take(&s_ctl[C_PX_COUNT], &s_ctl[C_PX_HEAD], __popcll(mw), hb);
                const int rank = __popcll(mw & below);
                if (want_pixel && rank < granted) {
                    const uint32_t ppos = hb + (uint32_t)rank;
                    const unsigned long long* cell = (const unsigned long long*)(ring + (ppos & (ring_cap - 1u)));
                    const uint32_t want_tag = ST_LAP_TAG(ppos, ring_cap);
                    // the cell's writer may still be on its way (positions are handed out before they are written): wait for THIS
                    // lap's entry; pixel and count come in one 8-byte load
                    unsigned long long w = 0;
                    uint32_t y = 0, guard = 0;
                    while (true) {
                        w = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        y = (uint32_t)(w >> 32);
                        if ((y & ~ST_LEFT_MASK) == want_tag) break;
                        if (++guard >= (1u << 22)) { y = 0; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const uint32_t nidx = (uint32_t)w;
                    y &= ST_LEFT_MASK;
                    if (y != 0) {
                        st_begin_sample(S, W, slot, nidx, y);
                        s_wait[ls] = 1u;
                        push_closest = true;
                    } else {
                        atomicOr(status, 4u);    // (cannot happen: a granted entry was never written)
                        retire = true;
                    }
                } else if (want_pixel) {
                    retire = true;      // nothing left in the ring: the pixels still unfinished are all in flight in other slots
                }
            }
            // the slot's records are written: publish its rays
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            {
                const unsigned long long mc = __ballot(push_closest), ms = __ballot(push_shadow), ml = EXT ? __ballot(push_light) : 0ull;
                const unsigned nc = (unsigned)__popcll(mc), ns = (unsigned)__popcll(ms), nl = (unsigned)__popcll(ml);
                if (nc + ns + nl) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&s_ctl[C_RQ_TAIL], nc + ns + nl);
                    base = __shfl(base, 0, 64);
                    if (push_closest) s_rq[(base + (uint32_t)__popcll(mc & below)) & (ST_RQ_CAP - 1u)] = ls + 1u;
                    if (push_shadow) s_rq[(base + nc + (uint32_t)__popcll(ms & below)) & (ST_RQ_CAP - 1u)] = (ls | (1u << ST_KIND_SHIFT)) + 1u;
                    if (EXT && push_light) s_rq[(base + nc + ns + (uint32_t)__popcll(ml & below)) & (ST_RQ_CAP - 1u)] = (ls | (2u << ST_KIND_SHIFT)) + 1u;
                    if (lane == 0) atomicAdd(&s_ctl[C_RQ_COUNT], nc + ns + nl);      // (after the cells: LDS is in order per wave)
                }
                const unsigned long long mr = __ballot(retire);
                if (mr) {
                    uint32_t oldl = 0;
                    const uint32_t nr = (uint32_t)__popcll(mr);
                    if (lane == 0) {
                        oldl = atomicSub(&s_ctl[C_LIVE], nr);
                        if (oldl == nr) s_ctl[C_DONE] = 1;     // the workgroup's last slot has retired
                    }
                }
            }
            have = false;
        }
    }
    unsigned t0 = st_wave_sum(c_paths), t1 = st_wave_sum(c_bounce), t2 = st_wave_sum(c_rays), t3 = st_wave_sum(c_shaded), t4 = st_wave_sum(c_hdri);
    unsigned t5 = 0, t6 = 0, t7 = 0;
    if (COUNT) { t5 = st_wave_sum(c_nodes); t6 = st_wave_sum(c_tris); t7 = st_wave_sum(c_texels); }
    if (lane == 0) {
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
    return hipFuncGetAttributes(&a, (const void*)er_stream_kernel<false, false>);
}

void er_launch_stream(const DevScene& S, void* records, uint32_t slots, bool lights, void* spill, void* ring, uint32_t ring_cap, uint32_t* status,
                      uint32_t n_samples, bool count, uint32_t blocks, uint32_t tracers, hipStream_t stream) {
    static const uint32_t refill_min = [] {
        const char* e = getenv("ER_STREAM_REFILL_MIN");
        int v = e ? atoi(e) : 12;
        return (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
    }();
    static const uint32_t batch_min = [] {
        const char* e = getenv("ER_STREAM_BATCH_MIN");
        int v = e ? atoi(e) : 48;
        return (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
    }();
    if (S.owned_tile_count == 0 || n_samples == 0) return;
    const bool ext = er_ext_active(S);
    auto k = count ? (ext ? er_stream_kernel<true, true> : er_stream_kernel<true, false>) : (ext ? er_stream_kernel<false, true> : er_stream_kernel<false, false>);
    StState st;
    st.base = (char*)records;
    st.spill = (uint2*)spill;
    st.slots = slots;
    st.stride = er_stream_record_bytes(lights);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(ST_THREADS), 0, stream, S, st, (uint2*)ring, ring_cap, status, n_samples, tracers, refill_min, batch_min);
}

uint32_t er_stream_record_bytes(bool lights) { return lights ? ST_STRIDE_LIGHTS : ST_STRIDE_PLAIN; }
