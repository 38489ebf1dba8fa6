// ring_model.cpp -- host-thread model of ONE workgroup of the streaming schedule (csrc/er_stream.hip), built on the very ring
// functions the kernel uses (csrc/er_ring.h compiled with -DER_RING_HOST_MODEL): ray ring, shade ring, finish ring and the pixel ring with its
// "entry read" bits, all with SMALL capacities so that every ring wraps hundreds of times in a run, with the per-slot in-flight counters and the slot / pixel hand-offs
// in between.  "Waves" are threads of LANES lanes; what a wave does with one reservation (reserve n, put n cells, publish n;
// grant n, get n cells) is done in that order by its thread.  Slot records and per-pixel state are PLAIN memory, as on
// the device: only the protocol orders their accesses, so ThreadSanitizer (tests/test_stream_protocol_cpu.py builds this with
// -fsanitize=thread) reports any hand-off the protocol does not order.
//
// Checked at the end: every pixel received exactly `samples` samples, in order, never held by two slots; every ray was traced
// exactly once and its result was there when its slot was shaded; every ring ended empty with every cell in the state its lap
// implies; no guard expired.
//
//   ring_model <slots> <pixels> <samples> <variant> [tracers] [shaders] [rq_log2] [spec]
//     spec 1     with the speculative samples of round 6 (er_stream.hip ST_DRAWS_MASK): a pixel's next sample starts in a FREE slot (a fourth
//                checked ring) from a guessed stream state and is accumulated only after its predecessor and only if the predecessor left that
//                state -- the verdict word (exchange by the committing slot, compare-and-swap by a speculative slot that parks), the wake
//                through the finish ring and the dropped samples' slots falling free are the hand-offs; checked at the end: every pixel's
//                samples accumulated once, in order, each from the TRUE state (the pixel's stream state equals the sum of its samples' draws)
//     variant 0  the protocol of the kernel
//     variant 2  the same, but producers do NOT wait for the previous lap's reader (the protocol of round 2: plain overwrite) --
//                a negative control: with small rings this loses rays or reads the wrong lap, and the model says so
//     variant 3  the kernel's protocol with the model's ONE start-up precondition switched off: a ray ring with fewer cells than a
//                wave's reservation.  That is a REAL wait cycle (the producer's third put waits for the reader of its own first
//                entry, which cannot be granted before the producer publishes): the run ends in the guard and prints the state
//                dump -- what a cycle looks like, as opposed to a reader that was merely off its core.
//   On the first expired guard (wall clock, ER_RING_GUARD_MS) the model dumps, once: every ring's TAIL / COUNT / HEAD, and for every
//   thread its role and what it is waiting for (ring, position, the cell's word as last seen and the lap that position needs).
//   ring_model script
//     the SAME negative control as ONE scripted interleaving on one thread (no timing in it): a reader is granted a position and
//     stalls before reading its cell; the ring comes round; the checked producer of er_ring.h refuses to touch the cell until the
//     reader has been, round 2's producer overwrites it and the stalled reader never finds its entry.  Exit code 0 = both seen.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../elevenrender_amd/csrc/er_ring.h"

namespace {

constexpr int LANES = 4;            // lanes of a model wave
constexpr uint32_t SLOT_BITS = 10;  // as ST_SLOT_BITS
constexpr uint32_t FIN = 0x100u;    // as ST_FIN
constexpr uint32_t ESC = 0x200u, AMB1 = 0x400u, AMB2 = 0x800u;   // as ST_ESC, ST_AMB1, ST_AMB2

uint32_t hash32(uint32_t a) { a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16; return a; }

struct Slot {              // plain memory: one slot record (HBM on the device)
    uint32_t pixel = 0, left = 0, sample = 0, bounce = 0;
    uint32_t result[3] = {0, 0, 0};     // written by tracers: 1 + hash(ray identity)
    uint32_t pushed[3] = {0, 0, 0};     // ray identities the shader pushed for this step (0 = none)
    bool fin_next = false;
    uint32_t light = 0;                 // written by the shading step that ends the path, read by the finishing step (plain: the finish ring orders them)
    // speculative samples (plain unless said otherwise: the protocol orders them)
    std::atomic<uint32_t> spec_word{0}; // SP_* of a speculative slot: exchanged by the committing slot, compare-and-swapped by the slot itself
    uint32_t spec_link = 0;             // local slot + 1 of the speculative sample this slot's sample started
    uint64_t start_state = 0;           // the stream state this sample started from (a speculative one: the guess)
    uint32_t spec_entry = 0;            // the finish-ring payload a parked speculative slot arrived with
    bool is_spec = false;
};
enum { SP_NONE = 0, SP_PENDING = 1, SP_PARKED = 2, SP_VALID = 3, SP_INVALID = 4 };
struct Pixel {
    uint32_t done = 0;                  // plain: only the slot that accumulates touches it
    std::atomic<int> holders{0};        // slots running a NON-speculative sample of the pixel (never two)
    uint64_t state = 0;                 // the pixel's stream state = draws of all its accumulated samples (plain)
    uint32_t guess = 0, conf = 0;       // DevScene::px_draws
};

struct Ring {
    std::vector<uint32_t> cells;
    alignas(8) uint32_t ctl[ER_RING_WORDS] = {0, 0, 0, 0};
    uint32_t log2 = 0;
    void init(uint32_t l2) { log2 = l2; cells.assign(1u << l2, 0u); ctl[0] = ctl[1] = ctl[2] = ctl[3] = 0; }
};

struct Model {
    uint32_t n_slots, n_pixels, n_samples, variant;
    std::vector<Slot> slots;
    std::vector<Pixel> pixels;
    std::vector<uint32_t> s_wait;
    Ring rq, sq, fq, px, fr;   // px: only ctl is used as a ring; its cells are px_cells + px_bits; fr: free slots (speculative samples)
    bool spec = false;
    std::atomic<uint32_t> spec_started{0}, spec_right{0}, spec_wrong{0};
    uint32_t keep = 0;                  // 1 + the samples a pixel may be behind the most advanced one before it keeps its slot (0 = the rule is off)
    std::atomic<uint32_t> front{0xFFFFFFFFu}, kept{0};
    std::vector<uint64_t> px_cells;
    std::vector<uint32_t> px_bits;
    uint32_t px_cap = 0;
    uint32_t live = 0, done = 0;
    std::atomic<uint32_t> errors{0}, rays_traced{0}, rays_pushed{0}, shortcuts{0};
    // every model thread registers its wait word (er_ring.h) and its role: the state dump reads them
    struct ThreadInfo { const char* role; uint32_t id; const ErRingWait* wait; };
    std::mutex reg_mtx;
    std::vector<ThreadInfo> threads;
    std::atomic<bool> dumped{false};
    void reg(const char* role, uint32_t id) {
        std::lock_guard<std::mutex> lk(reg_mtx);
        threads.push_back({role, id, &er_ring_waiting});
    }
    const char* ring_name(const void* cells, uint32_t& log2) const {
        log2 = 0;
        if (cells == rq.cells.data()) { log2 = rq.log2; return "ray ring"; }
        if (cells == sq.cells.data()) { log2 = sq.log2; return "shade ring"; }
        if (cells == fq.cells.data()) { log2 = fq.log2; return "finish ring"; }
        if (cells == fr.cells.data()) { log2 = fr.log2; return "free ring"; }
        if (cells == px_bits.data()) return "pixel ring bits";
        if (cells == px_cells.data()) return "pixel ring cell";
        return "?";
    }
    // ONE dump per run, by the first thread whose guard expires: enough to tell a descheduled reader (everybody else idle or done, the
    // waited cell one step from the state the waiter needs, its reader / writer not waiting for anything) from a wait cycle
    // (threads waiting for each other's positions, COUNT 0 with TAIL - HEAD > 0).
    void dump_state(const char* why) {
        if (dumped.exchange(true)) return;
        std::lock_guard<std::mutex> lk(reg_mtx);
        fprintf(stderr, "==== ring model state dump (%s) ====\n", why);
        auto d = [&](const char* n, Ring& r) {
            fprintf(stderr, "  %-11s cap %4u  TAIL %u  COUNT %u  HEAD %u  (reserved and not yet granted: %u)\n", n, 1u << r.log2, er_ring_load(&r.ctl[ER_RING_TAIL]),
                    er_ring_load(&r.ctl[ER_RING_COUNT]), er_ring_load(&r.ctl[ER_RING_HEAD]), er_ring_load(&r.ctl[ER_RING_TAIL]) - er_ring_load(&r.ctl[ER_RING_HEAD]));
        };
        d("ray ring", rq); d("shade ring", sq); d("finish ring", fq); d("free ring", fr);
        fprintf(stderr, "  %-11s cap %4u  TAIL %u  COUNT %u  HEAD %u\n", "pixel ring", px_cap, er_ring_load(&px.ctl[ER_RING_TAIL]), er_ring_load(&px.ctl[ER_RING_COUNT]), er_ring_load(&px.ctl[ER_RING_HEAD]));
        fprintf(stderr, "  live slots %u, done %u\n", er_ring_load(&live), er_ring_load(&done));
        const auto now = std::chrono::steady_clock::now();
        for (const ThreadInfo& t : threads) {
            const char* what = __atomic_load_n(&t.wait->what, __ATOMIC_ACQUIRE);
            if (!what) { fprintf(stderr, "  %s %u: not waiting on a ring cell\n", t.role, t.id); continue; }
            uint32_t log2 = 0;
            const char* rn = ring_name(t.wait->cells, log2);
            const uint32_t pos = t.wait->pos, seen = __atomic_load_n(&t.wait->seen, __ATOMIC_RELAXED);
            const double ms = std::chrono::duration<double, std::milli>(now - t.wait->since).count();
            if (log2 || t.wait->cells == rq.cells.data())
                fprintf(stderr, "  %s %u: WAITING %.0f ms in %s, %s position %u (cell %u, needs lap %u); cell word last seen: lap %u %s payload %u\n", t.role, t.id, ms, what, rn, pos,
                        pos & ((1u << log2) - 1u), pos >> log2, seen >> ER_RING_LAP_SHIFT, (seen & ER_RING_FULL) ? "FULL" : "empty", seen & ER_RING_PAYLOAD_MASK);
            else
                fprintf(stderr, "  %s %u: WAITING %.0f ms in %s, %s index %u\n", t.role, t.id, ms, what, rn, pos);
        }
        fprintf(stderr, "==== end of dump ====\n");
    }

    void err(const char* what) {
        if (errors.fetch_add(1) < 10) fprintf(stderr, "model error: %s\n", what);
        if (strstr(what, "guard expired")) { dump_state(what); er_ring_store(&done, 1u); }      // (the run ends: the other threads leave at their next poll or at their own guard)
    }
    static uint32_t lap_tag(uint32_t pos, uint32_t cap) { return ((((pos / cap) & 0x7Fu) + 1u) << 24); }
    uint32_t path_len(uint32_t pixel, uint32_t sample) const { return 1u + hash32(pixel * 977u + sample * 131u + 7u) % 5u; }
    // does the path's last closest-hit ray leave the scene (else it ends at the bounce limit)?  is a shadow query's verdict ambiguous?
    bool escapes(uint32_t pixel, uint32_t sample) const { return (hash32(pixel * 419u + sample * 61u + 3u) & 3u) != 0; }
    bool ambiguous(uint32_t ident) const { return (hash32(ident ^ 0x5bd1e995u) & 7u) == 0; }
    // random numbers a sample draws: mostly the pixel's own figure, now and then another (so that guesses are right often and wrong sometimes)
    uint32_t draws(uint32_t pixel, uint32_t sample) const { return 5u + 5u * ((hash32(pixel * 0x9e37u + 11u) + ((hash32(pixel * 131u + sample * 977u) & 3u) == 0u ? 1u : 0u)) % 3u); }

    // wave-level push of up to LANES payloads (one reservation)
    void push(Ring& r, const uint32_t* payload, int n, bool checked = true) {
        if (n == 0) return;
        const uint32_t base = er_ring_reserve(r.ctl, (uint32_t)n);
        for (int i = 0; i < n; i++) {
            if (checked) {
                if (!er_ring_put(r.cells.data(), r.log2, base + (uint32_t)i, payload[i])) err("put guard expired");
            } else {        // round 2's producer: overwrite whatever is there
                uint32_t* cell = r.cells.data() + ((base + i) & ((1u << r.log2) - 1u));
                er_ring_store(cell, er_ring_lap(base + i, r.log2) | ER_RING_FULL | payload[i]);
            }
        }
        er_ring_publish(r.ctl, (uint32_t)n);
    }
    int take(Ring& r, uint32_t want, uint32_t* out) {
        uint32_t base = 0;
        const uint32_t g = er_ring_grant(r.ctl, want, base);
        for (uint32_t i = 0; i < g; i++)
            if (!er_ring_get(r.cells.data(), r.log2, base + i, out[i])) err("get guard expired");
        return (int)g;
    }

    void begin_sample(uint32_t s, uint32_t pixel, uint32_t left, bool speculative = false, uint32_t sample = 0, uint64_t guessed_state = 0) {
        Slot& S = slots[s];
        if (!speculative && pixels[pixel].holders.fetch_add(1) != 0) err("a pixel is held by two slots");
        S.pixel = pixel; S.left = left; S.sample = speculative ? sample : pixels[pixel].done; S.bounce = 0; S.fin_next = false;
        S.is_spec = speculative; S.spec_link = 0; S.spec_entry = 0;
        S.start_state = speculative ? guessed_state : pixels[pixel].state;
        S.spec_word.store(speculative ? SP_PENDING : SP_NONE, std::memory_order_relaxed);
        S.pushed[0] = 1u + hash32(pixel * 31u + S.sample * 7u);      // the camera ray
        S.pushed[1] = S.pushed[2] = 0;
        S.result[0] = S.result[1] = S.result[2] = 0;
        er_ring_store(&s_wait[s], 1u);
    }

    // ---- what a tracer does when a ray is complete ----
    void finish_ray(uint32_t s, uint32_t kind, uint32_t ident, uint32_t* sq_out, int& n_sq, uint32_t* fq_out, int& n_fq) {
        Slot& S = slots[s];
        if (S.pushed[kind] != ident) err("a tracer holds a ray its slot did not push");
        if (S.result[kind] != 0) err("a ray was traced twice");
        S.result[kind] = 1u + hash32(ident);
        rays_traced.fetch_add(1);
        // the flags the kernel's tracers add with the decrement: the path's last ray escapes / a verdict is ambiguous
        uint32_t add = (uint32_t)-1;
        if (kind == 0 && S.bounce + 1 >= path_len(S.pixel, S.sample) && escapes(S.pixel, S.sample)) add += ESC;
        if (kind != 0 && ambiguous(ident)) add += kind == 1 ? AMB1 : AMB2;
        const uint32_t old = er_ring_add(&s_wait[s], add);
        const uint32_t fin = old + add;
        if ((old & 0xFFu) == 0) err("in-flight counter below zero");
        if ((fin & 0xFFu) == 0u) {
            if ((fin & ESC) && !(fin & (AMB1 | AMB2))) fq_out[n_fq++] = s | (1u << SLOT_BITS);     // straight to the finish ring
            else sq_out[n_sq++] = s | ((fin & FIN) ? (1u << SLOT_BITS) : 0u);
        }
    }

    void first_tracer(uint32_t id) {
        reg("tracer", id);
        uint32_t progress = 0, idle = 0;
        while (true) {
            uint32_t e[LANES];
            const int g = take(rq, LANES, e);
            if (g == 0) {
                if (er_ring_load(&done)) break;
                if ((idle & 16383u) == 16383u) std::this_thread::sleep_for(std::chrono::milliseconds(1)); else std::this_thread::yield();      // (an idle wave mostly yields)
                const uint32_t pr = er_ring_load(&rq.ctl[ER_RING_TAIL]) + er_ring_load(&sq.ctl[ER_RING_TAIL]) + er_ring_load(&fq.ctl[ER_RING_TAIL]);
                if (pr != progress) { progress = pr; idle = 0; }
                if (++idle > 40000000u) { err("tracer watchdog"); er_ring_store(&done, 1u); break; }
                continue;
            }
            idle = 0;
            uint32_t out[LANES], fout[LANES];
            int n = 0, nf = 0;
            for (int i = 0; i < g; i++) {
                const uint32_t s = e[i] & ((1u << SLOT_BITS) - 1u), kind = e[i] >> SLOT_BITS;
                if (s >= n_slots || kind > 2) { err("garbage ray-ring entry"); continue; }
                finish_ray(s, kind, slots[s].pushed[kind], out, n, fout, nf);
            }
            push(sq, out, n, variant != 2);
            push(fq, fout, nf, variant != 2);
        }
    }

    void shader(uint32_t id) {
        reg("shader", id);
        uint32_t progress = 0, idle = 0;
        while (true) {
            // a full finishing batch first, else a shading batch, else whatever the finish ring holds (er_stream.hip's shader loop)
            uint32_t e[LANES];
            bool fin_mode = er_ring_peek_count(er_ring_peek(fq.ctl)) >= (uint32_t)LANES;
            int g = fin_mode ? take(fq, LANES, e) : 0;
            if (g == 0) { fin_mode = false; g = take(sq, LANES, e); }
            if (g == 0) { fin_mode = true; g = take(fq, LANES, e); }
            if (g == 0) {
                if (er_ring_load(&done)) break;
                if ((idle & 16383u) == 16383u) std::this_thread::sleep_for(std::chrono::milliseconds(1)); else std::this_thread::yield();
                const uint32_t pr = er_ring_load(&rq.ctl[ER_RING_TAIL]) + er_ring_load(&sq.ctl[ER_RING_TAIL]) + er_ring_load(&fq.ctl[ER_RING_TAIL]);
                if (pr != progress) { progress = pr; idle = 0; }
                if (++idle > 40000000u) { err("shader watchdog"); er_ring_store(&done, 1u); break; }
                continue;
            }
            idle = 0;
            uint32_t rays[4 * LANES];
            int n_rays = 0;
            uint32_t want_px[LANES], back_px[LANES], back_left[LANES];
            int n_want = 0, n_back = 0;
            uint32_t fin[LANES];
            int n_fin = 0;
            uint32_t wake[LANES], frees[2 * LANES], spawned[LANES];
            int n_wake = 0, n_free = 0, n_spawned = 0;
            uint32_t dropped = 0;
            uint32_t kept_slot[LANES], kept_px[LANES], kept_left[LANES];
            int n_kept = 0;
            // the pixel's next sample at once, in a free slot, from the state the sample just begun in slot `h` leaves if it draws what the pixel's samples have been drawing
            auto spawn = [&](uint32_t h, uint32_t pixel, uint32_t left) {
                if (!(spec && left > 1u && pixels[pixel].conf >= 2u)) return;
                uint32_t f = 0;
                if (take(fr, 1u, &f) == 1) {
                    Slot& H = slots[h];
                    begin_sample(f, pixel, left - 1u, true, H.sample + 1u, H.start_state + pixels[pixel].guess);
                    H.spec_link = f + 1u;
                    er_ring_add(&live, 1u);
                    spec_started.fetch_add(1);
                    spawned[n_spawned++] = f;
                }
            };
            for (int i = 0; i < g && fin_mode; i++) {
                // FINISHING step: the sample is accumulated, the pixel goes back, the slot takes the next one
                const uint32_t s = e[i] & ((1u << SLOT_BITS) - 1u);
                const bool from_tracer = (e[i] >> SLOT_BITS) != 0;
                if (s >= n_slots) { err("garbage finish-ring entry"); continue; }
                Slot& S = slots[s];
                if (spec && S.is_spec) {
                    // a speculative sample reaches its finishing step: verdict there -> accumulate or drop; not yet -> park (er_stream.hip)
                    uint32_t sw = S.spec_word.load(std::memory_order_acquire);
                    if (sw == SP_PENDING) {
                        S.spec_entry = e[i];
                        uint32_t expect = SP_PENDING;
                        if (S.spec_word.compare_exchange_strong(expect, SP_PARKED, std::memory_order_acq_rel)) continue;      // the committing slot wakes it
                        sw = expect;
                    }
                    if (sw == SP_INVALID) {      // dropped: nothing of it is accumulated, the slot falls free
                        for (int k = 0; k < 3; k++) { S.pushed[k] = 0; S.result[k] = 0; }
                        S.light = 0; S.is_spec = false;
                        frees[n_free++] = s; dropped++;
                        continue;
                    }
                    if (sw != SP_VALID) { err("a speculative slot in an impossible state"); continue; }
                    S.is_spec = false;
                    S.spec_word.store(SP_NONE, std::memory_order_relaxed);
                    if (pixels[S.pixel].holders.fetch_add(1) != 0) err("a pixel is held by two slots");      // it is the pixel's sample in flight now
                }
                if (from_tracer) {
                    shortcuts.fetch_add(1);
                    // straight from the tracers: the last ray escaped and no verdict is ambiguous -- what the shading step checks and does
                    if (S.fin_next) err("a finalise-only slot came through the tracers' finish entry");
                    for (int k = 0; k < 3; k++) {
                        if (S.pushed[k] && S.result[k] != 1u + hash32(S.pushed[k])) err("a slot was finished before its ray was traced");
                        if (!S.pushed[k] && S.result[k]) err("a result without a ray");
                        if (k && S.pushed[k] && ambiguous(S.pushed[k])) err("an ambiguous verdict took the short cut");
                        S.pushed[k] = 0; S.result[k] = 0;
                    }
                    S.bounce++;
                    if (S.bounce < path_len(S.pixel, S.sample) || !escapes(S.pixel, S.sample)) err("a path that has not left the scene took the short cut");
                } else {
                    if (S.light != 1u + hash32(S.pixel * 53u + S.sample * 19u)) err("a sample was finished before its path was over");
                    S.light = 0;
                }
                Pixel& P = pixels[S.pixel];
                if (P.done != S.sample) err("a pixel's samples ran out of order");
                if (S.start_state != P.state) err("a sample was accumulated from a wrong guess of the stream state");
                const uint32_t nd = draws(S.pixel, S.sample);
                P.state += nd;
                if (nd == P.guess) P.conf = P.conf < 7u ? P.conf + 1u : 7u;
                else if (P.conf >= 2u) P.conf -= 2u;
                else { P.guess = nd; P.conf = 1u; }
                P.done++;
                P.holders.fetch_sub(1);
                bool has_spec = false;
                if (spec && S.spec_link) {
                    // the verdict for the speculative sample this one started: right iff this sample left exactly the state it assumed
                    Slot& F = slots[S.spec_link - 1u];
                    const bool ok = F.start_state == P.state;
                    const uint32_t old = F.spec_word.exchange(ok ? SP_VALID : SP_INVALID, std::memory_order_acq_rel);
                    (ok ? spec_right : spec_wrong).fetch_add(1);
                    if (old == SP_PARKED) wake[n_wake++] = F.spec_entry;
                    else if (old != SP_PENDING) err("a verdict written twice");
                    has_spec = ok;
                    S.spec_link = 0;
                }
                // a pixel that is `keep` - 1 or more samples behind the most advanced one goes on in the slot it has (er_stream.hip s_front)
                bool keep_own = false;
                if (keep) {
                    const uint32_t la = S.left - 1;
                    uint32_t fr0 = front.load(std::memory_order_relaxed);
                    while (la < fr0 && !front.compare_exchange_weak(fr0, la, std::memory_order_relaxed)) {}
                    keep_own = la > 0 && !has_spec && la >= fr0 + keep;
                }
                if (keep_own) { kept_slot[n_kept] = s; kept_px[n_kept] = S.pixel; kept_left[n_kept] = S.left - 1; n_kept++; kept.fetch_add(1); }
                else {
                    if (S.left - 1 > 0 && !has_spec) { back_px[n_back] = S.pixel; back_left[n_back] = S.left - 1; n_back++; }
                    want_px[n_want++] = s;
                }
            }
            for (int i = 0; i < g && !fin_mode; i++) {
                const uint32_t s = e[i] & ((1u << SLOT_BITS) - 1u);
                const bool fin_only = (e[i] >> SLOT_BITS) != 0;
                if (s >= n_slots) { err("garbage shade-ring entry"); continue; }
                Slot& S = slots[s];
                if (fin_only != S.fin_next) err("finalise flag lost");
                for (int k = 0; k < 3; k++) {        // every pushed ray has its result, and only those
                    if (S.pushed[k] && S.result[k] != 1u + hash32(S.pushed[k])) err("a slot was shaded before its ray was traced");
                    if (!S.pushed[k] && S.result[k]) err("a result without a ray");
                    S.pushed[k] = 0; S.result[k] = 0;
                }
                if (spec && S.is_spec && S.spec_word.load(std::memory_order_acquire) == SP_INVALID) {      // dropped at its next step
                    S.light = 0; S.is_spec = false; S.fin_next = false;
                    frees[n_free++] = s; dropped++;
                    continue;
                }
                bool donep = fin_only;
                bool pend_shadow = false, pend_light = false;
                if (!fin_only) {
                    S.bounce++;
                    const uint32_t h = hash32(S.pixel * 13u + S.sample * 5u + S.bounce);
                    pend_shadow = (h & 1u) != 0;
                    pend_light = (h & 6u) == 6u;
                    if (S.bounce >= path_len(S.pixel, S.sample)) {
                        donep = true;
                        if (escapes(S.pixel, S.sample)) pend_shadow = pend_light = false;      // (a ray that left the scene starts no query)
                    }
                }
                if (donep && (pend_shadow || pend_light)) {
                    S.fin_next = true;
                } else if (donep) {
                    // the path is over, nothing pending: the sample goes to the finish ring, its light through the slot record
                    S.fin_next = false;
                    S.light = 1u + hash32(S.pixel * 53u + S.sample * 19u);
                    fin[n_fin++] = s;
                    continue;
                }
                uint32_t n = 0;
                if (!donep) { S.pushed[0] = 1u + hash32(S.pixel * 31u + S.sample * 7u + S.bounce * 1009u); rays[n_rays++] = s; n++; }
                if (pend_shadow) { S.pushed[1] = 2u + hash32(S.pixel * 37u + S.sample * 11u + S.bounce * 2003u); rays[n_rays++] = s | (1u << SLOT_BITS); n++; }
                if (pend_light) { S.pushed[2] = 3u + hash32(S.pixel * 41u + S.sample * 17u + S.bounce * 3001u); rays[n_rays++] = s | (2u << SLOT_BITS); n++; }
                er_ring_store(&s_wait[s], n + (S.fin_next ? FIN : 0u));
            }
            push(fq, fin, n_fin, variant != 2);
            // finished samples: pixels back to the ring (cell behind its "entry read" bit), then as many taken as there are
            uint32_t retire = 0;
            if (n_want) {
                if (n_back) {
                    const uint32_t base = er_ring_reserve(px.ctl, (uint32_t)n_back);
                    for (int i = 0; i < n_back; i++) {
                        const uint32_t pos = base + (uint32_t)i, cell = pos & (px_cap - 1u);
                        if (variant != 2 && !er_bits_acquire(px_bits.data(), cell)) err("pixel bit guard expired");
                        __atomic_store_n(&px_cells[cell], ((uint64_t)(back_left[i] | lap_tag(pos, px_cap)) << 32) | back_px[i], __ATOMIC_RELEASE);
                    }
                    er_ring_publish(px.ctl, (uint32_t)n_back);
                }
                uint32_t base = 0;
                const uint32_t got = er_ring_grant(px.ctl, (uint32_t)n_want, base);
                for (int i = 0; i < n_want; i++) {
                    if ((uint32_t)i >= got) { retire++; continue; }
                    const uint32_t pos = base + (uint32_t)i, cell = pos & (px_cap - 1u);
                    uint64_t w;
                    uint32_t guard = 0;
                    er_ring_wait_begin("take: the writer of the pixel cell", px_cells.data(), pos);
                    while ((((uint32_t)((w = __atomic_load_n(&px_cells[cell], __ATOMIC_ACQUIRE)) >> 32)) & 0xFF000000u) != lap_tag(pos, px_cap)) {
                        if (er_ring_expired(guard, (uint32_t)(w >> 32))) { err("pixel cell guard expired"); break; }
                        er_ring_pause();
                    }
                    er_ring_wait_end();
                    if (variant != 2) er_bits_release(px_bits.data(), cell);
                    const uint32_t left = (uint32_t)(w >> 32) & 0x00FFFFFFu, pixel = (uint32_t)w;
                    if (pixel >= n_pixels || left == 0) { err("garbage pixel-ring entry"); retire++; continue; }
                    begin_sample(want_px[i], pixel, left);
                    rays[n_rays++] = want_px[i];
                    spawn(want_px[i], pixel, left);
                }
                for (int i = 0; i < n_want; i++) if ((uint32_t)i >= got && spec) frees[n_free++] = want_px[i];      // no pixel left for the slot: it is free
            }
            for (int i = 0; i < n_kept; i++) {      // the pixels that went on in their slots: no ring, the next sample from the state this one left
                begin_sample(kept_slot[i], kept_px[i], kept_left[i]);
                rays[n_rays++] = kept_slot[i];
                spawn(kept_slot[i], kept_px[i], kept_left[i]);
            }
            for (int i = 0; i < n_spawned; i++) rays[n_rays++] = spawned[i];
            push(fq, wake, n_wake, variant != 2);
            // slots without a sample fall free: the dropped speculative ones (which also leave the count of samples in flight) and those that found no pixel
            push(fr, frees, n_free > LANES ? LANES : n_free, variant != 2);
            if (n_free > LANES) push(fr, frees + LANES, n_free - LANES, variant != 2);
            retire += dropped;
            // publish the rays (one reservation per LANES entries, as st_push)
            rays_pushed.fetch_add((uint32_t)n_rays);
            for (int o = 0; o < n_rays; o += LANES) push(rq, rays + o, n_rays - o < LANES ? n_rays - o : LANES, variant != 2);
            if (retire) {
                const uint32_t old = er_ring_add(&live, (uint32_t)-(int32_t)retire);
                if (old == retire) er_ring_store(&done, 1u);
            }
        }
    }

    int run(uint32_t tracers, uint32_t shaders, uint32_t rq_log2) {
        slots = std::vector<Slot>(n_slots);
        pixels = std::vector<Pixel>(n_pixels);
        s_wait.assign(n_slots, 0u);
        uint32_t sq_log2 = 0;
        while ((1u << sq_log2) < n_slots) sq_log2++;
        rq.init(rq_log2); sq.init(sq_log2); fq.init(sq_log2); fr.init(sq_log2);
        px_cap = 1;
        while (px_cap < n_pixels) px_cap <<= 1;
        px_cells.assign(px_cap, 0);
        px_bits.assign((px_cap + 31) / 32, 0u);
        const uint32_t in_slots = n_slots < n_pixels ? n_slots : n_pixels;
        for (uint32_t p = in_slots; p < n_pixels; p++) {
            const uint32_t pos = p - in_slots;
            px_bits[pos >> 5] |= 1u << (pos & 31u);
            px_cells[pos] = ((uint64_t)(n_samples | lap_tag(pos, px_cap)) << 32) | p;
        }
        px.ctl[ER_RING_TAIL] = px.ctl[ER_RING_COUNT] = n_pixels - in_slots;
        px.ctl[ER_RING_HEAD] = 0;
        live = in_slots;
        done = in_slots == 0 ? 1u : 0u;
        std::vector<uint32_t> first;
        for (uint32_t s = 0; s < in_slots; s++) { begin_sample(s, s, n_samples); first.push_back(s); }
        if (spec) {      // the slots no pixel started in are free from the beginning
            std::vector<uint32_t> idle;
            for (uint32_t s = in_slots; s < n_slots; s++) idle.push_back(s);
            for (size_t o = 0; o < idle.size(); o += LANES) push(fr, idle.data() + o, (int)(idle.size() - o < (size_t)LANES ? idle.size() - o : LANES));
        }
        rays_pushed.fetch_add(in_slots);
        std::vector<std::thread> th;
        reg("main (camera rays)", 0);
        for (uint32_t t = 0; t < tracers; t++) th.emplace_back([this, t] { first_tracer(t); });
        for (uint32_t t = 0; t < shaders; t++) th.emplace_back([this, t] { shader(t); });
        // (the camera rays go in while the waves already run: the model's ray ring may be smaller than the slots, the kernel's is
        // not -- static_assert in er_stream.hip -- and fills it before its waves start)
        for (size_t o = 0; o < first.size(); o += LANES) push(rq, first.data() + o, (int)(first.size() - o < (size_t)LANES ? first.size() - o : LANES));
        std::atomic<bool> stop{false};
        std::thread monitor([&] {       // ER_MODEL_DEBUG=1: the rings' counters once a second (to see where a stuck run is stuck)
            if (!getenv("ER_MODEL_DEBUG")) return;
            while (!stop.load()) {
                std::this_thread::sleep_for(std::chrono::seconds(1));
                auto d = [&](const char* n, Ring& r) { fprintf(stderr, " %s t%u c%u h%u", n, er_ring_load(&r.ctl[ER_RING_TAIL]), er_ring_load(&r.ctl[ER_RING_COUNT]), er_ring_load(&r.ctl[ER_RING_HEAD])); };
                d("rq", rq); d("sq", sq); d("fq", fq); d("px", px);
                fprintf(stderr, " live %u done %u\n", er_ring_load(&live), er_ring_load(&done));
            }
        });
        for (auto& t : th) t.join();
        stop.store(true);
        monitor.join();
        uint32_t short_px = 0;
        for (uint32_t p = 0; p < n_pixels; p++) {
            if (pixels[p].done != n_samples) short_px++;
            uint64_t want = 0;      // every sample accumulated once, from the true state: the stream state is the sum of the samples' draws
            for (uint32_t k = 0; k < n_samples; k++) want += draws(p, k);
            if (pixels[p].state != want) { short_px++; if (errors.fetch_add(1) < 10) fprintf(stderr, "model error: pixel %u ends in stream state %llu, not %llu\n", p, (unsigned long long)pixels[p].state, (unsigned long long)want); }
        }
        auto ring_clean = [&](Ring& r, const char* name, uint32_t expect) {
            if (r.ctl[ER_RING_COUNT] != expect || r.ctl[ER_RING_TAIL] - r.ctl[ER_RING_HEAD] != expect) { fprintf(stderr, "%s: count %u, tail - head %u, expected %u\n", name, r.ctl[ER_RING_COUNT], r.ctl[ER_RING_TAIL] - r.ctl[ER_RING_HEAD], expect); return 1u; }
            // every cell in the state its next position implies: the first position >= HEAD that maps to the cell is either
            // inside [HEAD, TAIL) -- its entry is there, (lap, full) -- or still to be written, (lap, empty)
            uint32_t bad = 0;
            const uint32_t cap = 1u << r.log2, head = r.ctl[ER_RING_HEAD];
            for (uint32_t i = 0; i < cap; i++) {
                const uint32_t pos = head + ((i - head) & (cap - 1u));
                const bool inside = (uint32_t)(pos - head) < expect;
                const uint32_t want = er_ring_lap(pos, r.log2) | (inside ? ER_RING_FULL : 0u);
                if ((r.cells[i] & ~ER_RING_PAYLOAD_MASK) != want) bad++;
            }
            if (bad) fprintf(stderr, "%s: %u cells in an impossible state\n", name, bad);
            return bad;
        };
        uint32_t bad = 0;
        if (variant != 2) {
            bad += ring_clean(rq, "ray ring", 0) + ring_clean(sq, "shade ring", 0) + ring_clean(fq, "finish ring", 0);
            for (uint32_t w : px_bits) if (w) bad++;
        }
        if (px.ctl[ER_RING_COUNT] != 0) bad++;
        if (spec && variant != 2) {      // at the end every slot is free, once
            bad += ring_clean(fr, "free ring", n_slots);
            std::vector<uint32_t> seen(n_slots, 0u);
            for (uint32_t i = 0; i < (1u << fr.log2); i++) if (fr.cells[i] & ER_RING_FULL) { const uint32_t v = fr.cells[i] & ER_RING_PAYLOAD_MASK; if (v < n_slots) seen[v]++; }
            for (uint32_t v : seen) if (v != 1u) bad++;
            printf("speculative samples: %u started, %u guesses right, %u wrong\n", spec_started.load(), spec_right.load(), spec_wrong.load());
        }
        if (keep) printf("samples begun in the slot their pixel had: %u\n", kept.load());
        const uint32_t lost = rays_pushed.load() - rays_traced.load();
        printf("variant %u slots %u pixels %u samples %u: %u pixels short, %u rays pushed, %u lost, %u protocol errors, %u ring faults, laps: ray ring %u, pixel ring %u, short cuts %u\n",
               variant, in_slots, n_pixels, n_samples, short_px, rays_pushed.load(), lost, errors.load(), bad, rq.ctl[ER_RING_TAIL] >> rq.log2,
               px.ctl[ER_RING_TAIL] / px_cap, shortcuts.load());
        return (short_px || lost || errors.load() || bad) ? 1 : 0;
    }
};

}  // namespace

// The lapped-overwrite interleaving of round 2, forced: deterministic, single-threaded, on the ring functions themselves.
static int scripted_negative_control() {
    const uint32_t log2 = 2, cap = 1u << log2;
    int failures = 0;
    for (int checked = 1; checked >= 0; checked--) {
        uint32_t cells[4] = {0, 0, 0, 0};
        alignas(8) uint32_t ctl[ER_RING_WORDS] = {0, 0, 0, 0};
        // lap 0: four entries written and published; a reader is granted position 0 and STALLS before it reads its cell
        uint32_t base = er_ring_reserve(ctl, cap);
        for (uint32_t i = 0; i < cap; i++)
            if (!er_ring_put(cells, log2, base + i, 100u + i)) { printf("script: put of lap 0 failed\n"); failures++; }
        er_ring_publish(ctl, cap);
        uint32_t gbase = 0;
        const uint32_t granted = er_ring_grant(ctl, 1u, gbase);
        if (granted != 1u || gbase != 0u) { printf("script: unexpected grant\n"); failures++; }
        // the ring comes round: a producer is handed position 4 = cell 0 of lap 1 while the reader of lap 0 has not been there
        base = er_ring_reserve(ctl, 1u);
        bool put_ok;
        if (checked) {
            put_ok = er_ring_put(cells, log2, base, 200u);        // must WAIT for the reader (here: until the guard expires)
        } else {
            er_ring_store(&cells[base & (cap - 1u)], er_ring_lap(base, log2) | ER_RING_FULL | 200u);      // round 2's producer
            put_ok = true;
        }
        // now the stalled reader reads position 0
        uint32_t payload = 0;
        const bool got = er_ring_get(cells, log2, gbase, payload);
        if (checked) {
            const bool ok = !put_ok && got && payload == 100u;
            printf("script: checked producer %s; the stalled reader then %s its entry (payload %u)\n", put_ok ? "OVERWROTE an unread cell" : "waited for the reader (guard)",
                   got ? "found" : "LOST", payload);
            if (!ok) failures++;
        } else {
            const bool ok = !got;      // the entry of lap 0 is gone: the reader waits for a cell state that can no longer come
            printf("script: round 2's producer overwrote the unread cell; the stalled reader %s\n", got ? "still found an entry (payload of the wrong lap?)" : "never finds its entry (guard)");
            if (!ok) failures++;
        }
    }
    return failures ? 1 : 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && strcmp(argv[1], "script") == 0) return scripted_negative_control();
    Model m;
    m.n_slots = argc > 1 ? (uint32_t)atoi(argv[1]) : 8;
    m.n_pixels = argc > 2 ? (uint32_t)atoi(argv[2]) : 8;
    m.n_samples = argc > 3 ? (uint32_t)atoi(argv[3]) : 50;
    m.variant = argc > 4 ? (uint32_t)atoi(argv[4]) : 0;
    const uint32_t tracers = argc > 5 ? (uint32_t)atoi(argv[5]) : 3, shaders = argc > 6 ? (uint32_t)atoi(argv[6]) : 2;
    const uint32_t rq_log2 = argc > 7 ? (uint32_t)atoi(argv[7]) : 3;
    m.spec = argc > 8 && atoi(argv[8]) != 0;
    m.keep = argc > 9 ? (uint32_t)atoi(argv[9]) : 0;
    if (m.n_slots > (1u << SLOT_BITS) || m.n_slots == 0 || m.n_pixels == 0 || (m.variant != 0 && m.variant != 2 && m.variant != 3)) return 2;
    // Start-up preconditions (the kernel's are static_asserts in er_stream.hip):
    //  * a wave's reservation must fit the ray ring -- else its later puts wait for readers of its own unpublished entries: a cycle.
    //    The model REFUSES such a configuration unless asked for it on purpose (variant 3);
    //  * the kernel's ray ring also holds every ray its slots can have in flight (2 per slot, 3 with the light extension), so that
    //    its producers never wait at all.  The model runs BELOW that on purpose -- small rings are what makes producers wait and
    //    positions come round -- and says so.
    const bool fits = (1u << rq_log2) >= (uint32_t)LANES;
    fprintf(stderr, "model: ray ring of %u cells, reservations of up to %d: %s; %u slots x 3 rays %s the ring (the kernel asserts they do: its producers never wait)\n",
            1u << rq_log2, LANES, fits ? "fit" : "DO NOT FIT (a wait cycle)", m.n_slots, (1u << rq_log2) >= 3u * m.n_slots ? "fit" : "do not fit");
    if (!fits && m.variant != 3) { fprintf(stderr, "model: refused (variant 3 runs it on purpose)\n"); return 2; }
    if (m.variant == 3) m.variant = 0;
    return m.run(tracers, shaders, rq_log2);
}
