// er_ring.h -- the bounded multi-producer / multi-consumer rings of the CU-resident streaming schedule (er_stream.hip),
// written once for the device (LDS words, LDS atomics) and for the host-thread model of the protocol
// (tests/native/ring_model.cpp, compiled with -DER_RING_HOST_MODEL and run under ThreadSanitizer on the CPU).
//
// Why this file exists: the first three ring protocols of the streaming schedule each rested on a timing argument ("a lap of
// the ring takes longer than a lane needs to read the cell it was granted") and two of them failed on hardware.  The rule
// here is the bounded-queue rule of Vyukov's MPMC ring: every cell carries the LAP of the position it serves and whether it is
// FULL, producers and consumers both CHECK the cell before they touch it and nobody ever clears anything:
//
//     cell = lap(18 bits) | full(1 bit) | payload(13 bits)          all cells start as lap 0, empty
//     producer of position p:  wait until cell == (lap(p), empty)   -> store (lap(p), full, payload)
//     consumer of position p:  wait until cell == (lap(p), full, *) -> take payload, store (lap(p) + 1, empty)
//
// A cell can therefore only walk the chain (L, empty) -> (L, full) -> (L + 1, empty) -> ...: a producer that comes round to a
// cell whose previous entry has been granted but not yet read WAITS for that reader (it is on its way: a granted lane reads
// its cell unconditionally), and a consumer that is granted a position whose producer has reserved but not yet written it
// waits for that writer.  No entry can be lost, duplicated or read from the wrong lap, whatever the interleaving; the only
// number that could alias is the 18-bit lap, i.e. a lane would have to stall between two adjacent instructions while its ring
// turns 262 144 times.  (Positions are 32-bit and wrap; capacities are powers of two, so `pos & (cap - 1)` and `pos >> log2(cap)`
// stay consistent across the wrap.)
//
// Counters of a ring: TAIL = positions reserved by producers, COUNT = entries published (written) and not yet granted, HEAD =
// positions granted to consumers (COUNT and HEAD share one 64-bit word).  A producer wave reserves n positions with one add to TAIL, its lanes put
// their cells, then it adds n to COUNT; a consumer wave is granted min(want, COUNT) entries AND their positions with one compare-and-swap on
// (COUNT, HEAD) (exact at every instant: never below zero).  COUNT counts entries, not positions:
// because reservations complete out of order, the cells at the granted positions need not be the ones whose writers have
// published -- that is what the per-cell wait is for.
//
// The pixel ring of a workgroup lives in HBM (its capacity scales with the frame) and keeps its own cell format (pixel,
// samples left | lap tag: er_stream.hip); its "previous entry has been read" check is one bit per cell in LDS
// (er_bits_acquire / er_bits_release below): the producer sets the bit before it writes the cell and waits while it is
// still set, the consumer clears it after it has read the cell.
#pragma once
#include <stdint.h>

#ifdef ER_RING_HOST_MODEL
#include <chrono>
#include <thread>
#define ER_RING_FN static inline
ER_RING_FN uint32_t er_ring_load(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
ER_RING_FN void er_ring_store(uint32_t* p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
ER_RING_FN uint32_t er_ring_add(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
ER_RING_FN uint32_t er_ring_or(uint32_t* p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_ACQ_REL); }
ER_RING_FN uint32_t er_ring_and(uint32_t* p, uint32_t v) { return __atomic_fetch_and(p, v, __ATOMIC_ACQ_REL); }
ER_RING_FN uint32_t er_ring_cas(uint32_t* p, uint32_t expect, uint32_t desired) {
    __atomic_compare_exchange_n(p, &expect, desired, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
    return expect;      // (the value found, like atomicCAS)
}
ER_RING_FN unsigned long long er_ring_load64(const unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
ER_RING_FN unsigned long long er_ring_add64(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
ER_RING_FN unsigned long long er_ring_cas64(unsigned long long* p, unsigned long long expect, unsigned long long desired) {
    __atomic_compare_exchange_n(p, &expect, desired, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE);
    return expect;
}
// (a waiter yields; every 8192nd poll of a thread it sleeps a millisecond instead, so that on an oversubscribed machine -- the CPU
// suite beside a compile job -- the thread it waits for gets its core long before the poll guard expires: round 5 saw a correct run
// of the model end in "put guard expired" under exactly that load)
ER_RING_FN void er_ring_pause() {
    static thread_local uint32_t polls = 0;
    if ((++polls & 8191u) == 0u) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    else std::this_thread::yield();
}
// What this thread is waiting for, readable by the thread that dumps the model's state when a guard expires (round 6): a wait that
// outlasts the guard must explain itself -- "a reader that was off its core" and "a wait cycle" look alike from one line of text,
// and different in (ring, position, the cell's word, who else waits for what).
struct ErRingWait {
    const char* what = nullptr;            // "put" / "get" / "bits" / ... ; nullptr = not waiting
    const void* cells = nullptr;           // which ring (its cell array)
    uint32_t pos = 0, seen = 0;            // the position waited for; the cell's word at the last poll
    std::chrono::steady_clock::time_point since;
};
inline thread_local ErRingWait er_ring_waiting;
#ifndef ER_RING_GUARD_MS
#define ER_RING_GUARD_MS 120000            // the host model's guard is wall-clock time: a poll count measures the machine's load, not the protocol
#endif
ER_RING_FN void er_ring_wait_begin(const char* what, const void* cells, uint32_t pos) {
    er_ring_waiting.cells = cells; er_ring_waiting.pos = pos; er_ring_waiting.seen = 0;
    er_ring_waiting.since = std::chrono::steady_clock::now();
    __atomic_store_n(&er_ring_waiting.what, what, __ATOMIC_RELEASE);
}
ER_RING_FN void er_ring_wait_end() { __atomic_store_n(&er_ring_waiting.what, (const char*)nullptr, __ATOMIC_RELEASE); }
ER_RING_FN bool er_ring_expired(uint32_t& guard, uint32_t seen) {
    __atomic_store_n(&er_ring_waiting.seen, seen, __ATOMIC_RELAXED);
    if ((++guard & 1023u) != 0u) return false;
    return std::chrono::steady_clock::now() - er_ring_waiting.since > std::chrono::milliseconds(ER_RING_GUARD_MS);
}
#else
#include <hip/hip_runtime.h>
#define ER_RING_FN __device__ __forceinline__
// LDS words: volatile accesses are single ds_read / ds_write instructions, LDS operations of one wave execute in order and
// all waves of the workgroup see one LDS.  The pointers are cast to the LDS address space explicitly: a volatile access through
// a generic pointer is compiled to flat_load / flat_store (the address-space inference leaves volatile accesses alone), which
// take the vector-memory path to LDS and make every later wait of the wave a full one.
typedef __attribute__((address_space(3))) uint32_t er_lds_u32;
ER_RING_FN uint32_t er_ring_load(const uint32_t* p) { return *(const volatile er_lds_u32*)p; }
ER_RING_FN void er_ring_store(uint32_t* p, uint32_t v) { *(volatile er_lds_u32*)p = v; }
ER_RING_FN uint32_t er_ring_add(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
ER_RING_FN uint32_t er_ring_or(uint32_t* p, uint32_t v) { return atomicOr(p, v); }
ER_RING_FN uint32_t er_ring_and(uint32_t* p, uint32_t v) { return atomicAnd(p, v); }
ER_RING_FN uint32_t er_ring_cas(uint32_t* p, uint32_t expect, uint32_t desired) { return atomicCAS(p, expect, desired); }
typedef __attribute__((address_space(3))) unsigned long long er_lds_u64;
ER_RING_FN unsigned long long er_ring_load64(const unsigned long long* p) { return *(const volatile er_lds_u64*)p; }
ER_RING_FN unsigned long long er_ring_add64(unsigned long long* p, unsigned long long v) { return atomicAdd(p, v); }
ER_RING_FN unsigned long long er_ring_cas64(unsigned long long* p, unsigned long long expect, unsigned long long desired) { return atomicCAS(p, expect, desired); }
ER_RING_FN void er_ring_pause() { __builtin_amdgcn_s_sleep(1); }
ER_RING_FN void er_ring_wait_begin(const char*, const void*, uint32_t) {}
ER_RING_FN void er_ring_wait_end() {}
#endif

#define ER_RING_PAYLOAD_BITS 13
#define ER_RING_PAYLOAD_MASK ((1u << ER_RING_PAYLOAD_BITS) - 1u)
#define ER_RING_FULL (1u << ER_RING_PAYLOAD_BITS)
#define ER_RING_LAP_SHIFT (ER_RING_PAYLOAD_BITS + 1)
// control words of a ring (an 8-byte aligned array of four): TAIL, a pad, then COUNT and HEAD as the low and high half of ONE
// 64-bit word, so that a grant is a single compare-and-swap (count - g, head + g) instead of a CAS on the count and an add on
// the head -- one LDS round trip less for every wave that takes entries
enum { ER_RING_TAIL = 0, ER_RING_COUNT = 2, ER_RING_HEAD = 3, ER_RING_WORDS = 4 };
// A wait that outlasts this many polls means the protocol itself is broken (a writer or reader that never comes): the
// caller raises the launch's status word instead of hanging.  Far longer than any wait a correct run can see.
#ifndef ER_RING_GUARD
#define ER_RING_GUARD (1u << 22)
#endif
#ifndef ER_RING_HOST_MODEL
ER_RING_FN bool er_ring_expired(uint32_t& guard, uint32_t) { return ++guard >= ER_RING_GUARD; }
#endif

ER_RING_FN uint32_t er_ring_lap(uint32_t pos, uint32_t cap_log2) { return (pos >> cap_log2) << ER_RING_LAP_SHIFT; }

// ---- per lane ----
// producer of position `pos` (reserved for this lane): false = guard expired (status word, never in a correct run)
ER_RING_FN bool er_ring_put(uint32_t* cells, uint32_t cap_log2, uint32_t pos, uint32_t payload) {
    uint32_t* cell = cells + (pos & ((1u << cap_log2) - 1u));
    const uint32_t empty = er_ring_lap(pos, cap_log2);
    uint32_t guard = 0, v;
    if ((v = er_ring_load(cell)) != empty) {     // the previous lap's entry is still being read
        er_ring_wait_begin("put: the previous lap's reader of the cell", cells, pos);
        do {
            if (er_ring_expired(guard, v)) return false;      // (the wait word stays: the dump shows what was waited for)
            er_ring_pause();
        } while ((v = er_ring_load(cell)) != empty);
        er_ring_wait_end();
    }
    er_ring_store(cell, empty | ER_RING_FULL | (payload & ER_RING_PAYLOAD_MASK));
    return true;
}
// consumer of position `pos` (granted to this lane)
ER_RING_FN bool er_ring_get(uint32_t* cells, uint32_t cap_log2, uint32_t pos, uint32_t& payload) {
    uint32_t* cell = cells + (pos & ((1u << cap_log2) - 1u));
    const uint32_t full = er_ring_lap(pos, cap_log2) | ER_RING_FULL;
    uint32_t v, guard = 0;
    if (((v = er_ring_load(cell)) & ~ER_RING_PAYLOAD_MASK) != full) {         // its writer is on its way
        er_ring_wait_begin("get: the writer of the cell", cells, pos);
        do {
            if (er_ring_expired(guard, v)) { payload = 0; return false; }
            er_ring_pause();
        } while (((v = er_ring_load(cell)) & ~ER_RING_PAYLOAD_MASK) != full);
        er_ring_wait_end();
    }
    payload = v & ER_RING_PAYLOAD_MASK;
    er_ring_store(cell, er_ring_lap(pos + (1u << cap_log2), cap_log2));        // (lap + 1, empty): free for the next lap's writer
    return true;
}

// ---- per wave (one lane calls; the caller broadcasts the result) ----
ER_RING_FN uint32_t er_ring_reserve(uint32_t* ctl, uint32_t n) { return er_ring_add(&ctl[ER_RING_TAIL], n); }
ER_RING_FN void er_ring_publish(uint32_t* ctl, uint32_t n) { er_ring_add64((unsigned long long*)&ctl[ER_RING_COUNT], (unsigned long long)n); }   // (COUNT <= capacity: no carry into HEAD)
// what a wave reads to decide whether a take is worth trying: (COUNT, HEAD) in one load; the same value may be handed to
// er_ring_grant as its first guess
ER_RING_FN unsigned long long er_ring_peek(const uint32_t* ctl) { return er_ring_load64((const unsigned long long*)&ctl[ER_RING_COUNT]); }
ER_RING_FN uint32_t er_ring_peek_count(unsigned long long peek) { return (uint32_t)peek; }
// grants min(want, COUNT) entries; `base` = the first granted position.  One compare-and-swap on (COUNT, HEAD) keeps COUNT exact
// at every instant (a subtract-then-restore lets it dip below zero while several waves ask at once: the fault of the second
// protocol of round 2) and hands out the positions with it.
ER_RING_FN uint32_t er_ring_grant(uint32_t* ctl, uint32_t want, uint32_t& base, unsigned long long seen) {
    unsigned long long* word = (unsigned long long*)&ctl[ER_RING_COUNT];
    base = 0;
    while (true) {
        const uint32_t count = (uint32_t)seen, head = (uint32_t)(seen >> 32);
        const uint32_t granted = count < want ? count : want;
        if (granted == 0 || count > 0x7fffffffu) return 0;
        const unsigned long long next = (unsigned long long)(count - granted) | ((unsigned long long)(uint32_t)(head + granted) << 32);
        const unsigned long long found = er_ring_cas64(word, seen, next);
        if (found == seen) { base = head; return granted; }
        seen = found;
    }
}
ER_RING_FN uint32_t er_ring_grant(uint32_t* ctl, uint32_t want, uint32_t& base) { return er_ring_grant(ctl, want, base, er_ring_peek(ctl)); }

// ---- one "occupied" bit per cell of a ring whose cells live elsewhere (the HBM pixel ring) ----
// producer, before it writes cell `idx`: waits while the previous lap's entry of that cell has not been read
ER_RING_FN bool er_bits_acquire(uint32_t* bits, uint32_t idx) {
    const uint32_t m = 1u << (idx & 31u);
    uint32_t guard = 0;
    if (er_ring_or(&bits[idx >> 5], m) & m) {
        er_ring_wait_begin("bits: the previous lap's reader of the pixel cell", bits, idx);
        do {
            if (er_ring_expired(guard, m)) return false;
            er_ring_pause();
        } while (er_ring_or(&bits[idx >> 5], m) & m);
        er_ring_wait_end();
    }
    return true;
}
// consumer, after it has read cell `idx`
ER_RING_FN void er_bits_release(uint32_t* bits, uint32_t idx) { er_ring_and(&bits[idx >> 5], ~(1u << (idx & 31u))); }
