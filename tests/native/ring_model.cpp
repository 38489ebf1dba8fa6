// ring_model.cpp -- host-thread model of the streaming schedule's pixel-ring exchange (csrc/er_stream.hip: st_take, lap-tagged
// cells, "put the pixel back, then take one"), run under contention on CPU threads.  Each "slot" thread repeatedly finishes a
// sample of its pixel, puts the pixel back (unless that was its last sample) and takes the next one; a slot that gets none
// retires.  Checked: every pixel receives exactly n samples, no pixel is ever held by two slots, all slots retire.
// Built and run by tests/test_stream_protocol_cpu.py with g++ -O2 -pthread (optionally -fsanitize=thread).
//   ring_model <slots> <pixels> <samples> <variant>      variant 0 = the protocol in use (compare-and-swap take),
//                                                        variant 1 = subtract-then-restore take (the fault of the first version)
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static std::atomic<uint32_t> g_head{0}, g_tail{0};
static std::atomic<int32_t> g_count{0};
static std::vector<std::atomic<uint64_t>> g_ring;
static uint32_t g_cap = 0;
static std::vector<std::atomic<int>> g_owner;     // pixel -> number of slots holding it (must stay <= 1)
static std::vector<std::atomic<uint32_t>> g_done; // pixel -> samples finished
static std::atomic<int> g_errors{0};

static uint32_t lap_tag(uint32_t pos) { return (((pos / g_cap) & 0x7Fu) + 1u) << 24; }

static bool take(int variant, uint32_t& pos) {
    if (variant == 0) {
        int32_t seen = g_count.load(std::memory_order_relaxed);
        while (true) {
            if (seen <= 0) return false;
            if (g_count.compare_exchange_weak(seen, seen - 1, std::memory_order_acq_rel)) break;
        }
    } else {
        const int32_t old = g_count.fetch_sub(1, std::memory_order_acq_rel);
        if (old <= 0) {
            std::this_thread::yield();                       // (widens the window of the dip, as a busy GPU does)
            g_count.fetch_add(1, std::memory_order_acq_rel);
            return false;
        }
    }
    pos = g_head.fetch_add(1, std::memory_order_acq_rel);
    return true;
}

static void slot_thread(uint32_t pixel, uint32_t left, int variant) {
    bool have = true;
    while (have) {
        if (g_owner[pixel].fetch_add(1) != 0) g_errors++;      // a second holder
        g_done[pixel].fetch_add(1);                            // "run one sample"
        g_owner[pixel].fetch_sub(1);
        left--;
        if (left > 0) {                                        // put the pixel back: cell first, then the count
            const uint32_t pos = g_tail.fetch_add(1, std::memory_order_acq_rel);
            g_ring[pos % g_cap].store(((uint64_t)(left | lap_tag(pos)) << 32) | pixel, std::memory_order_release);
            g_count.fetch_add(1, std::memory_order_release);
        }
        uint32_t pos = 0;
        if (!take(variant, pos)) { have = false; break; }      // retire
        const uint32_t want = lap_tag(pos);
        uint64_t w;
        while ((((uint32_t)((w = g_ring[pos % g_cap].load(std::memory_order_acquire)) >> 32)) & 0xFF000000u) != want) std::this_thread::yield();
        pixel = (uint32_t)w;
        left = (uint32_t)(w >> 32) & 0x00FFFFFFu;
    }
}

int main(int argc, char** argv) {
    const uint32_t slots = argc > 1 ? (uint32_t)atoi(argv[1]) : 8, pixels = argc > 2 ? (uint32_t)atoi(argv[2]) : 8;
    const uint32_t samples = argc > 3 ? (uint32_t)atoi(argv[3]) : 50;
    const int variant = argc > 4 ? atoi(argv[4]) : 0;
    // (a ring long enough that no position comes round again within the run: a host thread can be descheduled for
    // milliseconds between taking a position and reading its cell, which a wave cannot; the lap tags are exercised on the GPU)
    g_cap = 1u << 15;
    while (g_cap < pixels * samples + pixels) g_cap <<= 1;
    g_ring = std::vector<std::atomic<uint64_t>>(g_cap);
    for (auto& c : g_ring) c.store(0);
    g_owner = std::vector<std::atomic<int>>(pixels);
    g_done = std::vector<std::atomic<uint32_t>>(pixels);
    for (uint32_t p = 0; p < pixels; p++) { g_owner[p].store(0); g_done[p].store(0); }
    const uint32_t in_slots = slots < pixels ? slots : pixels;
    for (uint32_t p = in_slots; p < pixels; p++) {             // the pixels that do not start in a slot wait in the ring
        const uint32_t pos = p - in_slots;
        g_ring[pos % g_cap].store(((uint64_t)(samples | lap_tag(pos)) << 32) | p);
    }
    g_tail.store(pixels - in_slots);
    g_count.store((int32_t)(pixels - in_slots));
    std::vector<std::thread> th;
    for (uint32_t s = 0; s < in_slots; s++) th.emplace_back(slot_thread, s, samples, variant);
    for (auto& t : th) t.join();
    uint32_t short_px = 0;
    for (uint32_t p = 0; p < pixels; p++) if (g_done[p].load() != samples) short_px++;
    printf("slots %u pixels %u samples %u variant %d: %u pixels short, %d double holders, %d left in the ring\n", in_slots, pixels, samples, variant, short_px,
           g_errors.load(), (int)g_count.load());
    return (short_px || g_errors.load() || g_count.load() != 0) ? 1 : 0;
}
