// Micro-benchmark: does a wave's SECOND access to a cache line whose miss is still in flight hold up the CU's L1 for everybody?
// The traversal step fetches an 80-byte node as five dwordx4 loads of one or two lines (TCP_PENDING_STALL_CYCLES is 0.49 of the
// streaming kernel's cycles).  Dependent chains of random nodes, as in gather_bench `lane`, fetching per node
//   p1 : one 16-byte piece                      (one access per line)
//   p2 : two pieces, 64 bytes apart             (second access hits the pending line -- or the neighbouring one)
//   p5 : five pieces, the node                  (today)
//   p5s: five pieces, but pieces 1-4 only AFTER piece 0 has arrived (s_waitcnt between them)
// reported as nodes/s and 128-byte lines/s.  Build: hipcc --offload-arch=gfx950 -O3 pending_bench.hip -o pending_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t a) { a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16; return a; }

template <int MODE>
__global__ __launch_bounds__(64) void gather(const uint4* __restrict__ nodes, uint32_t n_nodes, int iters, uint32_t* out) {
    uint32_t gid = blockIdx.x * 64 + threadIdx.x, acc = 0;
    uint32_t idx = mix(gid + 1) % n_nodes;
    for (int i = 0; i < iters; i++) {
        const uint4* p = nodes + (size_t)idx * 5;
        uint4 a = {0, 0, 0, 0}, b = a, c = a, d = a, e = a;
        if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=&v"(a) : "v"(p) : "memory");
        if (MODE == 2) asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:64\n s_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(e) : "v"(p) : "memory");
        if (MODE == 5) asm volatile("global_load_dwordx4 %0, %5, off\n global_load_dwordx4 %1, %5, off offset:16\n global_load_dwordx4 %2, %5, off offset:32\n"
                                    "global_load_dwordx4 %3, %5, off offset:48\n global_load_dwordx4 %4, %5, off offset:64\n s_waitcnt vmcnt(0)"
                                    : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e) : "v"(p) : "memory");
        if (MODE == 6) asm volatile("global_load_dwordx4 %0, %5, off\n global_load_dwordx4 %4, %5, off offset:64\n s_waitcnt vmcnt(0)\n"
                                    "global_load_dwordx4 %1, %5, off offset:16\n global_load_dwordx4 %2, %5, off offset:32\n"
                                    "global_load_dwordx4 %3, %5, off offset:48\n s_waitcnt vmcnt(0)"
                                    : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e) : "v"(p) : "memory");
        uint32_t s = a.x ^ b.y ^ c.z ^ d.w ^ e.x;
        acc += s;
        idx = mix(s + idx + i) % n_nodes;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
    uint32_t n_nodes = argc > 1 ? (uint32_t)atol(argv[1]) : 1300000u;
    const int iters = 500;
    std::vector<uint32_t> h((size_t)n_nodes * 20);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
    uint4* d; uint32_t* out;
    CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMalloc(&out, 4));
    CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int modes[4] = {1, 2, 5, 6};
    const char* names[4] = {"p1", "p2", "p5", "p5s"};
    const double lines[4] = {1.0, 1.5, 1.5, 1.5};     // 80-byte stride: half the nodes straddle a line (p2's second piece is the straddling one)
    for (int w = 8; w <= 32; w *= 2)
        for (int m = 0; m < 4; m++) {
            int blocks = 256 * w;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                if (modes[m] == 1) gather<1><<<blocks, 64>>>(d, n_nodes, iters, out);
                if (modes[m] == 2) gather<2><<<blocks, 64>>>(d, n_nodes, iters, out);
                if (modes[m] == 5) gather<5><<<blocks, 64>>>(d, n_nodes, iters, out);
                if (modes[m] == 6) gather<6><<<blocks, 64>>>(d, n_nodes, iters, out);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double nodes_s = (double)blocks * 64 * iters / best * 1e-6;
            printf("nodes=%u waves/CU=%d %-4s %.3f ms  %.1f Gnode/s  ~%.0f G lines/s  %.2f us per round of a wave\n", n_nodes, w, names[m], best, nodes_s, nodes_s * lines[m], best * 1e3 / iters);
        }
    return 0;
}
