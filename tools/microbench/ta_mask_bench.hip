// Micro-benchmark: what does a vector-memory instruction cost the CU's address path (TA/TCP) as a function of its EXEC mask?
// er_trav.h's fetch issues all eleven dwordx4 loads of a step for all 64 lanes, parking the lanes that do not need a piece on one
// shared address.  If the address unit skips inactive lanes (or whole inactive quads), masking those loads instead would shorten
// its busy time (TA_TA_BUSY is 0.72 of the streaming kernel's cycles, profiles/r03_pmc_ta_kernel.txt).
//   all64   : 64 lanes, every lane its own random node (5 dwordx4)
//   park48  : 16 lanes (contiguous) their own node, 48 lanes read node 0 (today's "dummy address")
//   mask16c : only lanes 0-15 execute the loads (EXEC-masked), contiguous
//   mask16q : only lane 0 of every quad executes the loads (16 lanes, every quad touched)
//   park48q : lane 0 of every quad its own node, the other three read node 0
// Build: hipcc --offload-arch=gfx950 -O3 ta_mask_bench.hip -o ta_mask_bench ; run: ./ta_mask_bench [nodes] [waves_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t a) { a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16; return a; }

template <int MODE>
__global__ __launch_bounds__(64) void gather(const uint4* __restrict__ nodes, uint32_t n_nodes, int iters, uint32_t* out) {
    const uint32_t lane = threadIdx.x;
    uint32_t gid = blockIdx.x * 64 + lane;
    uint32_t acc = 0;
    const bool real = MODE == 0 ? true : (MODE == 1 || MODE == 2) ? lane < 16 : (lane & 3) == 0;
    const bool masked = MODE == 2 || MODE == 3;
    uint32_t idx = mix(gid + 1) % n_nodes;
    if (!masked || real) {
        for (int i = 0; i < iters; i++) {
            const uint4* p = nodes + (size_t)(real ? idx : 0u) * 5;
            uint4 a, b, c, d, e;
            asm volatile("global_load_dwordx4 %0, %5, off\n global_load_dwordx4 %1, %5, off offset:16\n"
                         "global_load_dwordx4 %2, %5, off offset:32\n global_load_dwordx4 %3, %5, off offset:48\n"
                         "global_load_dwordx4 %4, %5, off offset:64\n s_waitcnt vmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e) : "v"(p) : "memory");
            uint32_t s = a.x ^ b.y ^ c.z ^ d.w ^ e.x;
            acc += s;
            idx = mix(s + idx + i) % n_nodes;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
    uint32_t n_nodes = argc > 1 ? (uint32_t)atol(argv[1]) : 1300000u;
    int w = argc > 2 ? atoi(argv[2]) : 16;
    const int iters = 500;
    std::vector<uint32_t> h((size_t)n_nodes * 20);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
    uint4* d; uint32_t* out;
    CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMalloc(&out, 4));
    CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[5] = {"all64", "park48", "mask16c", "mask16q", "park48q"};
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const double ghz = prop.clockRate * 1e-6;
    for (int mode = 0; mode < 5; mode++) {
        int blocks = 256 * w;
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) gather<0><<<blocks, 64>>>(d, n_nodes, iters, out);
            if (mode == 1) gather<1><<<blocks, 64>>>(d, n_nodes, iters, out);
            if (mode == 2) gather<2><<<blocks, 64>>>(d, n_nodes, iters, out);
            if (mode == 3) gather<3><<<blocks, 64>>>(d, n_nodes, iters, out);
            if (mode == 4) gather<4><<<blocks, 64>>>(d, n_nodes, iters, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double instr_per_cu = (double)w * iters * 5;
        printf("nodes=%u waves/CU=%d %-8s %.3f ms  %.1f cycles per wave-instruction per CU (at %.2f GHz)  %.2f G real nodes/s\n", n_nodes, w, names[mode], best,
               best * 1e-3 * ghz * 1e9 / instr_per_cu, ghz, (double)blocks * (mode == 0 ? 64 : 16) * iters / best * 1e-6);
    }
    return 0;
}
