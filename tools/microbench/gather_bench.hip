// Micro-benchmark behind DESIGN.md's trace-kernel analysis: how fast can a CU pull randomly addressed
// 80-byte BVH8 nodes out of a large buffer, as a function of how the 16-byte pieces map onto lanes?
//   lane   : every lane owns a ray and issues five dwordx4 loads for its own node (er_trav.h today)
//   coop8  : eight adjacent lanes own one ray; lanes 0-4 load one piece each (one load instruction)
//   coop4  : four adjacent lanes own one ray; piece k per lane + one more load for the fifth piece
//   top3   : like `lane`, but the chain is a sequence of 21-hop "rays" (the C2 soup's mean node visits per ray) whose
//            first three hops address the 73 nodes of BVH8 levels 0-2; all 21 hops from global memory (the top
//            nodes stay in L1 / L2 by themselves)
//   top3lds: the same, the 73 top nodes staged once per workgroup (256 threads) in LDS and read with ds_read_b128
//            (north_star: "top BVH levels staged in LDS") -- what the staging can buy at best, without the
//            instruction cost of selecting between the two sources inside the traversal step
// Each chain is dependent (the next node index is derived from loaded data), as in traversal.
// Build: hipcc --offload-arch=gfx950 -O3 gather_bench.hip -o gather_bench ; run: ./gather_bench [nodes] [waves_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t a) { a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16; return a; }

template <int MODE>
__global__ __launch_bounds__(256) void gather(const uint4* __restrict__ nodes, uint32_t n_nodes, int iters, uint32_t* out) {
    const uint32_t lane = threadIdx.x;
    uint32_t gid = blockIdx.x * 64 + lane;
    uint32_t acc = 0;
    if (MODE == 0) {
        uint32_t idx = mix(gid + 1) % n_nodes;
        for (int i = 0; i < iters; i++) {
            const uint4* p = nodes + (size_t)idx * 5;
            uint4 a, b, c, d, e;
            asm volatile("global_load_dwordx4 %0, %5, off\n global_load_dwordx4 %1, %5, off offset:16\n"
                         "global_load_dwordx4 %2, %5, off offset:32\n global_load_dwordx4 %3, %5, off offset:48\n"
                         "global_load_dwordx4 %4, %5, off offset:64\n s_waitcnt vmcnt(0)"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e) : "v"(p) : "memory");
            uint32_t s = a.x ^ b.y ^ c.z ^ d.w ^ e.x;
            acc += s;
            idx = mix(s + idx + i) % n_nodes;
        }
    } else if (MODE == 3 || MODE == 4) {
        __shared__ uint4 s_top[73 * 5];
        if (MODE == 4) {
            for (uint32_t k = threadIdx.x; k < 73 * 5; k += blockDim.x) s_top[k] = nodes[k];
            __syncthreads();
        }
        uint32_t idx = 0;
        for (int i = 0; i < iters; i++) {
            const int hop = i % 21;
            uint4 a, b, c, d, e;
            if (MODE == 4 && hop < 3) {
                const uint4* p = s_top + idx * 5;
                a = p[0]; b = p[1]; c = p[2]; d = p[3]; e = p[4];
            } else {
                const uint4* p = nodes + (size_t)idx * 5;
                asm volatile("global_load_dwordx4 %0, %5, off\n global_load_dwordx4 %1, %5, off offset:16\n"
                             "global_load_dwordx4 %2, %5, off offset:32\n global_load_dwordx4 %3, %5, off offset:48\n"
                             "global_load_dwordx4 %4, %5, off offset:64\n s_waitcnt vmcnt(0)"
                             : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e) : "v"(p) : "memory");
            }
            uint32_t s = a.x ^ b.y ^ c.z ^ d.w ^ e.x;
            acc += s;
            const uint32_t r = mix(s + idx + i + gid);
            // next hop: root (0) at the start of a ray, then level 1 (1..8), level 2 (9..72), then anywhere
            idx = hop == 20 ? 0u : (hop == 0 ? 1u + (r & 7u) : (hop == 1 ? 9u + (r & 63u) : r % n_nodes));
        }
    } else if (MODE == 1) {
        const uint32_t grp = gid >> 3, sub = lane & 7;
        uint32_t idx = mix(grp + 1) % n_nodes;
        for (int i = 0; i < iters; i++) {
            const uint4* p = nodes + (size_t)idx * 5 + (sub < 5 ? sub : 0);
            uint4 a;
            asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=&v"(a) : "v"(p) : "memory");
            uint32_t s = a.x ^ a.y;
            // group-wide combine (3 xor-shuffles), as a stand-in for the child-slot reduction
            s ^= __shfl_xor(s, 1); s ^= __shfl_xor(s, 2); s ^= __shfl_xor(s, 4);
            acc += s;
            idx = mix(s + idx + i) % n_nodes;
        }
    } else {
        const uint32_t grp = gid >> 2, sub = lane & 3;
        uint32_t idx = mix(grp + 1) % n_nodes;
        for (int i = 0; i < iters; i++) {
            const uint4* p = nodes + (size_t)idx * 5;
            const uint4* q = p + sub;
            uint4 a, b;
            asm volatile("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %3, off offset:64\n s_waitcnt vmcnt(0)"
                         : "=&v"(a), "=&v"(b) : "v"(q), "v"(p) : "memory");
            uint32_t s = a.x ^ a.y ^ b.z;
            s ^= __shfl_xor(s, 1); s ^= __shfl_xor(s, 2);
            acc += s;
            idx = mix(s + idx + i) % n_nodes;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;   // keep the loads alive
}

int main(int argc, char** argv) {
    uint32_t n_nodes = argc > 1 ? (uint32_t)atol(argv[1]) : 1300000u;
    int wpc = argc > 2 ? atoi(argv[2]) : 16;
    const int iters = 504;   // a multiple of 21
    std::vector<uint32_t> h((size_t)n_nodes * 20);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
    uint4* d; uint32_t* out;
    CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMalloc(&out, 4));
    CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[5] = {"lane", "coop8", "coop4", "top3", "top3lds"};
    const int rays_per_wave[5] = {64, 8, 16, 64, 64};
    for (int w = wpc; w <= 32; w *= 2) {
        for (int mode = 0; mode < 5; mode++) {
            int blocks = 256 * w;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) gather<0><<<blocks, 64>>>(d, n_nodes, iters, out);
                if (mode == 1) gather<1><<<blocks, 64>>>(d, n_nodes, iters, out);
                if (mode == 2) gather<2><<<blocks, 64>>>(d, n_nodes, iters, out);
                if (mode == 3) gather<3><<<blocks / 4, 256>>>(d, n_nodes, iters, out);
                if (mode == 4) gather<4><<<blocks / 4, 256>>>(d, n_nodes, iters, out);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            double fetches = (double)blocks * rays_per_wave[mode] * iters;
            printf("nodes=%u waves/CU=%d %-6s %.3f ms  %.2f Gnode/s  %.0f GB/s algorithmic\n", n_nodes, w, names[mode], best,
                   fetches / best * 1e-6, fetches * 80 / best * 1e-6);
        }
    }
    return 0;
}
