// Micro-benchmark: does a wave64 vector-ALU instruction cost less when one 32-lane half of EXEC is empty?
// gfx950's SIMDs are 32 lanes wide and a wave64 instruction takes two passes.  If the hardware skips the pass of an empty half,
// a traversal step whose active lanes are compacted into one half would cost half the vector issue time -- and two half-waves of
// rays could run out of phase in one wave (one half computing while the other half's loads are in flight) at no extra vector cost.
//   all64   : every lane active
//   low32   : lanes 0-31 active (EXEC[63:32] = 0)
//   high32  : lanes 32-63 active
//   even32  : even lanes active (both halves half full)
//   low16   : lanes 0-15
// Each wave runs ITERS x 64 independent v_fma_f32 (8 accumulators) / v_pk_fma_f32 / v_cvt_f32_ubyte0 + v_max3_f32 mixes.
// Build: hipcc --offload-arch=gfx950 -O3 valu_mask_bench.hip -o valu_mask_bench ; run: ./valu_mask_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE, int KIND>
__global__ __launch_bounds__(256) void valu(int iters, float* out) {
    const unsigned lane = threadIdx.x & 63;
    const bool on = MODE == 0 ? true : MODE == 1 ? lane < 32 : MODE == 2 ? lane >= 32 : MODE == 3 ? (lane & 1) == 0 : lane < 16;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    const float m = 1.0000001f, c = 0.5f;
    if (on) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (KIND == 0)
                    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
                else
                    asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                                 "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                                 : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(const double*)&m), "v"(*(const double*)&c));
            }
        }
    }
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 1.2345f) out[0] = s;
}

template <int MODE, int KIND>
float run(int blocks, int iters, float* out) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        valu<MODE, KIND><<<blocks, 256>>>(iters, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 4));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const double ghz = prop.clockRate * 1e-6;
    const int iters = 4000;
    const char* names[5] = {"all64", "low32", "high32", "even32", "low16"};
    for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD: blocks of 256 threads = one wave per SIMD each
        const int blocks = prop.multiProcessorCount * wps;
        for (int kind = 0; kind < 2; kind++) {
            float t[5];
            t[0] = kind ? run<0, 1>(blocks, iters, out) : run<0, 0>(blocks, iters, out);
            t[1] = kind ? run<1, 1>(blocks, iters, out) : run<1, 0>(blocks, iters, out);
            t[2] = kind ? run<2, 1>(blocks, iters, out) : run<2, 0>(blocks, iters, out);
            t[3] = kind ? run<3, 1>(blocks, iters, out) : run<3, 0>(blocks, iters, out);
            t[4] = kind ? run<4, 1>(blocks, iters, out) : run<4, 0>(blocks, iters, out);
            for (int m = 0; m < 5; m++)
                printf("waves/SIMD=%d %-12s %-7s %.3f ms  %.2f cycles per wave-instruction per SIMD (at %.2f GHz nominal)\n", wps, kind ? "v_pk_fma_f32" : "v_fma_f32", names[m], t[m],
                       t[m] * 1e-3 * ghz * 1e9 / ((double)iters * 64 * wps), ghz);
        }
    }
    return 0;
}
