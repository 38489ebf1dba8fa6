// Micro-benchmark: cycles per wave64 instruction per SIMD on gfx950 for the vector instructions of the traversal step
// (er_trav.h: v_cvt_f32_ubyteN, v_pk_fma_f32, v_max3 / v_min3, VOPC compares into SGPR pairs, v_cndmask on an SGPR pair, ...),
// at 1, 2 and 4 waves per SIMD.  Every instruction stream is 64 instructions per loop iteration over 8 independent register
// groups (dependency distance 8).  The table decides which formulation of a block is cheaper; instruction COUNT alone does not
// (v_pk_fma_f32 turned out to cost more than two v_fma_f32).
// Build: hipcc --offload-arch=gfx950 -O3 valu_cost_bench.hip -o valu_cost_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define R8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
// one instruction per accumulator, 8 accumulators; operand %8 = m, %9 = c (floats), pairs use a0a1.. as 64-bit
#define K_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define K_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define K_ADD(i) "v_add_f32 %" #i ", %" #i ", %9\n"
#define K_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define K_CVTUB(i) "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define K_CVTU32(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define K_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define K_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define K_CMP(i) "v_cmp_le_f32 s[20:21], %" #i ", %8\n"
#define K_CMPVCC(i) "v_cmp_le_f32 vcc, %" #i ", %8\n"
#define K_CNDS(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[20:21]\n"
#define K_CNDVCC(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define K_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define K_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 1, %8\n"
#define K_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define K_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define K_SAND(i) "s_and_b64 s[22:23], s[20:21], s[24:25]\n"
#define K_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define K_MEDF(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define K_SUBREV(i) "v_sub_f32 %" #i ", %8, %" #i "\n"
#define K_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define K_CND64VCC(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define K_CMPCNDVCC(i) "v_cmp_le_f32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define K_CMPCNDS(i) "v_cmp_le_f32 s[20:21], %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, s[20:21]\n"
#define K_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define K_MIN3(i) "v_min3_f32 %" #i ", %" #i ", %8, %9\n"
#define K_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define K_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define K_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %4, %5\n"
#define K_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %4\n"
#define K_ADD64(i) "v_add_f64 %" #i ", %" #i ", %5\n"
// packed: 4 register pairs, each used twice per 8 (dependency distance 4)
#define P_FMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %4, %5\n"
#define P_MUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %4\n"
#define P_ADD(i) "v_pk_add_f32 %" #i ", %" #i ", %5\n"
#define R4x2(T) T(0) T(1) T(2) T(3) T(0) T(1) T(2) T(3)

template <int KIND>
__global__ __launch_bounds__(256) void valu(int iters, float* out) {
    const unsigned lane = threadIdx.x & 63;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    const float m = 1.0000001f, c = 0.5f;
    double p0 = lane, p1 = lane + 1, p2 = lane + 2, p3 = lane + 3;
    const double pm = 1.0, pc = 0.5;
#define BODY(STR) asm volatile(STR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "s20", "s21", "s22", "s23", "s24", "s25", "vcc")
#define PBODY(STR) asm volatile(STR : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm), "v"(pc))
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (KIND == 0) BODY(R8(K_FMA));
            if (KIND == 1) BODY(R8(K_MUL));
            if (KIND == 2) BODY(R8(K_ADD));
            if (KIND == 3) BODY(R8(K_MAX3));
            if (KIND == 4) BODY(R8(K_CVTUB));
            if (KIND == 5) BODY(R8(K_CVTU32));
            if (KIND == 6) BODY(R8(K_AND));
            if (KIND == 7) BODY(R8(K_BFE));
            if (KIND == 8) BODY(R8(K_CMP));
            if (KIND == 9) BODY(R8(K_CMPVCC));
            if (KIND == 10) BODY(R8(K_CNDS));
            if (KIND == 11) BODY(R8(K_CNDVCC));
            if (KIND == 12) BODY(R8(K_RCP));
            if (KIND == 13) BODY(R8(K_LSHLOR));
            if (KIND == 14) BODY(R8(K_ALIGNBIT));
            if (KIND == 15) BODY(R8(K_MOV));
            if (KIND == 16) BODY(R8(K_SAND));
            if (KIND == 17) BODY(R8(K_PERM));
            if (KIND == 18) BODY(R8(K_MEDF));
            if (KIND == 19) BODY(R8(K_SUBREV));
            if (KIND == 20) BODY(R8(K_FMAC));
            if (KIND == 21) PBODY(R4x2(P_FMA));
            if (KIND == 22) PBODY(R4x2(P_MUL));
            if (KIND == 23) PBODY(R4x2(P_ADD));
            if (KIND == 25) BODY(R8(K_CND64VCC));
            if (KIND == 26) BODY(R8(K_CMPCNDVCC));
            if (KIND == 27) BODY(R8(K_CMPCNDS));
            if (KIND == 28) BODY(R8(K_MAX));
            if (KIND == 29) BODY(R8(K_MIN3));
            if (KIND == 30) BODY(R8(K_OR3));
            if (KIND == 31) BODY(R8(K_LSHL));
            if (KIND == 32) PBODY(R4x2(K_FMA64));
            if (KIND == 33) PBODY(R4x2(K_MUL64));
            if (KIND == 34) PBODY(R4x2(K_ADD64));
            if (KIND == 24) BODY(K_FMA(0) K_SAND(0) K_FMA(1) K_SAND(0) K_FMA(2) K_SAND(0) K_FMA(3) K_SAND(0) K_FMA(4) K_SAND(0) K_FMA(5) K_SAND(0) K_FMA(6) K_SAND(0) K_FMA(7) K_SAND(0));   // 8 VALU + 8 SALU interleaved
        }
    }
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(p0 + p1 + p2 + p3);
    if (s == 1.2345f) out[0] = s;
}

typedef void (*kern_t)(int, float*);
int main() {
    float* out; CHECK(hipMalloc(&out, 4));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const double ghz = prop.clockRate * 1e-6;
    const int iters = 2000;
    const char* names[35] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_max3_f32", "v_cvt_f32_ubyte1", "v_cvt_f32_u32", "v_and_b32", "v_bfe_u32", "v_cmp_le_f32 -> sgpr pair",
                             "v_cmp_le_f32 -> vcc", "v_cndmask_b32 (sgpr pair)", "v_cndmask_b32 (vcc)", "v_rcp_f32", "v_lshl_or_b32", "v_alignbit_b32", "v_mov_b32", "s_and_b64",
                             "v_perm_b32", "v_med3_f32", "v_sub_f32", "v_fmac_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "8 v_fma + 8 s_and interleaved (per pair)", "v_cndmask_b32_e64 (vcc)", "v_cmp vcc + v_cndmask vcc (per pair)", "v_cmp sgpr + v_cndmask sgpr (per pair)", "v_max_f32", "v_min3_f32", "v_or3_b32", "v_lshlrev_b32", "v_fma_f64", "v_mul_f64", "v_add_f64"};
    kern_t ks[35] = {valu<0>, valu<1>, valu<2>, valu<3>, valu<4>, valu<5>, valu<6>, valu<7>, valu<8>, valu<9>, valu<10>, valu<11>, valu<12>, valu<13>, valu<14>, valu<15>, valu<16>,
                     valu<17>, valu<18>, valu<19>, valu<20>, valu<21>, valu<22>, valu<23>, valu<24>, valu<25>, valu<26>, valu<27>, valu<28>, valu<29>, valu<30>, valu<31>, valu<32>, valu<33>, valu<34>};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-44s %10s %10s %10s   (cycles per wave64 instruction per SIMD at %.2f GHz nominal)\n", "instruction", "1 wave/SIMD", "2", "4", ghz);
    for (int kind = 0; kind < 35; kind++) {
        double cyc[3];
        for (int w = 0; w < 3; w++) {
            const int wps = 1 << w, blocks = prop.multiProcessorCount * wps;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                ks[kind]<<<blocks, 256>>>(iters, out);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            cyc[w] = best * 1e-3 * ghz * 1e9 / ((double)iters * 64 * wps);
        }
        printf("%-44s %10.2f %10.2f %10.2f\n", names[kind], cyc[0], cyc[1], cyc[2]);
    }
    return 0;
}
