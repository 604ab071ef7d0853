// Microbenchmark (round 4): does the issue rate of v_mfma_f32_32x32x16_f16 depend on WHICH registers hold its A / B operands
// (VGPR bank conflicts between srcA and srcB, B in AGPRs, accumulators in VGPRs)?  One wave per SIMD; per iteration 8 MFMAs on four
// rotating accumulators with explicit physical registers.     mfma_banks [iters]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define STR2(x) #x
#define STR(x) STR2(x)
// A at v[AB:AB+3], B at v[BB:BB+3] (or a[BB..] when BAGPR), accumulators a[0:63] (or v[64:127] when CV)
#define BODY(A, B, ACC0, ACC1, ACC2, ACC3)                                         \
    "v_mfma_f32_32x32x16_f16 " ACC0 ", " A ", " B ", " ACC0 "\n\t"                 \
    "v_mfma_f32_32x32x16_f16 " ACC1 ", " A ", " B ", " ACC1 "\n\t"                 \
    "v_mfma_f32_32x32x16_f16 " ACC2 ", " A ", " B ", " ACC2 "\n\t"                 \
    "v_mfma_f32_32x32x16_f16 " ACC3 ", " A ", " B ", " ACC3 "\n\t"                 \
    "v_mfma_f32_32x32x16_f16 " ACC0 ", " A ", " B ", " ACC0 "\n\t"                 \
    "v_mfma_f32_32x32x16_f16 " ACC1 ", " A ", " B ", " ACC1 "\n\t"                 \
    "v_mfma_f32_32x32x16_f16 " ACC2 ", " A ", " B ", " ACC2 "\n\t"                 \
    "v_mfma_f32_32x32x16_f16 " ACC3 ", " A ", " B ", " ACC3 "\n\t"
#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27", \
    "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
    "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63", \
    "a64","a65","a66","a67","a68","a69","a70","a71"

template <int KIND>
__global__ __launch_bounds__(256, 1) void k(int iters, unsigned long long* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // operands: v8..v27 and a64..a71 = 1.0 (fp16 pairs)
    asm volatile("v_mov_b32 v8, 0x3c003c00\n\tv_mov_b32 v9, v8\n\tv_mov_b32 v10, v8\n\tv_mov_b32 v11, v8\n\tv_mov_b32 v12, v8\n\tv_mov_b32 v13, v8\n\tv_mov_b32 v14, v8\n\tv_mov_b32 v15, v8\n\t"
                 "v_mov_b32 v16, v8\n\tv_mov_b32 v17, v8\n\tv_mov_b32 v18, v8\n\tv_mov_b32 v19, v8\n\tv_mov_b32 v20, v8\n\tv_mov_b32 v21, v8\n\tv_mov_b32 v22, v8\n\tv_mov_b32 v23, v8\n\t"
                 "v_mov_b32 v24, v8\n\tv_mov_b32 v25, v8\n\tv_mov_b32 v26, v8\n\tv_mov_b32 v27, v8\n\t"
                 "v_accvgpr_write_b32 a64, v8\n\tv_accvgpr_write_b32 a65, v8\n\tv_accvgpr_write_b32 a66, v8\n\tv_accvgpr_write_b32 a67, v8\n\t"
                 "v_accvgpr_write_b32 a68, v8\n\tv_accvgpr_write_b32 a69, v8\n\tv_accvgpr_write_b32 a70, v8\n\tv_accvgpr_write_b32 a71, v8\n\ts_nop 4" ::: CLOB);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(BODY("v[8:11]", "v[12:15]", "a[0:15]", "a[16:31]", "a[32:47]", "a[48:63]") ::: CLOB);   // A = 0 mod 4, B = 0 mod 4

        if (KIND == 2) asm volatile(BODY("v[8:11]", "v[14:17]", "a[0:15]", "a[16:31]", "a[32:47]", "a[48:63]") ::: CLOB);   // B = 2 mod 4

        if (KIND == 4) asm volatile(BODY("v[10:13]", "v[14:17]", "a[0:15]", "a[16:31]", "a[32:47]", "a[48:63]") ::: CLOB);  // A = 2, B = 2
        if (KIND == 5) asm volatile(BODY("v[10:13]", "v[16:19]", "a[0:15]", "a[16:31]", "a[32:47]", "a[48:63]") ::: CLOB);  // A = 2, B = 0
        if (KIND == 6) asm volatile(BODY("v[8:11]", "a[64:67]", "a[0:15]", "a[16:31]", "a[32:47]", "a[48:63]") ::: CLOB);   // B in AGPRs
        if (KIND == 7) asm volatile(BODY("a[68:71]", "a[64:67]", "a[0:15]", "a[16:31]", "a[32:47]", "a[48:63]") ::: CLOB);  // A and B in AGPRs
        if (KIND == 8) asm volatile(BODY("v[8:11]", "v[8:11]", "a[0:15]", "a[16:31]", "a[32:47]", "a[48:63]") ::: CLOB);    // A = B (same registers)
        if (KIND == 9) asm volatile(BODY("v[8:11]", "v[12:15]", "a[0:15]", "a[0:15]", "a[0:15]", "a[0:15]") ::: CLOB);      // one accumulator: every MFMA depends on its predecessor
        if (KIND == 10) asm volatile(BODY("v[8:11]", "v[12:15]", "a[0:15]", "a[0:15]", "a[16:31]", "a[16:31]") ::: CLOB);   // dependent pairs
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int KIND> static void run(int iters, unsigned long long* out, const char* what) {
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<KIND>, dim3(64), dim3(256), 0, 0, iters, out);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", what); exit(2); }
    }
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), out, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-64s %6.2f cycles per MFMA\n", what, (double)h[128] / iters / 8.0);
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned long long* out;
    if (hipMalloc(&out, 256 * sizeof(unsigned long long)) != hipSuccess) return 1;
    run<0>(iters, out, "A v[8:11] (0 mod 4), B v[12:15] (0 mod 4)");

    run<2>(iters, out, "A v[8:11], B v[14:17] (2 mod 4)");

    run<4>(iters, out, "A v[10:13] (2 mod 4), B v[14:17] (2 mod 4)");
    run<5>(iters, out, "A v[10:13] (2 mod 4), B v[16:19] (0 mod 4)");
    run<6>(iters, out, "A v[8:11], B a[64:67] (AGPR)");
    run<7>(iters, out, "A a[68:71], B a[64:67] (both AGPR)");
    run<8>(iters, out, "A = B = v[8:11]");
    run<9>(iters, out, "one accumulator (every MFMA depends on the one before)");
    run<10>(iters, out, "dependent pairs (two accumulators)");
    return 0;
}
