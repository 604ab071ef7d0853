// Microbenchmark: do packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) overlap with a running MFMA of the same
// wave (one wave per SIMD), or do they wait for the matrix pipe?  hipcc's post-RA "unpack" peephole splits packed ops that sit in an
// MFMA's shadow into two plain ones, which suggests the latter.  Per iteration: 3 independent v_mfma_f32_32x32x16_f16 and, behind each,
// N ops of one kind.     pk_mfma [iters]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND 6..11: other epilogue instructions (see below).  KIND 0: nothing   1: N x v_mul_f32   2: N x v_pk_mul_f32   3: 2N x v_mul_f32 (the unpacked equivalent of 2)   4: N x v_pk_add_f32   5: N x v_pk_fma_f32
template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void k(int iters, unsigned long long* out, float* sink) {
    extern __shared__ unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 a0 = {}, a1 = {}, a2 = {};
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(lane * 0.001f + i); y[i] = (_Float16)(0.5f + i * 0.01f); }
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{1.0f + lane * 1e-3f, 1.0f + i * 1e-3f};
    const f32x2 c = {1.0000001f, 0.9999999f};
    float ag = 1.0f + lane;
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = 0x3c003c00u + lane + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            if (m == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
            if (m == 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
            if (m == 2) a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a2, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                f32x2& r = v[i & 7];
                if (KIND == 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[0]) : "v"(c[0]));
                if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(r) : "v"(c));
                if (KIND == 3) asm volatile("v_mul_f32 %0, %2, %0\n\tv_mul_f32 %1, %3, %1" : "+v"(r[0]), "+v"(r[1]) : "v"(c[0]), "v"(c[1]));
                if (KIND == 4) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(r) : "v"(c));
                if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(r) : "v"(c));
                // the other instructions of the chain kernels' epilogues (one each; plain 32-bit VALU unless noted)
                if (KIND == 6) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i & 7]) : "v"(r[0]), "v"(r[1]));
                if (KIND == 7) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r[0]) : "v"(u[i & 7]));
                if (KIND == 8) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(r[0]) : "v"(u[i & 7]), "v"(u[(i + 1) & 7]));      // VOP3P
                if (KIND == 9) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r[0]) : "a"(ag));      // ("a": the compiler allocates the AGPR)
                if (KIND == 10) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i & 7]) : "v"(r[0]));
                if (KIND == 11) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(c[0]), "v"(c[1]));
                // what v_cvt_pk_f16_f32 could be replaced by: two single conversions (the second into the high half) [+ a pack]
                if (KIND == 12) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u[i & 7]) : "v"(r[0]));
                if (KIND == 13) asm volatile("v_cvt_f16_f32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(u[i & 7]) : "v"(r[1]));
                if (KIND == 14) asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(u[i & 7]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (KIND == 15) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r[0]) : "v"(u[i & 7]));
                if (KIND == 16) asm volatile("v_max_i32 %0, 0, %0" : "+v"(u[i & 7]));
                if (KIND == 17) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[0]) : "v"(r[1]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i];
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1] + (float)u[i];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int KIND, int N>
static void run(int iters, unsigned long long* out, float* sink, const char* what) {
    (void)hipFuncSetAttribute((const void*)k<KIND, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<KIND, N>), dim3(256), dim3(256), 100 * 1024, 0, iters, out, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); exit(2); }
    }
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), out, sizeof(unsigned long long) * 1024, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-44s N = %d per MFMA: %7.1f cycles per iteration (3 MFMA = 96)\n", what, N, (double)h[512] / iters);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4096;
    unsigned long long* out; float* sink;
    if (hipMalloc(&out, 1024 * sizeof(unsigned long long)) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    run<0, 1>(iters, out, sink, "MFMAs only");
    run<1, 2>(iters, out, sink, "v_mul_f32");
    run<2, 2>(iters, out, sink, "v_pk_mul_f32");
    run<3, 2>(iters, out, sink, "2 x v_mul_f32 (unpacked pair)");
    run<4, 2>(iters, out, sink, "v_pk_add_f32");
    run<5, 2>(iters, out, sink, "v_pk_fma_f32");
    run<1, 4>(iters, out, sink, "v_mul_f32");
    run<2, 4>(iters, out, sink, "v_pk_mul_f32");
    run<3, 4>(iters, out, sink, "2 x v_mul_f32 (unpacked pair)");
    run<4, 4>(iters, out, sink, "v_pk_add_f32");
    run<5, 4>(iters, out, sink, "v_pk_fma_f32");
    run<1, 6>(iters, out, sink, "v_mul_f32");
    run<2, 6>(iters, out, sink, "v_pk_mul_f32");
    run<2, 1>(iters, out, sink, "v_pk_mul_f32");
    run<6, 4>(iters, out, sink, "v_cvt_pk_f16_f32");
    run<7, 4>(iters, out, sink, "v_cvt_f32_f16");
    run<8, 4>(iters, out, sink, "v_dot2_f32_f16");
    run<9, 4>(iters, out, sink, "v_accvgpr_read_b32");
    run<10, 4>(iters, out, sink, "v_alignbit_b32");
    run<11, 4>(iters, out, sink, "v_fma_f32");
    run<12, 4>(iters, out, sink, "v_cvt_f16_f32");
    run<13, 4>(iters, out, sink, "v_cvt_f16_f32_sdwa dst_sel:WORD_1");
    run<14, 4>(iters, out, sink, "v_pack_b32_f16");
    run<15, 4>(iters, out, sink, "v_cvt_f32_f16_sdwa src0_sel:WORD_1");
    run<16, 4>(iters, out, sink, "v_max_i32");
    run<17, 4>(iters, out, sink, "v_sub_f32");
    run<12, 6>(iters, out, sink, "v_cvt_f16_f32");
    run<6, 6>(iters, out, sink, "v_cvt_pk_f16_f32");
    run<8, 6>(iters, out, sink, "v_dot2_f32_f16");
    run<9, 6>(iters, out, sink, "v_accvgpr_read_b32");
    return 0;
}
