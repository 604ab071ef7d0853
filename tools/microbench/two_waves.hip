// Microbenchmark (round 4): does a SECOND wave on the SIMD hide the non-MFMA instructions that one wave cannot?
// epi_mix.hip found that one wave per SIMD gets about 3 VALU instructions per v_mfma_f32_32x32x16_f16 for free and pays ~3.4 cycles
// for every further one.  Here the same loop (3 MFMAs + 3 N instructions of one kind per iteration) runs with 1 or 2 waves per SIMD
// (blocks of 256 or 512 threads, one block per CU); the figure is SIMD cycles per 3 MFMAs = elapsed / iterations / waves per SIMD
// (floor 96).  If the matrix pipe and the VALU issue of DIFFERENT waves overlap, the 2-wave column stays near 96 where the 1-wave
// column grows.
//   two_waves [iters]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0)

template <int KIND, int N, int NT>
__global__ __launch_bounds__(NT, 1) void k(int iters, unsigned long long* out, float* sink) {
    extern __shared__ unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 a0 = {}, a1 = {}, a2 = {};
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(lane * 0.001f + i); y[i] = (_Float16)(0.5f + i * 0.01f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + lane * 1e-3f + i;
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = 0x3c003c00u + lane + i;
    const float c0 = 1.0000001f;
    float ag = 1.0f + lane;
    unsigned ldsa = (unsigned)(uintptr_t)lds + lane * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            if (m == 0) MFMA(a0);
            if (m == 1) MFMA(a1);
            if (m == 2) MFMA(a2);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                float& r = v[i & 7];
                unsigned& w = u[i & 7];
                if (KIND == 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r) : "v"(c0));
                if (KIND == 3) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(w) : "v"(r), "v"(c0));
                if (KIND == 9) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(ag));
                if (KIND == 30) asm volatile("s_add_i32 s20, s20, 0x4000" ::: "s20", "scc");
                if (KIND == 31) asm volatile("s_nop 0");
                if (KIND == 32) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (KIND == 33) asm volatile("s_add_i32 s20, s20, 0x4000\n\ts_cmp_lg_u32 s20, 0x1c000\n\ts_cselect_b32 s20, s20, 0" ::: "s20", "scc");     // a ring pointer's wrap: 3 instructions
                if (KIND == 11) { u32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(ldsa)); asm volatile("" :: "v"(t)); }
                if (KIND == 20) {              // the real mixture of a training pair: acc read, fma, alignbit, med3, mix, cvt_pk ...
                    if (i % 6 == 0) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(ag));
                    if (i % 6 == 1) asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(r) : "v"(c0));
                    if (i % 6 == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(w) : "v"(r));
                    if (i % 6 == 3) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(r) : "v"(c0));
                    if (i % 6 == 4) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(r) : "v"(w), "v"(c0));
                    if (i % 6 == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w) : "v"(r), "v"(v[(i + 1) & 7]));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = ag;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i];
    for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * (NT / 64) + wave] = t1 - t0;
}

template <int KIND, int N, int NT>
static double run(int iters, unsigned long long* out, float* sink) {
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<KIND, N, NT>), dim3(256), dim3(NT), 16 * 1024, 0, iters, out, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); exit(2); }
    }
    const int n = 256 * NT / 64;
    std::vector<unsigned long long> h(n);
    (void)hipMemcpy(h.data(), out, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return (double)h[n / 2] / iters / (NT / 256);
}
template <int KIND, int N>
static void both(int iters, unsigned long long* out, float* sink, const char* what) {
    const double one = run<KIND, N, 256>(iters, out, sink), two = run<KIND, N, 512>(iters, out, sink), three = run<KIND, N, 768>(iters, out, sink);
    printf("%-34s N = %2d per MFMA: SIMD cycles per 3 MFMAs (floor 96): 1 wave %6.1f   2 waves %6.1f   3 waves %6.1f\n", what, N, one, two, three);
    fflush(stdout);
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 4096;
    unsigned long long* out; float* sink;
    if (hipMalloc(&out, 4096 * sizeof(unsigned long long)) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    both<0, 1>(iters, out, sink, "MFMAs only");
    both<1, 2>(iters, out, sink, "v_mul_f32");
    both<1, 4>(iters, out, sink, "v_mul_f32");
    both<1, 6>(iters, out, sink, "v_mul_f32");
    both<1, 8>(iters, out, sink, "v_mul_f32");
    both<1, 12>(iters, out, sink, "v_mul_f32");
    both<3, 4>(iters, out, sink, "v_fma_mixlo_f16");
    both<3, 6>(iters, out, sink, "v_fma_mixlo_f16");
    both<9, 4>(iters, out, sink, "v_accvgpr_read_b32");
    both<9, 6>(iters, out, sink, "v_accvgpr_read_b32");
    both<11, 1>(iters, out, sink, "ds_read_b128");
    both<30, 2>(iters, out, sink, "s_add_i32");
    both<30, 4>(iters, out, sink, "s_add_i32");
    both<30, 8>(iters, out, sink, "s_add_i32");
    both<31, 4>(iters, out, sink, "s_nop 0");
    both<31, 8>(iters, out, sink, "s_nop 0");
    both<32, 4>(iters, out, sink, "s_waitcnt (satisfied)");
    both<33, 2>(iters, out, sink, "ring wrap (3 SALU) x N");
    both<20, 6>(iters, out, sink, "training-pair mixture");
    both<20, 12>(iters, out, sink, "training-pair mixture");
    return 0;
}
