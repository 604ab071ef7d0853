// Microbenchmark: the x3 inner loop (A fragments re-read from LDS, B pieces in registers, 6 bf16 MFMAs per product tile) in the
// two bf16 MFMA shapes at the same output tile per wave (256 features x 32 samples), random data, one wave per SIMD.
// Decides whether moving the chain kernels to v_mfma_f32_16x16x32_bf16 is worth it (MI355X_MICROARCH.md, DVFS give-back item 7).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define KSTEPS 16          // 256-wide layer as 16 k-steps of 16
#define LAYERS 64

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void k_loop(const u32x4* __restrict__ w, const u32x4* __restrict__ x, float* out, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    // 96 KB of "weights" in LDS (4 k-steps x 24 fragments), random
    for (int i = threadIdx.x; i < 96 * 64; i += 256) reinterpret_cast<u32x4*>(lds)[i] = w[(blockIdx.x % 7) * 96 * 64 + i];
    __syncthreads();
    bf16x8 b1[KSTEPS], b2[KSTEPS], b3[KSTEPS];
#pragma unroll
    for (int k = 0; k < KSTEPS; ++k) {
        b1[k] = __builtin_bit_cast(bf16x8, x[(k * 3 + 0) * 64 + lane]);
        b2[k] = __builtin_bit_cast(bf16x8, x[(k * 3 + 1) * 64 + lane]);
        b3[k] = __builtin_bit_cast(bf16x8, x[(k * 3 + 2) * 64 + lane]);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = f32x16{};
        for (int l = 0; l < LAYERS; ++l) {
#pragma unroll
            for (int k = 0; k < KSTEPS; ++k) {
                const unsigned char* base = lds + (k & 3) * 24 * 1024 + lane * 16;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(base + (t * 3 + 0) * 1024);
                    const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(base + (t * 3 + 1) * 1024);
                    const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(base + (t * 3 + 2) * 1024);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1[k], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2[k], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3[k], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1[k], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2[k], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1[k], acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[t][r];
    } else {
        // 16 feature tiles x 2 sample sub-tiles; a k-step is 32 wide: b*[2q] / b*[2q+1] are the two sample sub-tiles of k-step q
        f32x4 acc[16][2];
#pragma unroll
        for (int t = 0; t < 16; ++t) { acc[t][0] = f32x4{}; acc[t][1] = f32x4{}; }
        for (int l = 0; l < LAYERS; ++l) {
#pragma unroll
            for (int q = 0; q < KSTEPS / 2; ++q) {
                const unsigned char* base = lds + (q & 1) * 48 * 1024 + lane * 16;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(base + (t * 3 + 0) * 1024);
                    const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(base + (t * 3 + 1) * 1024);
                    const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(base + (t * 3 + 2) * 1024);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1[2 * q + u], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2[2 * q + u], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3[2 * q + u], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1[2 * q + u], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2[2 * q + u], acc[t][u], 0, 0, 0);
                        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1[2 * q + u], acc[t][u], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sum += acc[t][0][r] + acc[t][1][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int n_wg = 256;
    std::vector<unsigned> hw(7 * 96 * 64 * 4), hx(KSTEPS * 3 * 64 * 4);
    srand(1);
    auto rb = [] { // a random bf16 in about [-1, 1]: sign, exponent 120..127, random mantissa
        return (unsigned)(((rand() & 1) << 15) | ((120 + (rand() & 7)) << 7) | (rand() & 127)); };
    for (auto& v : hw) v = rb() | (rb() << 16);
    for (auto& v : hx) v = rb() | (rb() << 16);
    unsigned *dw, *dx; float* dout; unsigned long long* dclk;
    hipMalloc(&dw, hw.size() * 4); hipMalloc(&dx, hx.size() * 4); hipMalloc(&dout, n_wg * 256 * 4); hipMalloc(&dclk, n_wg * 16);
    hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k_loop<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void*)k_loop<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = 2.0 * 256 * 256 * 32 * 6 * LAYERS * 4 * n_wg;     // executed bf16 FLOP per launch
    for (int round = 0; round < 3; ++round)
        for (int shape : {32, 16}) {
            auto launch = [&] {
                if (shape == 32) hipLaunchKernelGGL(k_loop<32>, dim3(n_wg), dim3(256), 96 * 1024, 0, (const u32x4*)dw, (const u32x4*)dx, dout, dclk);
                else             hipLaunchKernelGGL(k_loop<16>, dim3(n_wg), dim3(256), 96 * 1024, 0, (const u32x4*)dw, (const u32x4*)dx, dout, dclk);
            };
            for (int i = 0; i < 300; ++i) launch();                        // warm: the clock settles under load
            hipEventRecord(e0);
            const int reps = 300;
            for (int i = 0; i < reps; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> hc(n_wg * 2);
            hipMemcpy(hc.data(), dclk, n_wg * 16, hipMemcpyDeviceToHost);
            double cyc = 0, real = 0; for (int i = 0; i < n_wg; ++i) { cyc += hc[2 * i]; real += hc[2 * i + 1]; }
            printf("shape %2d: %.4f ms/launch  %.1f TFLOP/s bf16 executed  (%.1f fp32-equivalent)  cycles/WG %.0f  clock %.3f GHz\n", shape, ms / reps,
                   flop / (ms / reps * 1e-3) / 1e12, flop / 6 / (ms / reps * 1e-3) / 1e12, cyc / n_wg, cyc / real * 0.1);
        }
    return 0;
}
