// Diagnostic micro-benchmark: cycles per v_mfma_f32_32x32x2_f32 for 1, 2 and 4 independent accumulators,
// one wave per SIMD (256 threads / block, 1 block / CU).  hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC> void run(float* out, unsigned long long* cyc) {
    int iters = 4096 / NACC;
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("accumulators=%d: %.2f cycles per MFMA (s_memtime ticks at 100MHz? raw %llu for %d mfma)\n", NACC, (double)c / (iters * 16.0 * NACC), c, iters * 16 * NACC);
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) { run<1>(out, cyc); run<2>(out, cyc); run<4>(out, cyc); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int n : {1, 2, 4}) {
        int iters = 65536 / n;
        hipEventRecord(e0);
        if (n == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        if (n == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        if (n == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = 256.0 * 4 * iters * 16.0 * n * 4096.0;
        printf("accumulators=%d: %.3f ms  %.1f TFLOP/s\n", n, ms, fl / ms / 1e9);
    }
    return 0;
}
