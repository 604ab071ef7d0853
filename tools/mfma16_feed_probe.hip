// Diagnostic micro-benchmark for a possible 16-sample-tile fp32 chain (two or three waves per SIMD):
// dependent chains of v_mfma_f32_16x16x4_f32 whose A operand arrives by raw_buffer_load_b128 (one 1 KB load per
// 4 MFMAs per wave, 6 in flight), every wave of every CU streaming the same 2 MB of weights.
// Reports cycles per MFMA per SIMD (32 = the matrix pipe's rate).   hipcc --offload-arch=gfx950 -O3 tools/mfma16_feed_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int NTHR>
__global__ __launch_bounds__(NTHR) void k(const float* __restrict__ w, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 2048 * 1024, 0x00020000);
    const int voff = lane * 16;
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float b = 1.0f + lane * 1e-4f;
    constexpr int PF = 6;
    f32x4 ring[PF];
    const int wave_off = (threadIdx.x >> 6) * 37;          // waves start at different fragments
    for (int i = 0; i < PF; ++i) ring[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, ((wave_off + i) & 2047) * 1024, 0));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 48; ++g) {
            const f32x4 a4 = ring[g % PF];
            ring[g % PF] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, ((wave_off + it * 48 + g + PF) & 2047) * 1024, 0));
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[(g * 4 + p) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[p], b, acc[(g * 4 + p) % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = ring[0][0];
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * NTHR + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC, int NTHR> void run(const float* w, float* out, unsigned long long* cyc) {
    const int iters = 512;
    hipLaunchKernelGGL((k<NACC, NTHR>), dim3(256), dim3(NTHR), 0, 0, w, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("16x16x4, %d accumulator chain(s), %d wave(s)/SIMD: %.2f cycles per MFMA per SIMD\n", NACC, NTHR / 256, (double)c / (iters * 48.0 * 4 * (NTHR / 256)));
}
int main() {
    float *w, *out; unsigned long long* cyc;
    hipMalloc(&w, 2048 * 1024); hipMemset(w, 0, 2048 * 1024); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<1, 256>(w, out, cyc); run<2, 256>(w, out, cyc);
        run<1, 512>(w, out, cyc); run<2, 512>(w, out, cyc); run<4, 512>(w, out, cyc);
        run<1, 768>(w, out, cyc); run<2, 768>(w, out, cyc);
        run<1, 1024>(w, out, cyc);
    }
    return 0;
}
