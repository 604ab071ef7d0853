// Diagnostic micro-benchmark: a dependent chain of v_mfma_f32_32x32x2_f32 whose A operand arrives
//   mode 0: from registers (no memory)          mode 1: global_load_dwordx4 per 4 MFMAs, 6 loads in flight
//   mode 2: ds_read_b128 per 4 MFMAs, 2 in flight
// one wave per SIMD.  Reports cycles per MFMA.   hipcc --offload-arch=gfx950 -O3 tools/mfma_feed_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define M4(a4, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], b, acc, 0,0,0); acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], b, acc, 0,0,0); \
                  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], b, acc, 0,0,0); acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], b, acc, 0,0,0);
template <int MODE, int NTHR>
__global__ __launch_bounds__(NTHR) void k(const float* __restrict__ w, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 256];          // 64 KB
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 256; i += NTHR) lds[i] = w[i];
    __syncthreads();
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float b = 1.0f + lane * 1e-4f;
    const f32x4* gp = reinterpret_cast<const f32x4*>(w) + lane;
    const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + lane;
    constexpr int PF = (MODE == 1 || MODE == 3 || MODE == 4) ? 6 : 2;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 256 * 256 * 16, 0x00020000);
    const int voff = lane * 16;
    f32x16 acc2; for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    f32x4 ring[6];
    if (MODE == 4) { for (int i = 0; i < PF; ++i) ring[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, i * 1024, 0)); }
    else for (int i = 0; i < PF; ++i) ring[i] = (MODE == 1 || MODE == 3) ? gp[i * 64] : (MODE == 2 ? lp[i * 64] : f32x4{1.f, 2.f, 3.f, 4.f});
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {                                     // 48 groups per iteration (divisible by 6 and 2)
#pragma unroll
        for (int g = 0; g < 48; ++g) {
            const f32x4 a4 = ring[g % PF];
            if (MODE == 1 || MODE == 3) ring[g % PF] = gp[((it * 48 + g + PF) & 255) * 64];
            if (MODE == 2) ring[g % PF] = lp[((it * 48 + g + PF) & 63) * 64];
            if (MODE == 4) ring[g % PF] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, ((it * 48 + g + PF) & 255) * 1024, 0));
            if (MODE == 3) {            // two independent chains: even groups -> acc, odd groups -> acc2, MFMAs alternate
                if (g & 1) { acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], b, acc2, 0,0,0); acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], b, acc, 0,0,0);
                             acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], b, acc2, 0,0,0); acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], b, acc, 0,0,0); }
                else       { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], b, acc, 0,0,0); acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], b, acc2, 0,0,0);
                             acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], b, acc, 0,0,0); acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], b, acc2, 0,0,0); }
            } else { M4(a4, b) }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
    out[blockIdx.x * NTHR + threadIdx.x] = s + ring[0][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE, int NTHR> void run(const float* w, float* out, unsigned long long* cyc, const char* name) {
    const int iters = 512;
    hipLaunchKernelGGL((k<MODE, NTHR>), dim3(256), dim3(NTHR), 0, 0, w, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %d waves/SIMD: %.2f cycles per MFMA per SIMD\n", name, NTHR / 256, (double)c / (iters * 48.0 * 4 * (NTHR / 256)));
}
int main() {
    float *w, *out; unsigned long long* cyc;
    hipMalloc(&w, 256 * 256 * 4 * 4); hipMemset(w, 0, 256 * 256 * 4 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 256>(w, out, cyc, "registers");
        run<1, 256>(w, out, cyc, "global_load_dwordx4 ring 6");
        run<2, 256>(w, out, cyc, "ds_read_b128 ring 2");
        run<3, 256>(w, out, cyc, "global ring 6, 2 acc chains");
        run<4, 256>(w, out, cyc, "raw_buffer_load_b128 ring 6");
        run<0, 512>(w, out, cyc, "registers");
        run<1, 512>(w, out, cyc, "global_load_dwordx4 ring 6");
        run<2, 512>(w, out, cyc, "ds_read_b128 ring 2");
    }
    return 0;
}
