// Microbenchmark: how many bytes per clock can a CU pull from a 2.3 MB L2-resident weight stream when ALL CUs pull the same stream
// (the chain kernels' situation), by path:
//   mode 0: LDS-DMA (global_load_lds_dwordx4), the four waves share a 16 KB stage          (what the kernels do)
//   mode 1: global_load_dwordx4 -> VGPR, every wave loads EVERY fragment (the vector L1 sees 4x the bytes, L2 the same)
//   mode 2: global_load_dwordx4 -> VGPR, every wave loads its quarter only (raw L2 -> CU register path)
//   mode 3: both at once: half of each stage by LDS-DMA, the other half by every wave into VGPRs
//   l2_feed <mode> [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define STAGE 16384
__device__ __forceinline__ void glds16(const void* src, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(src), "s"(lds_dst) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const unsigned char* stream, int n_stage, int passes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    u32x4 acc = {0, 0, 0, 0};
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    int slot = 0;
    for (int p = 0; p < passes; ++p) {
        for (int s = 0; s < n_stage; ++s) {
            const unsigned char* st = stream + (size_t)s * STAGE;
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) glds16(st, (wave * 4 + i) * 1024 + lane * 16, lds0 + slot * STAGE + (wave * 4 + i) * 1024);
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { const u32x4 v = *reinterpret_cast<const u32x4*>(st + i * 1024 + lane * 16); acc ^= v; }
            } else if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { const u32x4 v = *reinterpret_cast<const u32x4*>(st + (wave * 4 + i) * 1024 + lane * 16); acc ^= v; }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) glds16(st, (wave * 2 + i) * 1024 + lane * 16, lds0 + slot * STAGE + (wave * 2 + i) * 1024);
#pragma unroll
                for (int i = 8; i < 16; ++i) { const u32x4 v = *reinterpret_cast<const u32x4*>(st + i * 1024 + lane * 16); acc ^= v; }
            }
            slot = (slot + 1) % 7;
            if ((s & 3) == 3) { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }      // keep ~1.5 stages of this wave's loads in flight
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 1 && MODE != 2) acc[0] ^= *reinterpret_cast<unsigned*>(lds + threadIdx.x * 4);
    if (acc[0] == 0x12345678u) sink[0] = acc[1] ^ acc[2] ^ acc[3];
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, passes = argc > 2 ? atoi(argv[2]) : 64;
    const int n_stage = 141;                                            // 2.26 MB, the 8x256 forward stream
    unsigned char* d; unsigned* sink;
    hipMalloc(&d, (size_t)n_stage * STAGE); hipMalloc(&sink, 4);
    hipMemset(d, 1, (size_t)n_stage * STAGE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        const size_t l = 7 * STAGE;
        if (mode == 0) { hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l); hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), l, 0, d, n_stage, passes, sink); }
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, n_stage, passes, sink);
        else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, n_stage, passes, sink);
        else { hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l); hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), l, 0, d, n_stage, passes, sink); }
    };
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)passes * n_stage * STAGE * 256;       // stream bytes pulled per CU x 256 CUs (mode 1 / 3: bytes through L2, not through the vector L1)
    printf("mode %d: %.3f ms, %.2f TB/s chip-wide of stream bytes = %.1f GB/s per CU (%s)\n", mode, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256,
           mode == 0 ? "LDS-DMA" : mode == 1 ? "VGPR loads, every wave every fragment" : mode == 2 ? "VGPR loads, a quarter per wave" : "half LDS-DMA + half VGPR by every wave");
    return 0;
}
