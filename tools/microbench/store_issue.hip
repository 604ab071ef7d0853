// Microbenchmark: what does a stash store cost a wave that runs ALONE on its SIMD (the chain kernels' situation)?
// Every wave runs `iters` iterations of  [3 independent v_mfma_f32_32x32x16_f16 + NV v_add_f32 + the stores of one form]  and streams
// its stores to its own region of a buffer far larger than the caches (as the training kernels do: 2 stores per 3 MFMAs).
//   form 0: no stores (the MFMA / VALU floor)
//   form 1: 2 x global_store_dword, 64-bit VGPR address, nt           (what hipcc emits for the stash today)
//   form 2: 2 x global_store_dword, SGPR base + 32-bit VGPR offset, nt
//   form 3: 1 x global_store_dwordx2, 64-bit VGPR address, nt         (same bytes, half the instructions)
//   form 4: 1 x global_store_dwordx4 per 1 KB                         (same bytes, a quarter of the instructions)
//   form 5: as 1 without nt
//   form 6: 2 x buffer_store_dword, buffer resource + 32-bit VGPR offset (out-of-range lanes are dropped by the hardware)
//   store_issue [iters]      prints cycles per iteration (s_memtime, median over waves), shader clock and the write rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int FORM, int NV, int EVERY>
__global__ __launch_bounds__(256, 1) void k(float* buf, size_t wave_bytes, int iters, unsigned long long* out) {
    extern __shared__ unsigned char lds[];           // 100 KB requested at launch: one workgroup per CU, one wave per SIMD
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wid = (size_t)blockIdx.x * 4 + wave;
    unsigned char* base = reinterpret_cast<unsigned char*>(buf) + wid * wave_bytes;
    {   // wave-uniform copy of the base for the SGPR form (readfirstlane returns int: widen through unsigned, no sign extension)
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)base & 0xffffffffu));
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)base >> 32));
        base = reinterpret_cast<unsigned char*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)wave_bytes, 0x00020000);
    f32x16 a0 = {}, a1 = {}, a2 = {};
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(lane * 0.001f + i); y[i] = (_Float16)(0.5f + i * 0.01f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned off = lane * 4;                          // byte offset inside the wave's region
    for (int it = 0; it < iters; ++it) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV / 2; ++i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(v[i & 7]));
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
        // EVERY = 2: 512 B per two iterations = 256 B per 3 MFMAs, the training kernels' ratio (one fp32 stash store per three MFMAs)
        const bool now = (it % EVERY) == 0, now4 = (it % (2 * EVERY)) == 0;
        if ((FORM == 1 || FORM == 5) && now) {
            unsigned char* p = base + off;
            if (FORM == 1) asm volatile("global_store_dword %0, %1, off nt\n\tglobal_store_dword %0, %2, off offset:256 nt" :: "v"(p), "v"(v[0]), "v"(v[1]) : "memory");
            else           asm volatile("global_store_dword %0, %1, off\n\tglobal_store_dword %0, %2, off offset:256" :: "v"(p), "v"(v[0]), "v"(v[1]) : "memory");
            off += 512;
        } else if (FORM == 2 && now) {
            asm volatile("global_store_dword %0, %1, %3 nt\n\tglobal_store_dword %0, %2, %3 offset:256 nt" :: "v"(off), "v"(v[0]), "v"(v[1]), "s"(base) : "memory");
            off += 512;
        } else if (FORM == 3 && now) {
            unsigned char* p = base + (off - lane * 4) + lane * 8;
            const f32x2 d = {v[0], v[1]};
            asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(p), "v"(d) : "memory");
            off += 512;
        } else if (FORM == 6 && now) {                  // buffer form: V# (4 SGPRs) + 32-bit VGPR offset; lanes past num_records are dropped by the hardware
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[0]), rsrc, (int)off, 0, 2);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[1]), rsrc, (int)off + 256, 0, 2);
            off += 512;
        } else if (FORM == 4 && now4) {
            unsigned char* p = base + (off - lane * 4) + lane * 16;
            const f32x4 d = {v[0], v[1], v[2], v[3]};
            asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(d) : "memory");
            off += 1024;
        }
#pragma unroll
        for (int i = 0; i < NV - NV / 2; ++i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(v[(i + 4) & 7]));
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a2, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) buf[0] = s;
    if (lane == 0) { out[2 * wid] = t1 - t0; out[2 * wid + 1] = r1 - r0; }
}

template <int FORM, int NV, int EVERY>
static void run(float* buf, size_t wave_bytes, int iters, unsigned long long* out, const char* what) {
    hipFuncSetAttribute((const void*)k<FORM, NV, EVERY>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<FORM, NV, EVERY>), dim3(256), dim3(256), 100 * 1024, 0, buf, wave_bytes, iters, out);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); exit(2); }
    }
    std::vector<unsigned long long> h(2048);
    hipMemcpy(h.data(), out, sizeof(unsigned long long) * 2048, hipMemcpyDeviceToHost);
    std::vector<double> cyc, rt;
    for (int i = 0; i < 1024; ++i) { cyc.push_back((double)h[2 * i]); rt.push_back((double)h[2 * i + 1]); }
    std::sort(cyc.begin(), cyc.end()); std::sort(rt.begin(), rt.end());
    const double c = cyc[512], us = rt[512] / 100.0;
    const double bytes = FORM == 0 ? 0.0 : 1024.0 * iters * 512.0 / EVERY;
    printf("form %d NV %2d every %d  %-52s %7.1f cycles/iteration (3 MFMA = 96)  clock %.3f GHz  %.2f TB/s\n", FORM, NV, EVERY, what, c / iters, c / us * 1e-3, bytes / (us * 1e-6) * 1e-12);
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 4096;
    if (iters < 2 || iters > 8000) iters = 4096;      // a wave's 4 MB region is never wrapped (the dwordx4 form looks one iteration back)
    const size_t wave_bytes = (size_t)4 << 20;       // 4 MB per wave, 4 GB in all
    float* buf; unsigned long long* out;
    if (hipMalloc(&buf, wave_bytes * 1024) != hipSuccess || hipMalloc(&out, 2048 * sizeof(unsigned long long)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, wave_bytes * 1024);
    run<0, 8, 1>(buf, wave_bytes, iters, out, "no stores");
    run<1, 8, 2>(buf, wave_bytes, iters, out, "2 x dword, 64-bit vaddr, nt");
    run<2, 8, 2>(buf, wave_bytes, iters, out, "2 x dword, saddr + voffset, nt");
    run<3, 8, 2>(buf, wave_bytes, iters, out, "1 x dwordx2, 64-bit vaddr, nt");
    run<4, 8, 2>(buf, wave_bytes, iters, out, "1 x dwordx4 per 1 KB, nt");
    run<5, 8, 2>(buf, wave_bytes, iters, out, "2 x dword, 64-bit vaddr, default policy");
    run<6, 8, 2>(buf, wave_bytes, iters, out, "2 x buffer_store_dword, V# + voffset, slc");
    run<0, 12, 1>(buf, wave_bytes, iters, out, "no stores");
    run<1, 12, 2>(buf, wave_bytes, iters, out, "2 x dword, 64-bit vaddr, nt");
    run<2, 12, 2>(buf, wave_bytes, iters, out, "2 x dword, saddr + voffset, nt");
    run<3, 12, 2>(buf, wave_bytes, iters, out, "1 x dwordx2, 64-bit vaddr, nt");
    run<4, 12, 2>(buf, wave_bytes, iters, out, "1 x dwordx4 per 1 KB, nt");
    run<6, 12, 2>(buf, wave_bytes, iters, out, "2 x buffer_store_dword, V# + voffset, slc");
    run<1, 8, 4>(buf, wave_bytes, iters, out, "2 x dword, 64-bit vaddr, nt");
    run<4, 8, 4>(buf, wave_bytes, iters, out, "1 x dwordx4 per 1 KB, nt");
    return 0;
}
