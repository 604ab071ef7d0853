// Microbenchmark (round 4): what the chain kernels' epilogue would cost with a different instruction diet.
// One wave per SIMD on every CU; per iteration SIX independent v_mfma_f32_32x32x16_f16 (two (tile, k-step) groups = 192 cycles of
// matrix pipe), the fragment reads of two groups (4 x ds_read_b128) and the epilogue of ONE register pair (two activations) — the
// ratio of a 256-wide layer.
//   part 1: single instruction kinds behind each MFMA (as pk_mfma.hip): v_fma_mix_f32 / mixlo / mixhi (the split without
//           v_cvt + v_sub), v_cmp -> SGPR pair, v_cndmask on an SGPR mask, v_addc_co, s_store_dwordx4, ds reads, buffer_store_dwordx4.
//   part 2: the whole per-pair chain, OLD (round 3: packed fp32, park write / read, dword stores, cvt / sub split) against NEW
//           (plain fp32, no parking, one dwordx4 store per two pairs, v_fma_mix split).
//   epi_mix [iters]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0)

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void k1(int iters, unsigned long long* out, float* sink, float* buf, size_t wave_bytes) {
    extern __shared__ unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wid = (size_t)blockIdx.x * 4 + wave;
    unsigned char* base = reinterpret_cast<unsigned char*>(buf) + wid * wave_bytes;
    {
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)base & 0xffffffffu));
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)base >> 32));
        base = reinterpret_cast<unsigned char*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)wave_bytes, 0x00020000);
    f32x16 a0 = {}, a1 = {}, a2 = {};
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(lane * 0.001f + i); y[i] = (_Float16)(0.5f + i * 0.01f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + lane * 1e-3f + i;
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = 0x3c003c00u + lane + i;
    const float c0 = 1.0000001f;
    float ag = 1.0f + lane;
    u32x4 sd = {1u, 2u, 3u, 4u};
    unsigned soff = 0;
    unsigned ldsa = (unsigned)(uintptr_t)lds + lane * 16;
    unsigned off = lane * 16;
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
                if (KIND == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(r) : "v"(w), "v"(c0));                 // f32 <- f16.lo * f32 + f32
                if (KIND == 3) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(w) : "v"(r), "v"(c0));                                    // f16.lo <- f32 * f32
                if (KIND == 4) asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(w) : "v"(r), "v"(c0), "v"(u[(i + 1) & 7]));   // f16.hi <- f32 * f32 - f16.lo
                if (KIND == 5) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, 0" :: "v"(r) : "s20", "s21");
                if (KIND == 6) asm volatile("v_cndmask_b32_e64 %0, 0, %0, s[20:21]" : "+v"(r));
                if (KIND == 7) asm volatile("v_cmp_gt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(r) : "vcc");   // 2 instructions
                if (KIND == 8) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ag) : "v"(r));
                if (KIND == 9) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(ag));
                if (KIND == 10) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(ldsa)); asm volatile("" :: "v"(t)); }
                if (KIND == 11) { u32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(ldsa)); asm volatile("" :: "v"(t)); }
                if (KIND == 12) asm volatile("v_add_u32 %0, 0x7fffffff, %1\n\tv_alignbit_b32 %2, %2, %0, 31" : "=&v"(w) : "v"(r), "v"(u[(i + 1) & 7]));   // 2 instructions: the sign bit of round 3
                if (KIND == 13) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r) : "v"(c0));
            }
            if (KIND == 14 && m == 0) {      // one s_store_dwordx4 per 3 MFMAs (16 B; a wave's stream of mask words)
                asm volatile("s_store_dwordx4 %0, %1, %2" :: "s"(sd), "s"(base), "s"(soff) : "memory");
                soff += 16;
            }
            if (KIND == 15 && m == 0) {      // one buffer_store_dwordx4 per 3 MFMAs (1 KB per wave)
                const u32x4 d = {u[0], u[1], u[2], u[3]};
                asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt" :: "v"(d), "v"(off), "s"(rsrc) : "memory");
                off += 1024;
            }
            if (KIND == 16 && m != 2) {      // two buffer_store_dword per 3 MFMAs (256 B each)
                asm volatile("buffer_store_dword %0, %1, %2, 0 offen nt" :: "v"(u[m]), "v"(off), "s"(rsrc) : "memory");
                off += 256;
            }
        }
    }
    if (KIND == 14) asm volatile("s_dcache_wb" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = ag;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i];
    for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

// ---- the whole per-pair chain.  Registers: every step reads values written several steps (slots) earlier, as the software-pipelined
// epilogue of the real kernels does: no instruction depends on its predecessor.
template <int NEW, int TRAIN, int FRAG>
__global__ __launch_bounds__(256, 1) void k2(int iters, unsigned long long* out, float* sink, float* buf, size_t wave_bytes) {
    extern __shared__ unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wid = (size_t)blockIdx.x * 4 + wave;
    unsigned char* base = reinterpret_cast<unsigned char*>(buf) + wid * wave_bytes;
    {
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)base & 0xffffffffu));
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((uintptr_t)base >> 32));
        base = reinterpret_cast<unsigned char*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)wave_bytes, 0x00020000);
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {}, a4 = {}, a5 = {};
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(lane * 0.001f + i); y[i] = (_Float16)(0.5f + i * 0.01f); }
    // value registers of the pipeline stages (pairs)
    f32x2 L = {1.f + lane, 2.f}, C = {1e-3f, 2e-3f}, B = {0.f, 0.f}, S1 = {1.f, 1.f}, S2 = {1.f, 2.f}, S3 = {2.f, 1.f}, S4 = {3.f, 1.f}, S5 = {1.f, 3.f}, F = {0.f, 0.f}, l1 = {0.f, 0.f};
    f32x2 Q0 = {1.f, 2.f}, Q1 = {3.f, 4.f};
    unsigned P1 = 0x3c003c00u, P2 = 0x3c003c00u, msk = 0, t1_ = 0, t2_ = 0;
    float agL0 = 1.f + lane, agL1 = 2.f, agC0 = 3.f, agC1 = 4.f, agP0 = 0.f, agP1 = 0.f;
    const float dsc = 1.0000001f, osc = 0.9999999f;
    unsigned ldsa = (unsigned)(uintptr_t)lds + lane * 16, ldsb = (unsigned)(uintptr_t)lds + 65536 + (lane & 7) * 8;
    unsigned off = lane * 4, off4 = lane * 16;
    u32x4 f0, f1, f2, f3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    auto body = [&](auto oddc) __attribute__((always_inline)) {
        constexpr bool odd = decltype(oddc)::value;
        // ---------------- slot 0
        MFMA(a0);
        if (FRAG) { asm volatile("ds_read_b128 %0, %1" : "=v"(f0) : "v"(ldsa)); asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(f1) : "v"(ldsa)); }
        asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=v"(L[0]), "=v"(L[1]) : "a"(agL0), "a"(agL1));
        asm volatile("ds_read_b64 %0, %1" : "=v"(B) : "v"(ldsb));
        if (NEW) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(P1) : "v"(S4[0]), "v"(osc));
        else     asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(S5) : "v"(S4), "v"(Q0));
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- slot 1
        MFMA(a1);
        asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=v"(C[0]), "=v"(C[1]) : "a"(agC0), "a"(agC1));
        if (NEW) {
            asm volatile("v_add_f32 %0, %2, %3\n\tv_add_f32 %1, %4, %5" : "=&v"(S1[0]), "=&v"(S1[1]) : "v"(Q0[0]), "v"(Q1[0]), "v"(Q0[1]), "v"(Q1[1]));
            asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(P1) : "v"(S4[1]), "v"(osc));
        } else {
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(S1) : "v"(Q0), "v"(Q1));
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(P1) : "v"(S5[0]), "v"(S5[1]));
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- slot 2
        MFMA(a2);
        asm volatile("v_fma_f32 %0, %2, %3, %4\n\tv_fma_f32 %1, %5, %3, %6" : "=&v"(S2[0]), "=&v"(S2[1]) : "v"(S1[0]), "v"(dsc), "v"(Q0[0]), "v"(S1[1]), "v"(Q0[1]));
        if (NEW) {
            asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(P2) : "v"(S4[0]), "v"(osc), "v"(t1_));
        } else {
            asm volatile("v_cvt_f32_f16_e32 %0, %2\n\tv_cvt_f32_f16_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=&v"(F[0]), "=&v"(F[1]) : "v"(t1_));
            if (TRAIN) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(agP0) : "v"(S3[0]));        // parking: half of the pairs
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- slot 3
        MFMA(a3);
        if (FRAG) { asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(f2) : "v"(ldsa)); asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(f3) : "v"(ldsa)); }
        asm volatile("v_max_i32 %0, 0, %1\n\tv_max_i32 %2, 0, %3" : "=&v"(S3[0]), "=&v"(S3[1]) : "v"(Q1[0]), "v"(Q1[1]));
        if (NEW) {
            asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(P2) : "v"(S4[1]), "v"(osc), "v"(t1_));
        } else {
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(Q1) : "v"(S5), "v"(F));
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(S4[0]) : "a"(agP1));
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- slot 4
        MFMA(a4);
        if (TRAIN) {
            if (NEW) {
                if (odd) { const f32x4 d = {S3[0], S3[1], S4[0], S4[1]}; asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen nt" :: "v"(d), "v"(off4), "s"(rsrc) : "memory"); off4 += 1024; }
            } else {
                asm volatile("buffer_store_dword %0, %1, %2, 0 offen nt" :: "v"(S3[0]), "v"(off), "s"(rsrc) : "memory");
                asm volatile("buffer_store_dword %0, %1, %2, 0 offen offset:256 nt" :: "v"(S3[1]), "v"(off), "s"(rsrc) : "memory");
                off += 512;
            }
            asm volatile("v_add_u32 %0, 0x7fffffff, %1\n\tv_add_u32 %2, 0x7fffffff, %3" : "=&v"(t1_), "=&v"(t2_) : "v"(Q0[0]), "v"(Q0[1]));
        }
        if (NEW) asm volatile("v_add_f32 %0, %0, %1" : "+v"(l1[0]) : "v"(Q0[0]));
        else     asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(l1) : "v"(Q0));
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- slot 5
        MFMA(a5);
        if (TRAIN) asm volatile("v_alignbit_b32 %0, %0, %1, 31\n\tv_alignbit_b32 %0, %0, %2, 31" : "+v"(msk) : "v"(u32x4{1, 2, 3, 4}[0] + lane), "v"(lane));
        if (NEW) asm volatile("v_add_f32 %0, %0, %1" : "+v"(l1[1]) : "v"(Q0[1]));
        else     asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(P2) : "v"(Q1[0]), "v"(Q1[1]));
        __builtin_amdgcn_sched_barrier(0);
        if (FRAG) asm volatile("" :: "v"(f0), "v"(f1), "v"(f2), "v"(f3));
    };
    for (int it = 0; it < iters; it += 2) { body(std::false_type{}); body(std::true_type{}); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = agP0 + agP1;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i] + a4[i] + a5[i];
    s += L[0] + L[1] + C[0] + C[1] + B[0] + B[1] + S1[0] + S1[1] + S2[0] + S2[1] + S3[0] + S3[1] + S4[0] + S5[0] + S5[1] + F[0] + F[1] + l1[0] + l1[1] + Q1[0] + Q1[1];
    s += (float)P1 + (float)P2 + (float)msk + (float)t1_ + (float)t2_;
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

static double median_cycles(unsigned long long* out, int iters) {
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), out, sizeof(unsigned long long) * 1024, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return (double)h[512] / iters;
}
template <int KIND, int N>
static void run1(int iters, unsigned long long* out, float* sink, float* buf, size_t wb, const char* what) {
    (void)hipFuncSetAttribute((const void*)k1<KIND, N>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k1<KIND, N>), dim3(256), dim3(256), 100 * 1024, 0, iters, out, sink, buf, wb);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", what); exit(2); }
    }
    printf("%-52s N = %d per MFMA: %7.1f cycles per iteration (3 MFMA = 96)\n", what, N, median_cycles(out, iters));
    fflush(stdout);
}
template <int NEW, int TRAIN, int FRAG>
static void run2(int iters, unsigned long long* out, float* sink, float* buf, size_t wb, const char* what) {
    (void)hipFuncSetAttribute((const void*)k2<NEW, TRAIN, FRAG>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k2<NEW, TRAIN, FRAG>), dim3(256), dim3(256), 100 * 1024, 0, iters, out, sink, buf, wb);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", what); exit(2); }
    }
    printf("%-72s %7.1f cycles per iteration (6 MFMA = 192)\n", what, median_cycles(out, iters));
    fflush(stdout);
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 2048;
    if (iters < 2 || iters > 2048) iters = 2048;        // a wave's 4 MB region is never overrun (1 KB per iteration at most ... 2 KB)
    const size_t wb = (size_t)4 << 20;
    unsigned long long* out; float* sink; float* buf;
    if (hipMalloc(&out, 1024 * sizeof(unsigned long long)) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess || hipMalloc(&buf, wb * 1024) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, wb * 1024);
    run1<0, 1>(iters, out, sink, buf, wb, "MFMAs only");
    run1<1, 4>(iters, out, sink, buf, wb, "v_mul_f32");
    run1<13, 4>(iters, out, sink, buf, wb, "v_add_f32");
    run1<2, 4>(iters, out, sink, buf, wb, "v_fma_mix_f32 (f16 x f32 + f32)");
    run1<3, 4>(iters, out, sink, buf, wb, "v_fma_mixlo_f16 (f32 x f32 -> f16.lo)");
    run1<4, 4>(iters, out, sink, buf, wb, "v_fma_mixhi_f16 (f32 x f32 - f16 -> f16.hi)");
    run1<5, 4>(iters, out, sink, buf, wb, "v_cmp_gt_f32_e64 -> SGPR pair");
    run1<6, 4>(iters, out, sink, buf, wb, "v_cndmask_b32 on an SGPR mask");
    run1<7, 2>(iters, out, sink, buf, wb, "v_cmp_gt_f32 vcc + v_addc_co (2 instr)");
    run1<12, 2>(iters, out, sink, buf, wb, "v_add_u32 + v_alignbit (2 instr)");
    run1<8, 4>(iters, out, sink, buf, wb, "v_accvgpr_write_b32");
    run1<9, 4>(iters, out, sink, buf, wb, "v_accvgpr_read_b32");
    run1<10, 2>(iters, out, sink, buf, wb, "ds_read_b64");
    run1<11, 2>(iters, out, sink, buf, wb, "ds_read_b128");
    run1<2, 6>(iters, out, sink, buf, wb, "v_fma_mix_f32");
    run1<3, 6>(iters, out, sink, buf, wb, "v_fma_mixlo_f16");
    run1<1, 6>(iters, out, sink, buf, wb, "v_mul_f32");
    run1<14, 1>(iters, out, sink, buf, wb, "s_store_dwordx4, one per 3 MFMAs");
    run1<15, 1>(iters, out, sink, buf, wb, "buffer_store_dwordx4 nt, one per 3 MFMAs (1 KB)");
    run1<16, 1>(iters, out, sink, buf, wb, "buffer_store_dword nt, two per 3 MFMAs (2 x 256 B)");
    printf("--- whole per-pair chains (one pair + 4 fragment reads per 6 MFMAs)\n");
    run2<0, 0, 1>(iters, out, sink, buf, wb, "OLD inference (pk fp32, cvt/sub split)");
    run2<1, 0, 1>(iters, out, sink, buf, wb, "NEW inference (plain fp32, v_fma_mix split)");
    run2<0, 1, 1>(iters, out, sink, buf, wb, "OLD training  (+ park, 2 dword stores, sign bits)");
    run2<1, 1, 1>(iters, out, sink, buf, wb, "NEW training  (+ dwordx4 store per 2 pairs, sign bits)");
    run2<0, 0, 0>(iters, out, sink, buf, wb, "OLD inference, no fragment reads");
    run2<1, 0, 0>(iters, out, sink, buf, wb, "NEW inference, no fragment reads");
    return 0;
}
