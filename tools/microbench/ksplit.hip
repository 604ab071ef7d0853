// Microbenchmark (round 4, for the next round's design): would a 256-wide layer walk with TWO waves per SIMD pay off if the two waves of
// a SIMD SPLIT THE K RANGE of one 32-sample tile?  Each wave then holds half of the tile's input pieces (64 registers) and one half-pass'
// accumulators (128), i.e. fits the 256 registers of a two-wave SIMD — but every half-pass ends with an exchange of partial sums
// through LDS (the partner's 64 summed accumulator registers: 16 ds_write_b128 + 16 ds_read_b128 per wave and half-pass) on top of
// the weight fragments every wave reads (2 x ds_read_b128 per 3 MFMAs).  LDS traffic at the matrix pipe's floor: 104 B/cycle/CU of 128.
//
// The loop: a "half-pass" = 32 groups x 3 MFMAs (8 k-steps x 4 tiles) with the fragments read one group ahead, a barrier every 8 groups
// (ring stage), NV plain VALU instructions per MFMA (the epilogue), and — EXCH — the exchange of the PREVIOUS half-pass spread over it:
// 2 ds_write_b128 per group in groups 0..7, 2 ds_read_b128 per group in groups 8..15 (behind the stage barrier), 6 v_add per group.
//   MODE 0: today's shape — 4 waves per CU, 64 groups per half-pass, no exchange
//   MODE 1: 8 waves per CU (two per SIMD), 32 groups per half-pass, exchange
//   MODE 2: 8 waves per CU, 32 groups, NO exchange (what the exchange costs)
// Prints SIMD cycles per MFMA (floor 32).
//   ksplit [passes]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define H8(x) __builtin_bit_cast(f16x8, (x))
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
    [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) { (f(std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, N>{});
}

// (asm operands cannot name variables captured by a lambda: helpers taking parameters)
__device__ __forceinline__ void x_vmul(float& r, float c) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r) : "v"(c)); }
__device__ __forceinline__ void x_vadd(float& r, float c) { asm volatile("v_add_f32 %0, %1, %0" : "+v"(r) : "v"(c)); }
template <int OFF> __device__ __forceinline__ void x_dswrite(unsigned a, const u32x4& d) { asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(a), "v"(d), "n"(OFF) : "memory"); }
template <int OFF> __device__ __forceinline__ void x_dsread(u32x4& t, unsigned a) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(a), "n"(OFF) : "memory"); }
__device__ __forceinline__ void x_wait(u32x4& a, u32x4& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) :: "memory"); }

template <int MODE, int NV>
__global__ __launch_bounds__(MODE == 0 ? 256 : 512, 1) void k(int passes, unsigned long long* out, float* sink) {
    constexpr int NT = MODE == 0 ? 256 : 512, NG = MODE == 0 ? 64 : 32, KS = MODE == 0 ? 16 : 8;
    constexpr bool EXCH = MODE == 1;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 36 * 1024; i += NT) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003800u + (i & 255);
    __syncthreads();
    u32x4 X1[KS], X2[KS];
    sfor<KS>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        X1[i] = u32x4{0x3c003c00u + lane, 0x38003a00u + i, 0x3c003c00u, 0x34003c00u}; X2[i] = u32x4{0x1c001c00u, 0x18001a00u + i, 0x1c001c00u + lane, 0x14001c00u};
    });
    f32x16 acc[8];
    sfor<8>([&](auto ic) __attribute__((always_inline)) { for (int r = 0; r < 16; ++r) acc[decltype(ic)::value][r] = 0.f; });
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + lane * 1e-3f + i;
    const float c0 = 1.0000001f;
    // exchange area: behind the 112 KB "ring", 16 KB per wave
    const unsigned xw = (unsigned)(uintptr_t)lds + 112 * 1024 + (wave & 7) * 4096 + lane * 16;      // (4 KB windows reused: the traffic is what counts)
    uint32_t cur = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int p = 0; p < passes; ++p) {
        f16x8 fa1[2], fa2[2];
        u32x4 tt[2] = {};
        auto load = [&](int slot, int g) __attribute__((always_inline)) {
            const unsigned char* b = lds + cur + lane * 16 + (g % 8) * 2048;
            fa1[slot] = *reinterpret_cast<const f16x8*>(b); fa2[slot] = *reinterpret_cast<const f16x8*>(b + 1024);
        };
        load(0, 0);
        sfor<NG>([&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value, ks = g / 4, tl = g % 4;
            if constexpr (g % 8 == 0 && g > 0) {
                __builtin_amdgcn_s_barrier();
                cur += 16384; if (cur == 7 * 16384) cur = 0;
            }
            const f16x8 a1 = fa1[g % 2], a2 = fa2[g % 2];
            const f16x8 b1 = H8(X1[ks]), b2 = H8(X2[ks]);
            sfor<3>([&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                if constexpr (m == 0) { acc[tl + 4] = MF(a2, b1, acc[tl + 4]); if constexpr (g + 1 < NG) load((g + 1) % 2, g + 1); }
                if constexpr (m == 1) acc[tl + 4] = MF(a1, b2, acc[tl + 4]);
                if constexpr (m == 2) acc[tl] = MF(a1, b1, acc[tl]);
                sfor<NV>([&](auto ic) __attribute__((always_inline)) { x_vmul(v[(m * NV + decltype(ic)::value) & 7], c0); });
                if constexpr (EXCH) {
                    if constexpr (g < 8 && m < 2) {                       // 2 ds_write_b128 per group: this wave's summed partials of the previous half-pass
                        const u32x4 d = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                        x_dswrite<((g * 2 + m) & 3) * 1024>(xw, d);
                    }
                    if constexpr (g >= 8 && g < 16 && m < 2) {            // 2 ds_read_b128 per group: the partner's; consumed behind the group's third MFMA
                        x_dsread<((g * 2 + m) & 3) * 1024>(tt[m], xw ^ 16384u);
                    }
                    if constexpr (g >= 8 && g < 16 && m == 2) {
                        x_wait(tt[0], tt[1]);
                        v[4] += __uint_as_float(tt[0][0]) + __uint_as_float(tt[1][0]); v[5] += __uint_as_float(tt[0][1]) + __uint_as_float(tt[1][1]);
                        v[6] += __uint_as_float(tt[0][2]) + __uint_as_float(tt[1][2]); v[7] += __uint_as_float(tt[0][3]) + __uint_as_float(tt[1][3]);
                    }
                    if constexpr (g < 16) { x_vadd(v[m], c0); x_vadd(v[m + 3], c0); }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        __builtin_amdgcn_s_barrier();
        cur += 16384; if (cur == 7 * 16384) cur = 0;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    sfor<8>([&](auto ic) __attribute__((always_inline)) { for (int r = 0; r < 16; ++r) s += acc[decltype(ic)::value][r]; });
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) out[blockIdx.x * (NT / 64) + wave] = t1 - t0;
}

template <int MODE, int NV>
static void run(int passes, unsigned long long* out, float* sink, const char* what) {
    constexpr int NT = MODE == 0 ? 256 : 512, NG = MODE == 0 ? 64 : 32;
    (void)hipFuncSetAttribute((const void*)k<MODE, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 148 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(NT), 148 * 1024, 0, passes, out, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", what); exit(2); }
    }
    const int n = 256 * NT / 64;
    std::vector<unsigned long long> h(n);
    (void)hipMemcpy(h.data(), out, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per_wave_mfma = (double)h[n / 2] / passes / (NG * 3);
    printf("%-58s NV = %d per MFMA: %6.1f cycles per MFMA and wave = %5.1f SIMD cycles per MFMA (floor 32)\n", what, NV, per_wave_mfma, per_wave_mfma / (NT / 256));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int passes = argc > 1 ? atoi(argv[1]) : 200;
    unsigned long long* out; float* sink;
    if (hipMalloc(&out, 4096 * sizeof(unsigned long long)) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    run<0, 0>(passes, out, sink, "one wave per SIMD, bare walk");
    run<0, 3>(passes, out, sink, "one wave per SIMD");
    run<0, 4>(passes, out, sink, "one wave per SIMD (today's epilogue density ~4-5)");
    run<0, 5>(passes, out, sink, "one wave per SIMD");
    run<2, 0>(passes, out, sink, "two waves per SIMD, k split, NO exchange, bare walk");
    run<2, 4>(passes, out, sink, "two waves per SIMD, k split, NO exchange");
    run<2, 5>(passes, out, sink, "two waves per SIMD, k split, NO exchange");
    run<1, 3>(passes, out, sink, "two waves per SIMD, k split, exchange through LDS");
    run<1, 4>(passes, out, sink, "two waves per SIMD, k split, exchange through LDS");
    run<1, 5>(passes, out, sink, "two waves per SIMD, k split, exchange through LDS");
    return 0;
}
