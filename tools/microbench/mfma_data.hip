// Microbenchmark (round 4): v_mfma_f32_32x32x16_f16 takes up to 2.3x longer when its operands carry "busy" data — on 8 CUs as on 256,
// at an unchanged 2.4 GHz shader clock: not a chip-level power cap but something local to the matrix pipe (an issue throttle driven
// by operand activity).  This probe walks the chain kernels' bare half-pass (pass_mfma.hip: 64 groups x 3 MFMAs, A fragments from
// LDS, B from registers, a barrier per 8 groups) over operand sets described at run time:
//   per operand (A = weights from LDS, B = activations in registers): sign (0 all +, 1 random), exponent spread (biased exponent in
//   [15 - e, 15]), mantissa bits kept (10 = all random, fewer = low bits zero), fraction of exact zeros in 1/8ths.
//   mfma_data [passes] [grid]
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
struct Mode { int sign, espread, mbits, zeros8, small2; };     // small2: the second piece is 2^-11 of the first (as a split's residual)
__device__ __forceinline__ unsigned rnd(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ unsigned half_of(unsigned r, const Mode& m, int piece) {
    if ((int)((r >> 28) & 7) < m.zeros8) return 0u;
    const unsigned man = (r & 0x3ffu) & ~((1u << (10 - m.mbits)) - 1u);
    int e = 15 - (m.espread ? (int)((r >> 10) % (unsigned)(m.espread + 1)) : 0);
    if (piece == 1 && m.small2) e -= 11;
    if (e < 1) e = 1;
    return ((m.sign ? (r >> 27) & 1u : 0u) << 15) | ((unsigned)e << 10) | man;
}
__device__ __forceinline__ unsigned word_of(unsigned seed, const Mode& m, int piece) {
    const unsigned r1 = rnd(seed), r2 = rnd(seed ^ 0x9e3779b9u);
    return half_of(r1, m, piece) | (half_of(r2, m, piece) << 16);
}

__global__ __launch_bounds__(256, 1) void k(int passes, unsigned long long* out, float* sink, Mode ma, Mode mb) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // LDS: 7 stages x 8 groups x [piece 0 | piece 1] x 1 KB
    for (int i = threadIdx.x; i < 28 * 1024; i += 256) reinterpret_cast<unsigned*>(lds)[i] = word_of(i * 2654435761u + 17u, ma, (i >> 8) & 1);
    __syncthreads();
    u32x4 X1[16], X2[16];
    sfor<16>([&](auto ic) __attribute__((always_inline)) { sfor<4>([&](auto qc) __attribute__((always_inline)) {      // static indices: the arrays must stay in registers
        constexpr int i = decltype(ic)::value, q = decltype(qc)::value;
        const unsigned seed = (i * 4 + q) * 40503u + threadIdx.x * 2246822519u + blockIdx.x * 7919u;
        Mode m0 = mb; m0.zeros8 = 0;                           // both pieces of a value are zero together (ReLU)
        unsigned w1 = word_of(seed, m0, 0), w2 = word_of(seed + 1u, m0, 1);
        const unsigned z = rnd(seed + 99u);
        const unsigned keep = (((int)(z & 7) < mb.zeros8) ? 0u : 0xffffu) | (((int)((z >> 3) & 7) < mb.zeros8) ? 0u : 0xffff0000u);
        X1[i][q] = w1 & keep; X2[i][q] = w2 & keep;
    }); });
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    uint32_t cur = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int p = 0; p < passes; ++p) {
        f16x8 fa1[2], fa2[2];
        auto load = [&](int slot, int g) __attribute__((always_inline)) {
            const unsigned char* b = lds + cur + lane * 16 + (g % 8) * 2048;
            fa1[slot] = *reinterpret_cast<const f16x8*>(b); fa2[slot] = *reinterpret_cast<const f16x8*>(b + 1024);
        };
        load(0, 0);
        sfor<64>([&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value, ks = g / 4, tl = g % 4;
            if constexpr (g % 8 == 0 && g > 0) { __builtin_amdgcn_s_barrier(); cur += 16384; if (cur == 7 * 16384) cur = 0; }
            const f16x8 a1 = fa1[g % 2], a2 = fa2[g % 2];
            const f16x8 b1 = H8(X1[ks]), b2 = H8(X2[ks]);
            acc[tl + 4] = MF(a2, b1, acc[tl + 4]);
            if constexpr (g + 1 < 64) load((g + 1) % 2, g + 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[tl + 4] = MF(a1, b2, acc[tl + 4]);
            __builtin_amdgcn_sched_barrier(0);
            acc[tl] = MF(a1, b1, acc[tl]);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = t1 - t0; out[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

static void run(int passes, int grid, unsigned long long* out, float* sink, Mode ma, Mode mb, const char* what) {
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 120 * 1024, 0, passes, out, sink, ma, mb);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", what); exit(2); }
    }
    std::vector<unsigned long long> h(2048);
    (void)hipMemcpy(h.data(), out, sizeof(unsigned long long) * 2048, hipMemcpyDeviceToHost);
    std::vector<double> c, r;
    for (int i = 0; i < grid * 4; ++i) { c.push_back((double)h[2 * i]); r.push_back((double)h[2 * i + 1]); }
    std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
    const double cyc = c[c.size() / 2] / passes, us = r[r.size() / 2] / 100.0 / passes;
    printf("A{s%d e%d m%2d z%d/8 r%d} B{s%d e%d m%2d z%d/8 r%d}  %-44s %6.0f cycles/pass = %5.1f per MFMA  clock %.2f GHz\n", ma.sign, ma.espread, ma.mbits, ma.zeros8, ma.small2,
           mb.sign, mb.espread, mb.mbits, mb.zeros8, mb.small2, what, cyc, cyc / 192.0, cyc / us * 1e-3);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int passes = argc > 1 ? atoi(argv[1]) : 300, grid = argc > 2 ? atoi(argv[2]) : 64;
    unsigned long long* out; float* sink;
    if (hipMalloc(&out, 2048 * sizeof(unsigned long long)) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const Mode C{0, 0, 0, 0, 0}, R{1, 5, 10, 0, 0};
    run(passes, grid, out, sink, C, C, "constant 1.0 x constant 1.0");
    run(passes, grid, out, sink, R, R, "random x random");
    run(passes, grid, out, sink, R, C, "random A x constant B");
    run(passes, grid, out, sink, C, R, "constant A x random B");
    run(passes, grid, out, sink, Mode{1, 0, 0, 0, 0}, Mode{1, 0, 0, 0, 0}, "random signs only");
    run(passes, grid, out, sink, Mode{0, 5, 0, 0, 0}, Mode{0, 5, 0, 0, 0}, "random exponents only");
    run(passes, grid, out, sink, Mode{0, 0, 10, 0, 0}, Mode{0, 0, 10, 0, 0}, "random mantissas only");
    run(passes, grid, out, sink, Mode{0, 0, 10, 0, 0}, C, "random mantissas A only");
    for (int mb = 2; mb <= 8; mb += 2) run(passes, grid, out, sink, Mode{1, 5, mb, 0, 0}, Mode{1, 5, mb, 0, 0}, "random, fewer mantissa bits");
    run(passes, grid, out, sink, Mode{1, 5, 7, 0, 0}, Mode{1, 5, 7, 0, 0}, "random, 7 mantissa bits (bf16-like)");
    for (int z = 1; z <= 6; ++z) run(passes, grid, out, sink, R, Mode{0, 5, 10, z, 0}, "random A x B >= 0 with zeros");
    run(passes, grid, out, sink, Mode{1, 5, 10, 0, 1}, Mode{0, 5, 10, 4, 1}, "split pieces: residuals 2^-11, B >= 0 half zero");
    run(passes, grid, out, sink, Mode{1, 5, 10, 0, 1}, Mode{0, 5, 10, 0, 1}, "split pieces: residuals 2^-11, B >= 0 dense");
    run(passes, grid, out, sink, Mode{1, 5, 10, 0, 1}, Mode{1, 5, 10, 4, 1}, "split pieces, B signed, half zero (dgrad)");
    run(passes, grid, out, sink, Mode{1, 2, 10, 0, 1}, Mode{0, 2, 10, 4, 1}, "split pieces, narrow exponents, B half zero");
    return 0;
}
