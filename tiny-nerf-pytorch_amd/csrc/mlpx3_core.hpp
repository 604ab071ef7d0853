// "x3" chain: the fp32 MLP chain on the bf16 matrix pipe with EXACT fp32 products.
//
// Every fp32 operand is the sum of three bf16 numbers (mantissa cut 8 + 8 + 8, each cut exact), a product of two bf16
// numbers is exact in fp32, so six v_mfma_f32_32x32x16_bf16 with fp32 accumulation (a3 b1, a2 b2, a1 b3, a2 b1, a1 b2,
// a1 b1) carry a * b up to terms below 2^-24 |ab|.  The fp32 pipe's chain kernels (mlp_core.hpp) sit at the clock-limited
// ceiling of that pipe; this form needs 6/16 of its matrix time.
//
// Orientation as everywhere in this library: weights = A operand, the wave's 32 samples on the lanes, the 32x32 fp32
// accumulator of n-tile t = the next layer's B operand for k-steps 2t, 2t+1 — here after bias / ReLU in fp32 and ONE exact
// split into three packed bf16 operand registers.  What differs from the bf16 mode (mlp16_core.hpp):
//   * k-step-major order.  A layer is walked k-step by k-step with ALL n-tiles' accumulators live (HID/32 x 16 registers);
//     the three pieces of the input activation X[s] are dead after k-step s, so the layer's output pieces are written back
//     into the same registers: one activation array (3 x HID/16 x 4 registers) instead of an in / out pair.  8x256: 192 + 128
//     registers — one wave per SIMD (512-register budget), four waves per workgroup.
//   * the weight stream carries three pieces per fragment (tnerf_internal.h, NetX3): per k-step record NT x 3 KB through an LDS
//     ring of 24 KB stages (LDS-DMA, counted vmcnt, one raw barrier per stage), shared by the four waves.
#pragma once
#include "mlp16_core.hpp"

#define TX_SLOT (TX_STAGE * 1024)              // bytes per stage
#define TX_NS 5                                // ring slots
#define TX_RING (TX_NS * TX_SLOT)              // 120 KB
#define TX_LEAD 3                              // stages in flight behind the published one (LEAD + 2 <= NS)
#define TX_DPW (TX_STAGE / 4)                  // DMA instructions per wave and stage (4 waves)
static_assert(TX_LEAD + 2 <= TX_NS && TX_STAGE % 4 == 0, "ring budget");

// x (two fp32) -> the three packed bf16 pieces of the pair: dword = (hi16 of piece(x1)) : (hi16 of piece(x0))
__device__ __forceinline__ void tx_split2(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    const uint32_t u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(u0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(u1 & 0xFFFF0000u);
    const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    const float s0 = r0 - __uint_as_float(v0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(v1 & 0xFFFF0000u);
    p1 = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    p2 = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    p3 = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);          // <= 8 significant bits left: exact
}

// Per-wave state of the weight stream (4 waves per workgroup).
struct PipeX {
    uint32_t lane16;             // lane * 16
    uint32_t cur;                // ring byte offset of the stage being consumed
    const unsigned char* src;    // packed record stream
    uint32_t src_off, stream_bytes;
    uint32_t dst_off;            // ring offset of the slot the next DMA fills
    uint32_t lds_dst0;           // absolute LDS address of ring + wave * TX_DPW KB
    uint32_t voff[TX_DPW];       // lane * 16 + (wave * TX_DPW + i) * 1024
};

__device__ __forceinline__ void tx_issue_stage(PipeX& p) {
    const unsigned char* s = p.src + p.src_off;
#pragma unroll
    for (int i = 0; i < TX_DPW; ++i) tn_glds16(s, p.voff[i], p.lds_dst0 + p.dst_off + i * 1024);
    p.src_off += TX_SLOT; if (p.src_off == p.stream_bytes) p.src_off = 0;
    p.dst_off += TX_SLOT; if (p.dst_off == TX_RING) p.dst_off = 0;
}

// Start of a stage: wait for this wave's DMA of the stage (LEAD-1 younger ones may stay in flight; STORES more operations
// are allowed to be outstanding — the training kernels interleave global stores with the stream, see mlp16_core.hpp),
// barrier (the stage is readable by everyone, the slot of the previous one is free), issue stage + LEAD.
template <int STORES>
__device__ __forceinline__ void tx_boundary(PipeX& p) {
    TN16_WAIT_VM(TX_DPW * (TX_LEAD - 1) + STORES);
    __builtin_amdgcn_s_barrier();
    tx_issue_stage(p);
    p.cur += TX_SLOT; if (p.cur == TX_RING) p.cur = 0;
}

// Workgroup prologue: biases -> LDS, LEAD stages in flight, the first one landed; the first tx_boundary publishes stage 0.
// `src` / `n_stage`: the stream this kernel walks (forward: packed, n.n_stage; dgrad: the backward stream behind it).
__device__ __forceinline__ void tx_prologue(PipeX& p, unsigned char* lds, const unsigned char* packed, const NetX3& n,
                                            const unsigned char* src, int n_stage, int lane, int wave) {
    {
        float* bl = reinterpret_cast<float*>(lds + TX_RING);
        const float* bg = reinterpret_cast<const float*>(packed + n.bias_off);
        for (int i = threadIdx.x; i < n.n_bias; i += 256) bl[i] = bg[i];
    }
    p.lane16 = lane * 16;
    p.src = src; p.src_off = 0; p.stream_bytes = (uint32_t)n_stage * TX_SLOT;
    p.dst_off = 0;
    p.lds_dst0 = (uint32_t)(uintptr_t)lds + wave * (TX_DPW * 1024);
#pragma unroll
    for (int i = 0; i < TX_DPW; ++i) p.voff[i] = lane * 16 + (wave * TX_DPW + i) * 1024;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < TX_LEAD; ++i) tx_issue_stage(p);
    p.cur = TX_RING - TX_SLOT;                   // the first boundary moves it onto slot 0
}

// The activation of a wave's 32-sample tile: three bf16 pieces of every k-step operand.
template <int HID>
struct ActX { bf16x8 p1[HID / 16], p2[HID / 16], p3[HID / 16]; };
struct EncX { bf16x8 p1[TN16_KE], p2[TN16_KE], p3[TN16_KE]; };

// Six exact partial products of one (n-tile, k-step): small terms first.  FIRST: the accumulator starts at zero.
template <bool FIRST>
__device__ __forceinline__ void tx_mfma6(f32x16& acc, const bf16x8& a1, const bf16x8& a2, const bf16x8& a3,
                                         const bf16x8& b1, const bf16x8& b2, const bf16x8& b3) {
    if constexpr (FIRST) { const f32x16 z = {}; acc = TN16_MFMA(a3, b1, z); }
    else                 acc = TN16_MFMA(a3, b1, acc);
    acc = TN16_MFMA(a2, b2, acc);
    acc = TN16_MFMA(a1, b3, acc);
    acc = TN16_MFMA(a2, b1, acc);
    acc = TN16_MFMA(a1, b2, acc);
    acc = TN16_MFMA(a1, b1, acc);
}

// The records of one layer: KIND 0: input k-steps only   1: hidden   2: hidden, then input (skip layer)   3: heads (tile 0 only)
// 4: heads^T of the backward stream (ONE k-step whose B operand is E.p*[0]; the stage is padded with empty records).
// RPS = records per stage (1 for 256-wide, 2 for 128-wide nets); every layer is a whole number of stages (4, 8, 12, 16, 20
// records), so the stage phase of record k of a layer is k % RPS.
template <int HID, int KIND, int STORES>
__device__ __forceinline__ void tx_layer_mfma(PipeX& p, const unsigned char* lds, const ActX<HID>& X, const EncX& E,
                                              f32x16 (&acc)[HID / 32]) {
    constexpr int NT = HID / 32, KH = HID / 16, RPS = TX_STAGE / (NT * 3);
    constexpr int NK = KIND == 0 ? TN16_KE : (KIND == 2 ? KH + TN16_KE : (KIND == 4 ? 1 : KH));
    constexpr int NTU = KIND == 3 ? 1 : NT;                        // tiles with MFMAs
    static_assert(NK % RPS == 0 || KIND == 4, "a layer must be a whole number of stages");
    tn_static_for<NK>([&](auto kc) TN_INLINE_LAMBDA {
        constexpr int k = decltype(kc)::value;
        if constexpr (k % RPS == 0) tx_boundary<STORES>(p);
        const unsigned char* base = lds + p.cur + (k % RPS) * (NT * 3 * 1024) + p.lane16;
        bf16x8 b1, b2, b3;
        if constexpr (KIND == 0 || KIND == 4 || (KIND == 2 && k >= KH)) { constexpr int u = (KIND == 0 || KIND == 4) ? k : k - KH; b1 = E.p1[u]; b2 = E.p2[u]; b3 = E.p3[u]; }
        else { b1 = X.p1[k]; b2 = X.p2[k]; b3 = X.p3[k]; }
        tn_static_for<NTU>([&](auto tc) TN_INLINE_LAMBDA {
            constexpr int t = decltype(tc)::value;
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(base + (t * 3 + 0) * 1024);
            const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(base + (t * 3 + 1) * 1024);
            const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(base + (t * 3 + 2) * 1024);
            tx_mfma6<k == 0>(acc[t], a1, a2, a3, b1, b2, b3);
        });
    });
}

// Epilogue of a hidden layer: bias (fp32, from LDS), ReLU, exact split back into the activation registers.
// vb: per-lane LDS byte offset of this layer's biases (+ 16 h).  `fin(t, v)` sees the 16 fp32 outputs of tile t (training stash).
template <int HID, typename Fin>
__device__ __forceinline__ void tx_layer_epilogue(const unsigned char* lds, uint32_t vb, const f32x16 (&acc)[HID / 32], ActX<HID>& X, Fin&& fin) {
    constexpr int NT = HID / 32;
    tn_static_for<NT>([&](auto tc) TN_INLINE_LAMBDA {
        constexpr int t = decltype(tc)::value;
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lds + vb + (32 * t + 8 * q) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * q + i] = fmaxf(acc[t][4 * q + i] + b[i], 0.0f);
        }
        fin(tc, v);
        u32x4 w1[2], w2[2], w3[2];
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned a_, b_, c_;
                tx_split2(v[8 * half + 2 * q], v[8 * half + 2 * q + 1], a_, b_, c_);
                w1[half][q] = a_; w2[half][q] = b_; w3[half][q] = c_;
            }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            X.p1[2 * t + half] = __builtin_bit_cast(bf16x8, w1[half]);
            X.p2[2 * t + half] = __builtin_bit_cast(bf16x8, w2[half]);
            X.p3[2 * t + half] = __builtin_bit_cast(bf16x8, w3[half]);
        }
    });
}

// Epilogue of a backward layer: ReLU backward with the forward's sign bits (mw: the words of the layer whose activation gradient
// this is), exact split back into the activation registers.  `fin(t, v)` sees the 16 fp32 values dZ of tile t (stash).
template <int HID, typename Fin>
__device__ __forceinline__ void tx_layer_epilogue_bwd(const f32x16 (&acc)[HID / 32], const uint32_t (&mw)[HID / 64], ActX<HID>& X, Fin&& fin) {
    constexpr int NT = HID / 32;
    tn_static_for<NT>([&](auto tc) TN_INLINE_LAMBDA {
        constexpr int t = decltype(tc)::value;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            v[r] = __int_as_float(__float_as_int(acc[t][r]) & __builtin_amdgcn_sbfe((int)mw[t / 2], (t & 1) * 16 + r, 1));
        fin(tc, v);
        u32x4 w1[2], w2[2], w3[2];
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned a_, b_, c_;
                tx_split2(v[8 * half + 2 * q], v[8 * half + 2 * q + 1], a_, b_, c_);
                w1[half][q] = a_; w2[half][q] = b_; w3[half][q] = c_;
            }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            X.p1[2 * t + half] = __builtin_bit_cast(bf16x8, w1[half]);
            X.p2[2 * t + half] = __builtin_bit_cast(bf16x8, w2[half]);
            X.p3[2 * t + half] = __builtin_bit_cast(bf16x8, w3[half]);
        }
    });
}

// PositionalEncoding(L, include_input=True) of one point, fp32-accurate (tn_sincos, as the fp32 kernels), in the slot map of
// the input k-steps (tnerf_internal.h), split into three pieces.                         reference src/encoding.py:27-33
// `out(st, value)`: the fp32 value of input step st = 8u + e (the step numbering of the fp32 path's pairing: the training stash).
template <typename Out>
__device__ __forceinline__ void tx_encode(float px, float py, float pz, int Lf, int h, EncX& E, Out&& out) {
    tn_static_for<TN16_KE>([&](auto uc) TN_INLINE_LAMBDA {
        constexpr int u = decltype(uc)::value;
        float v[8];
        tn_static_for<8>([&](auto ec) TN_INLINE_LAMBDA {
            constexpr int e = decltype(ec)::value;
            constexpr int a = 8 * u + e, k = a / 3, c = a % 3;
            const float pc = c == 0 ? px : (c == 1 ? py : pz);
            float r = 0.0f;
            if (a < 3 * Lf) {
                float sn, cs;
                tn_sincos(pc * (float)(1u << (k < 31 ? k : 0)), sn, cs);
                r = h ? cs : sn;
            } else if (a == 3 * Lf) {
                r = h ? py : px;
            } else if (a == 3 * Lf + 1) {
                r = h ? 0.0f : pz;
            }
            v[e] = r;
            out(std::integral_constant<int, a>{}, r);
        });
        u32x4 w1, w2, w3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned a_, b_, c_;
            tx_split2(v[2 * q], v[2 * q + 1], a_, b_, c_);
            w1[q] = a_; w2[q] = b_; w3[q] = c_;
        }
        E.p1[u] = __builtin_bit_cast(bf16x8, w1); E.p2[u] = __builtin_bit_cast(bf16x8, w2); E.p3[u] = __builtin_bit_cast(bf16x8, w3);
    });
}
