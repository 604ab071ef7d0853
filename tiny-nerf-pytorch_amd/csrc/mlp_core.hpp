// Register-resident MLP chain on fp32 MFMA (v_mfma_f32_32x32x2_f32), shared by the forward and
// backward kernels.
//
// Orientation ("sample on the lane"): a wavefront owns a tile of 32 samples and computes
//     H_out^T[n][m] = sum_k W[n][k] * H_in^T[k][m]          (n: feature row, m: sample column)
// with the WEIGHTS as the MFMA A operand and the ACTIVATIONS as the B operand.  The 32x32
// accumulator of n-tile t then holds, in register r of lane (j = lane&31, h = lane>>5),
//     H_out[feature 32t + (r&3) + 8(r>>2) + 4h][sample j]
// which is exactly what the next layer's B operand wants for the k-pair (k, k+4): the accumulator
// registers feed the next MFMA directly — no LDS round trip, no cross-lane traffic, no conversion.
// A wave keeps the whole activation vector of its 32 samples in HID/2 registers (+ HID/2
// accumulators), so the only memory traffic of the chain is the weight stream (L2-resident,
// pre-packed in fragment order by tnerf_mlp_pack so that every A load is a lane-linear 16-byte
// load) and, when training, the activation stash.
#pragma once
#include <utility>
#include <type_traits>
#include "dev_common.hpp"

#ifndef TN_PFDIST
#define TN_PFDIST 6
#endif
// The training stash (2.2 GB per kernel at cfg 2) is written once and read once, by the next kernel but one.
#ifdef TN_STASH_TEMPORAL
#define TN_STASH_STORE(ptr, v) (*(ptr) = (v))
#else
#define TN_STASH_STORE(ptr, v) __builtin_nontemporal_store((v), (ptr))
#endif
#define TN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>).  The hot loops
// below must be unrolled in the FRONT END (every register-array index a constant before the first
// SROA pass): left to `#pragma unroll`, the 256-wide layers exceed LLVM's unroll thresholds, the
// activation arrays stay allocas and the kernel runs out of scratch memory instead of registers.
template <int... Is, typename F>
__device__ __forceinline__ void tn_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void tn_static_for(F&& f) {
    tn_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}
#define TN_INLINE_LAMBDA __attribute__((always_inline))

// Weight fragments are fetched with BUFFER loads: descriptor in SGPRs, per-lane 32-bit offset (lane*16), the fragment's
// position as a scalar offset.  Measured (tools/mfma_feed_probe.hip): one such load per 4 dependent MFMAs is free
// (64.0 cycles/MFMA), while a global_load with a 64-bit per-lane address costs the wave ~29 cycles of issue that no MFMA
// overlaps (71.2 cycles/MFMA) at one wave per SIMD.  (Stash STORES were tried as bounds-checked buffer stores too:
// correct, 3 % slower than global stores with immediate row offsets, so they stay global.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tn_packed_rsrc(const float* packed, int64_t packed_floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(packed), 0, (int)(packed_floats * 4), 0x00020000);
}
__device__ __forceinline__ f32x4 tn_frag_load(__amdgpu_buffer_rsrc_t rsrc, int voff_bytes, int soff_bytes) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_bytes, soff_bytes, 0));
}

// One dense layer for a 32-sample tile.
//   HAS_HID: the layer consumes the previous hidden activations `hin` (HID/2 regs per lane)
//   HAS_ENC: the layer consumes the network input `enc` (NE regs per lane; layer 0 and the skip layer)
// `fin(integral_constant<t>, acc)` is called once per finished n-tile with the pre-activation accumulator.
template <int HID, int NE, bool HAS_HID, bool HAS_ENC, typename Fin>
__device__ __forceinline__ void tn_layer(__amdgpu_buffer_rsrc_t rsrc, int64_t off_bias, int64_t off_enc,
                                         int64_t off_hid, const float (&hin)[HID / 2], const float (&enc)[NE],
                                         int lane, Fin&& fin) {
    constexpr int NT = HID / 32;
    constexpr int GH = HAS_HID ? NT * 4 : 0;   // 4-step groups per n-tile fed by hidden activations
    constexpr int GE = HAS_ENC ? NE / 4 : 0;   // ... fed by the network input
    constexpr int GT = GH + GE;
    constexpr int TOTAL = NT * GT;
    constexpr int PF = TN_PFDIST;               // A-fragment prefetch distance (groups of 4 MFMAs = 256 cycles each)
    const int vlane = lane * 16, vbias = (lane >> 5) * 64;                      // per-lane byte offsets
    const int sh = (int)((HAS_HID ? off_hid : 0) * 4), se = (int)((HAS_ENC ? off_enc : 0) * 4), sb = (int)(off_bias * 4);
    auto Wh = [&](int idx) TN_INLINE_LAMBDA { return tn_frag_load(rsrc, vlane, sh + idx * 1024); };   // group idx of the hidden part
    auto We = [&](int idx) TN_INLINE_LAMBDA { return tn_frag_load(rsrc, vlane, se + idx * 1024); };
    auto Bf = [&](int idx) TN_INLINE_LAMBDA { return tn_frag_load(rsrc, vbias, sb + idx * 16); };

    f32x4 ring[PF];
    tn_static_for<PF>([&](auto ic) TN_INLINE_LAMBDA {
        constexpr int i = decltype(ic)::value;
        constexpr int t = i / GT, g = i % GT;
        if constexpr (i < TOTAL) {
            if constexpr (g < GH) ring[i] = Wh(t * GH + g);
            else                  ring[i] = We(t * GE + (g - GH));
        }
    });
    // The accumulator starts from the bias.  The bias of n-tile t+1 is fetched behind the first MFMA group of tile t:
    // loaded at the top of its own tile it cost the full L2 latency 8 times per layer (nothing else can run at one wave
    // per SIMD) — ~12 % of the kernel.
    f32x4 bn0 = Bf(0), bn1 = Bf(1), bn2 = Bf(2), bn3 = Bf(3);
    tn_static_for<NT>([&](auto tc) TN_INLINE_LAMBDA {
        constexpr int t = decltype(tc)::value;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc[r] = bn0[r]; acc[4 + r] = bn1[r]; acc[8 + r] = bn2[r]; acc[12 + r] = bn3[r]; }
        tn_static_for<GT>([&](auto gc) TN_INLINE_LAMBDA {
            constexpr int g = decltype(gc)::value;
            constexpr int i = t * GT + g;
            if constexpr (g == 1 && t + 1 < NT) {
                bn0 = Bf((t + 1) * 8 + 0); bn1 = Bf((t + 1) * 8 + 1); bn2 = Bf((t + 1) * 8 + 2); bn3 = Bf((t + 1) * 8 + 3);
            }
            const f32x4 a4 = ring[i % PF];
            if constexpr (i + PF < TOTAL) {
                constexpr int t2 = (i + PF) / GT, g2 = (i + PF) % GT;
                if constexpr (g2 < GH) ring[i % PF] = Wh(t2 * GH + g2);
                else                   ring[i % PF] = We(t2 * GE + (g2 - GH));
            }
            if constexpr (g < GH) {
                acc = TN_MFMA(a4[0], hin[g * 4 + 0], acc); acc = TN_MFMA(a4[1], hin[g * 4 + 1], acc);
                acc = TN_MFMA(a4[2], hin[g * 4 + 2], acc); acc = TN_MFMA(a4[3], hin[g * 4 + 3], acc);
            } else {
                constexpr int e = (g - GH) * 4;
                acc = TN_MFMA(a4[0], enc[e + 0], acc); acc = TN_MFMA(a4[1], enc[e + 1], acc);
                acc = TN_MFMA(a4[2], enc[e + 2], acc); acc = TN_MFMA(a4[3], enc[e + 3], acc);
            }
            // Pin the order {fragment load for group i+PF ; 4 MFMAs of group i}: left alone, the scheduler sinks the
            // loads next to their use (2 in flight, ~300 cycles ahead) and every group stalls on the L2 round trip.
            __builtin_amdgcn_sched_barrier(0);
        });
        fin(tc, acc);
    });
}

// Transposed layer for the backward chain: dH_in^T[k][m] = sum_n W[n][k] dZ^T[n][m]
// (A operand = W^T fragments packed at off_bw, B operand = dz registers).
template <int HID, typename Fin>
__device__ __forceinline__ void tn_layer_bwd(__amdgpu_buffer_rsrc_t rsrc, int64_t off_bw,
                                             const float (&dz)[HID / 2], int lane, Fin&& fin) {
    constexpr int NT = HID / 32;
    constexpr int GT = NT * 4;
    constexpr int TOTAL = NT * GT;
    constexpr int PF = TN_PFDIST;
    const int vlane = lane * 16, sw = (int)(off_bw * 4);
    auto Wt = [&](int idx) TN_INLINE_LAMBDA { return tn_frag_load(rsrc, vlane, sw + idx * 1024); };
    f32x4 ring[PF];
    tn_static_for<PF>([&](auto ic) TN_INLINE_LAMBDA { constexpr int i = decltype(ic)::value; ring[i] = Wt(i); });
    tn_static_for<NT>([&](auto tc) TN_INLINE_LAMBDA {
        constexpr int t = decltype(tc)::value;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        tn_static_for<GT>([&](auto gc) TN_INLINE_LAMBDA {
            constexpr int g = decltype(gc)::value;
            constexpr int i = t * GT + g;
            const f32x4 a4 = ring[i % PF];
            if constexpr (i + PF < TOTAL) ring[i % PF] = Wt(i + PF);
            acc = TN_MFMA(a4[0], dz[g * 4 + 0], acc); acc = TN_MFMA(a4[1], dz[g * 4 + 1], acc);
            acc = TN_MFMA(a4[2], dz[g * 4 + 2], acc); acc = TN_MFMA(a4[3], dz[g * 4 + 3], acc);
            __builtin_amdgcn_sched_barrier(0);
        });
        fin(tc, acc);
    });
}

// Network input for a tile whose points are generated in-kernel: PositionalEncoding(L, True)
// (reference src/encoding.py:27-33) in the sin/cos pairing of tnerf_input_pairing: step 3k+c holds
// sin(2^k p_c) on lane-half 0 and cos(2^k p_c) on lane-half 1; steps 3L, 3L+1 hold (x, y), (z, 0).
template <int NE>
__device__ __forceinline__ void tn_encode_point(float px, float py, float pz, int L, int h, float (&enc)[NE]) {
    tn_static_for<NE>([&](auto sc_) TN_INLINE_LAMBDA {
        constexpr int st = decltype(sc_)::value;
        constexpr int k = st / 3, c = st % 3;
        const float pc = c == 0 ? px : (c == 1 ? py : pz);
        float v = 0.0f;
        if (st < 3 * L) {
            float sn, cs;
            tn_sincos(pc * (float)(1u << k), sn, cs);
            v = h ? cs : sn;
        } else if (st == 3 * L) {
            v = h ? py : px;
        } else if (st == 3 * L + 1) {
            v = h ? 0.0f : pz;
        }
        enc[st] = v;
    });
}

// Network input for a tile read from x[M, in_dim] (TinyNeRF.forward called on its own).
template <int NE>
__device__ __forceinline__ void tn_load_input(const float* __restrict__ x, int64_t m, bool valid, const MlpLayout& L,
                                              int h, float (&enc)[NE]) {
    tn_static_for<NE>([&](auto sc_) TN_INLINE_LAMBDA {
        constexpr int st = decltype(sc_)::value;
        const int c = h ? L.emap[st][1] : L.emap[st][0];
        enc[st] = (valid && c >= 0) ? x[m * L.in_dim + c] : 0.0f;
    });
}
