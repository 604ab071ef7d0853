// bf16 mode (BASELINE cfg 4): the register-resident MLP chain on v_mfma_f32_32x32x16_bf16.
//
// Same orientation as the fp32 chain (mlp_core.hpp): weights are the MFMA A operand, a wavefront's samples sit on the
// lanes, and the 32x32 fp32 accumulator of n-tile t — register r of lane (j, h) = feature 32t + (r&3) + 8(r>>2) + 4h of
// sample j — becomes the next layer's B operand after ONE v_cvt_pk_bf16_f32 per register pair: registers 0..7 are the
// 8 elements of k-step 2t, registers 8..15 of k-step 2t+1 (the weights are packed with that k order, tnerf_internal.h).
//
// What changes against fp32 is the feed.  A bf16 MFMA retires 16x the FLOPs per cycle, so a wave needs a 1 KB weight
// fragment every 32 cycles — more than the L2->CU path can deliver to every SIMD.  Therefore the EIGHT waves of a
// workgroup (two per SIMD, one 32-sample tile each, ~230 registers) share one copy of the weights: the fragment STREAM
// (consumption order, 16 KB stages) flows HBM/L2 -> LDS ring (LDS-DMA, global_load_lds_dwordx4: no VGPR round trip)
// -> ds_read_b128 (one per MFMA: half of the LDS bandwidth) -> MFMA.  The second wave of a SIMD fills the MFMA pipe
// while the first does its encoder / epilogue / compositing arithmetic.
// One raw s_barrier per stage publishes the stage after the one being consumed; three stages are in flight behind it,
// tracked with a counted s_waitcnt vmcnt (the DMA is issued from inline asm so that hipcc neither drains it with
// vmcnt(0) at the barrier nor serialises the ds_reads behind it).
//
#pragma once
#include "mlp_core.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define TN16_NS 8                              // LDS ring slots (stages)
#define TN16_SLOT (TN16_STAGE * 1024)          // bytes per stage
#define TN16_RING (TN16_NS * TN16_SLOT)
#define TN16_PF 4                              // A-fragment prefetch distance (ds_read -> MFMA), in fragments
#define TN16_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// two fp32 -> one dword of two bf16 (RNE).  hipcc has no builtin for the PACKED form on gfx950 and scalarises a vector
// fptrunc into 2 cvt + 1 perm; the asm is register-only (inputs are VALU results here, never raw MFMA outputs: the
// MFMA -> VALU hazard is left to compiler-visible instructions).
__device__ __forceinline__ unsigned tn16_cvt2(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// ReLU on a packed bf16 pair: as signed 16-bit integers, negative floats (and -0) are negative.
__device__ __forceinline__ unsigned tn16_relu2(unsigned p) {
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), z));
}

// Per-wave state of the weight stream.
struct Pipe16 {
    // consumer side
    uint32_t lane16;             // lane * 16
    uint32_t va_cur, va_nxt;     // LDS byte offset (ring-relative) + lane16 of the stage being consumed / the one after
    uint32_t nxt_off;            // uniform: ring offset behind va_nxt
    bf16x8 afr[TN16_PF];
    // loader side (all uniform except voff)
    const unsigned char* src;    // packed fragment stream
    uint32_t src_off, stream_bytes;
    uint32_t dst_off;            // ring offset of the slot the next DMA fills
    bool lag;                    // waves 4..7: workgroup barriers are taken half a stage late
    uint32_t lds_dst0;           // absolute LDS address of ring + wave * 2048
    uint32_t voff[2];            // lane * 16 + wave * 2048 + i * 1024
};

__device__ __forceinline__ void tn16_glds(const unsigned char* src, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(src), "s"(lds_dst) : "memory");
}

// DMA this wave's eighth (2 fragments) of the next stage of the stream into the next ring slot.
__device__ __forceinline__ void tn16_issue_stage(Pipe16& p) {
    const unsigned char* s = p.src + p.src_off;
#pragma unroll
    for (int i = 0; i < 2; ++i) tn16_glds(s, p.voff[i], p.lds_dst0 + p.dst_off + i * 1024);
    p.src_off += TN16_SLOT; if (p.src_off == p.stream_bytes) p.src_off = 0;
    p.dst_off += TN16_SLOT; if (p.dst_off == TN16_RING) p.dst_off = 0;
}

// Twice per stage.  The workgroup-wide step — wait for this wave's DMA of stage k+1 (stages k+2, k+3 stay in flight),
// barrier (stage k+1 is now readable by everyone; the slot of stage k-4 is free), issue stage k+4 — is taken by waves
// 0..3 at the start of stage k and by waves 4..7 (p.lag) in the MIDDLE of stage k-1: the two waves of a SIMD thus run
// half a stage (one n-tile of a 256-wide layer) out of phase, and one's epilogue / encoder / compositing arithmetic
// overlaps the other's MFMAs instead of both leaving the pipe idle at the same time.
template <bool MID>
__device__ __forceinline__ void tn16_boundary(Pipe16& p) {
    if (p.lag == MID) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        tn16_issue_stage(p);
    }
    if constexpr (!MID) {
        p.va_cur = p.va_nxt;
        p.nxt_off += TN16_SLOT; if (p.nxt_off == TN16_RING) p.nxt_off = 0;
        p.va_nxt = p.lane16 + p.nxt_off;
    }
}

// One layer for the wave's 32-sample tile.
//   KIND 0: first layer (input k-steps only)   1: hidden   2: skip layer (hidden + input k-steps)   3: heads
// The epilogue of an n-tile (bias add in fp32, round to bf16, ReLU on the packed pair) follows its last MFMA; the MFMA
// pipe is kept busy meanwhile by the SIMD's other wave, which runs half a stage out of phase (tn16_boundary).
// vb: per-lane LDS byte offset of this layer's biases (+ 16 h).  KIND 3 leaves the raw head accumulator in `acc`.
template <int HID, int KIND>
__device__ __forceinline__ void tn16_layer(Pipe16& p, const unsigned char* lds, uint32_t vb,
                                           const bf16x8 (&bin)[HID / 16], const bf16x8 (&enc)[TN16_KE],
                                           bf16x8 (&bout)[HID / 16], f32x16& acc) {
    constexpr int NT = KIND == 3 ? 1 : HID / 32, KH = HID / 16;
    constexpr int KPT = KIND == 0 ? TN16_KE : (KIND == 1 ? KH : (KIND == 2 ? KH + TN16_KE : TN16_STAGE));
    constexpr int KUSE = KIND == 3 ? KH : KPT;                 // k-steps with MFMAs (the head stage is zero-padded)
    static_assert((NT * KPT) % TN16_STAGE == 0, "a layer must be a whole number of stages");
    tn_static_for<NT>([&](auto tc) TN_INLINE_LAMBDA {
        constexpr int t = decltype(tc)::value;
        tn_static_for<KPT>([&](auto sc) TN_INLINE_LAMBDA {
            constexpr int s = decltype(sc)::value;
            constexpr int F = t * KPT + s;
            if constexpr (F % (TN16_STAGE / 2) == 0) tn16_boundary<(F % TN16_STAGE) != 0>(p);
            const bf16x8 afrag = p.afr[F % TN16_PF];
            {
                constexpr int o = (F % TN16_STAGE) + TN16_PF;
                if constexpr (o < TN16_STAGE) p.afr[F % TN16_PF] = *reinterpret_cast<const bf16x8*>(lds + p.va_cur + o * 1024);
                else                          p.afr[F % TN16_PF] = *reinterpret_cast<const bf16x8*>(lds + p.va_nxt + (o - TN16_STAGE) * 1024);
            }
            if constexpr (s < KUSE) {
                bf16x8 b;
                if constexpr (KIND == 0)      b = enc[s];
                else if constexpr (KIND == 2) { if constexpr (s < KH) b = bin[s]; else b = enc[s - KH]; }
                else                          b = bin[s];
                if constexpr (s == 0) { const f32x16 z = {}; acc = TN16_MFMA(afrag, b, z); }
                else                  acc = TN16_MFMA(afrag, b, acc);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (KIND != 3) {
            tn_static_for<2>([&](auto hc) TN_INLINE_LAMBDA {
                constexpr int half = decltype(hc)::value;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(lds + vb + (32 * t + 16 * half) * 4);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(lds + vb + (32 * t + 16 * half + 8) * 4);
                u32x4 w;
                w[0] = tn16_relu2(tn16_cvt2(acc[8 * half + 0] + b0[0], acc[8 * half + 1] + b0[1]));
                w[1] = tn16_relu2(tn16_cvt2(acc[8 * half + 2] + b0[2], acc[8 * half + 3] + b0[3]));
                w[2] = tn16_relu2(tn16_cvt2(acc[8 * half + 4] + b1[0], acc[8 * half + 5] + b1[1]));
                w[3] = tn16_relu2(tn16_cvt2(acc[8 * half + 6] + b1[2], acc[8 * half + 7] + b1[3]));
                bout[2 * t + half] = __builtin_bit_cast(bf16x8, w);
            });
        }
    });
}

// PositionalEncoding(L, include_input=True) of one point as the bf16 B operand of the input k-steps (slot map:
// tnerf_internal.h).  reference src/encoding.py:27-33; sin/cos in fp32 (tn_sincos), rounded to bf16.
__device__ __forceinline__ void tn16_encode(float px, float py, float pz, int Lf, int h, bf16x8 (&enc)[TN16_KE]) {
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
        });
        u32x4 w;
        w[0] = tn16_cvt2(v[0], v[1]); w[1] = tn16_cvt2(v[2], v[3]); w[2] = tn16_cvt2(v[4], v[5]); w[3] = tn16_cvt2(v[6], v[7]);
        enc[u] = __builtin_bit_cast(bf16x8, w);
    });
}
