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
// while the first does its encoder / epilogue / compositing arithmetic or waits at the barrier.
// One raw s_barrier per stage publishes the stage after the one being consumed; TN16_LEAD more stages are in flight
// behind it, tracked with a counted s_waitcnt vmcnt (the DMA is issued from inline asm — tn_glds16, dev_common.hpp —
// so that hipcc neither drains it with vmcnt(0) at the barrier nor serialises the ds_reads behind it).
#pragma once
#include "mlp_core.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define TN16_NS 8                              // LDS ring slots (stages)
#define TN16_SLOT (TN16_STAGE * 1024)          // bytes per stage
#define TN16_RING (TN16_NS * TN16_SLOT)
#ifndef TN16_PF
#define TN16_PF 2                              // A-fragment prefetch distance (ds_read -> MFMA), in fragments.  4 measured the same
                                               // speed (round 3) but cost 8 registers the 256-wide kernels do not have: they spilled
#endif
#ifndef TN16_LEAD
#define TN16_LEAD 6                            // stages in flight behind the published one (LEAD + 2 <= TN16_NS)
#endif
#ifndef TN16_MIN_STORES
#define TN16_MIN_STORES 8                      // lower bound on the stores a training kernel issues in LEAD consecutive stages
#endif
#define TN16_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")
static_assert(TN16_LEAD + 2 <= TN16_NS && 2 * TN16_LEAD + TN16_MIN_STORES <= 63, "ring / vmcnt budget");
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
    uint32_t lds_dst0;           // absolute LDS address of ring + wave * 2048
    uint32_t voff[2];            // lane * 16 + wave * 2048 + i * 1024
};

// DMA this wave's eighth (2 fragments) of the next stage of the stream into the next ring slot.
__device__ __forceinline__ void tn16_issue_stage(Pipe16& p) {
    const unsigned char* s = p.src + p.src_off;
#pragma unroll
    for (int i = 0; i < 2; ++i) tn_glds16(s, p.voff[i], p.lds_dst0 + p.dst_off + i * 1024);
    p.src_off += TN16_SLOT; if (p.src_off == p.stream_bytes) p.src_off = 0;
    p.dst_off += TN16_SLOT; if (p.dst_off == TN16_RING) p.dst_off = 0;
}

// Start of stage k: wait for this wave's DMA of stage k+1 (stages k+2 .. k+LEAD stay in flight), barrier (stage k+1
// is now readable by everyone; the slot of stage k-1 is free), issue stage k+LEAD+1.
// STORES: the caller interleaves global stores (training stash) with the stream.  vmcnt retires in issue order and a
// store stays counted until it reaches L2, so waiting for the DMA of stage k+1 also waits for every older store: the
// DMA therefore runs TN16_LEAD stages ahead (its wait then only covers stores that are microseconds old), and the
// wait leaves TN16_MIN_STORES more operations outstanding — fewer than the stores any LEAD consecutive stages of the
// training kernels issue (>= 2 per n-tile).
// (Measured and dropped: waves 4..7 taking the barrier half a stage late so that the two waves of a SIMD run out of
// phase — 2 % slower for inference, 6 % for dgrad.  Ablation of the inference kernel, 0.259 ms: barrier 0.020,
// DMA issue + vmcnt wait 0.029.)
template <bool STORES>
__device__ __forceinline__ void tn16_boundary(Pipe16& p) {
    if constexpr (STORES) TN16_WAIT_VM(2 * (TN16_LEAD - 1) + TN16_MIN_STORES);
    else                  TN16_WAIT_VM(2 * (TN16_LEAD - 1));
    __builtin_amdgcn_s_barrier();
    tn16_issue_stage(p);
    p.va_cur = p.va_nxt;
    p.nxt_off += TN16_SLOT; if (p.nxt_off == TN16_RING) p.nxt_off = 0;
    p.va_nxt = p.lane16 + p.nxt_off;
}

// Where a training kernel puts the current tile's records (layout: tnerf_internal.h).
struct Stash16 {              // all wave-uniform: the lane term is added as a 32-bit offset (scalar-base addressing)
    unsigned char* frag;      // fragment region + (tile * n_ft) * 2048
    unsigned char* mask;      // mask region + tile * 64 * (hidden/64) * 4                (layer 0)
    int64_t mask_lstride;     // bytes between layers
};
#define TN16_SEL_OFF(n_bias) (TN16_RING + (uint32_t)(((n_bias) + 3) / 4 * 4) * 4)   // LDS byte offset of the selector pair

// Selector B operands of the transposing MFMAs (2 x 64 lanes x 16 B in LDS): sel_u[k-slot (h,e)][col c] = (c == TN_ACC_ROW(8u+e, h)).
// A x sel_0 + A' x sel_1 turns two packed k-steps (lane = sample, element = feature slot) into a 32 x 32 tile with the
// FEATURE on the lane and the 32 samples in the 16 accumulator registers: the K = samples operand of the weight-gradient
// MFMAs.  The products are exact (x * 1 + 0), so the transposed tile holds the very bf16 values the forward used.
__device__ __forceinline__ void tn16_write_selectors(unsigned char* lds, uint32_t sel_off, int lane) {
    const int c = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        unsigned short v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (c == TN_ACC_ROW(8 * u + e, hh)) ? (unsigned short)0x3F80 : (unsigned short)0;
        u32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (unsigned)v[2 * i] | ((unsigned)v[2 * i + 1] << 16);
        *reinterpret_cast<u32x4*>(lds + sel_off + u * 1024 + lane * 16) = w;
    }
}

// Transpose the k-step pair (b0, b1) and store it as feature tile `ft` of the current tile's stash record.
__device__ __forceinline__ void tn16_stash_tile(const unsigned char* lds, uint32_t sel_off, uint32_t lane16, const Stash16& st, int ft,
                                                const bf16x8& b0, const bf16x8& b1, f32x16& d) {
    const bf16x8 s0 = *reinterpret_cast<const bf16x8*>(lds + sel_off + lane16);
    const bf16x8 s1 = *reinterpret_cast<const bf16x8*>(lds + sel_off + 1024 + lane16);
    const f32x16 z = {};
    d = TN16_MFMA(b0, s0, z);
    d = TN16_MFMA(b1, s1, d);
    // (vector elements are copied to scalars first: __builtin_bit_cast applied directly to d[i] reads element 0)
    u32x4 p0, p1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                   // exact: the values are bf16-representable
        const float l0 = d[2 * i], h0 = d[2 * i + 1], l1 = d[8 + 2 * i], h1 = d[8 + 2 * i + 1];
        p0[i] = __builtin_amdgcn_perm(__float_as_uint(h0), __float_as_uint(l0), 0x07060302u);
        p1[i] = __builtin_amdgcn_perm(__float_as_uint(h1), __float_as_uint(l1), 0x07060302u);
    }
    unsigned char* dst = st.frag + (int64_t)ft * TN16_FT_BYTES;       // uniform
    // non-temporal: 1.1 GB per kernel that nothing re-reads before the weight-gradient kernel (measured: -12 % step time)
    __builtin_nontemporal_store(p0, reinterpret_cast<u32x4*>(dst + lane16));
    __builtin_nontemporal_store(p1, reinterpret_cast<u32x4*>(dst + 1024 + lane16));
}

// 16 ReLU sign bits of an n-tile from its packed outputs (dword q of lo|hi = registers 2q, 2q+1): bit r <-> register r.
// The halves are non-negative bf16 (<= 0x7FFF) after the ReLU, so half + 0x7FFF carries into its bit 15 exactly when
// the half is non-zero, and never into the neighbouring half.
__device__ __forceinline__ unsigned tn16_sign_bits(const u32x4& lo, const u32x4& hi) {
    unsigned am = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned w = q < 4 ? lo[q] : hi[q - 4];
        am = ((w + 0x7FFF7FFFu) & 0x80008000u) | (am >> 2);        // dword q ends at bits 2q+1 (low half) and 17+2q (high half)
    }
    return ((am >> 1) & 0x5555u) | ((am >> 16) & 0xAAAAu);
}

// One layer for the wave's 32-sample tile.
//   KIND 0: first layer (input k-steps only)   1: hidden   2: skip layer (hidden + input k-steps)   3: heads
// The epilogue of an n-tile (bias add in fp32, round to bf16, ReLU on the packed pair) follows its last MFMA; the MFMA
// pipe is kept busy meanwhile by the SIMD's other wave.
// vb: per-lane LDS byte offset of this layer's biases (+ 16 h).  KIND 3 leaves the raw head accumulator in `acc`.
// TRAIN: also stash the layer's output tiles (transposed, feature tiles ft0 + t) and its ReLU sign bits (layer l).
template <int HID, int KIND, bool TRAIN = false>
__device__ __forceinline__ void tn16_layer(Pipe16& p, const unsigned char* lds, uint32_t vb,
                                           const bf16x8 (&bin)[HID / 16], const bf16x8 (&enc)[TN16_KE],
                                           bf16x8 (&bout)[HID / 16], f32x16& acc,
                                           const Stash16& st = Stash16{}, uint32_t sel_off = 0, int ft0 = 0, int l = 0) {
    constexpr int NT = KIND == 3 ? 1 : HID / 32, KH = HID / 16;
    constexpr int KPT = KIND == 0 ? TN16_KE : (KIND == 1 ? KH : (KIND == 2 ? KH + TN16_KE : TN16_STAGE));
    constexpr int KUSE = KIND == 3 ? KH : KPT;                 // k-steps with MFMAs (the head stage is zero-padded)
    static_assert((NT * KPT) % TN16_STAGE == 0, "a layer must be a whole number of stages");
    uint32_t mb[HID / 64];
    tn_static_for<NT>([&](auto tc) TN_INLINE_LAMBDA {
        constexpr int t = decltype(tc)::value;
        tn_static_for<KPT>([&](auto sc) TN_INLINE_LAMBDA {
            constexpr int s = decltype(sc)::value;
            constexpr int F = t * KPT + s;
            if constexpr (F % TN16_STAGE == 0) tn16_boundary<TRAIN>(p);
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
            u32x4 w0, w1;
            {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(lds + vb + (32 * t) * 4);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(lds + vb + (32 * t + 8) * 4);
                w0[0] = tn16_relu2(tn16_cvt2(acc[0] + b0[0], acc[1] + b0[1]));
                w0[1] = tn16_relu2(tn16_cvt2(acc[2] + b0[2], acc[3] + b0[3]));
                w0[2] = tn16_relu2(tn16_cvt2(acc[4] + b1[0], acc[5] + b1[1]));
                w0[3] = tn16_relu2(tn16_cvt2(acc[6] + b1[2], acc[7] + b1[3]));
                bout[2 * t] = __builtin_bit_cast(bf16x8, w0);
            }
            {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(lds + vb + (32 * t + 16) * 4);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(lds + vb + (32 * t + 24) * 4);
                w1[0] = tn16_relu2(tn16_cvt2(acc[8] + b0[0], acc[9] + b0[1]));
                w1[1] = tn16_relu2(tn16_cvt2(acc[10] + b0[2], acc[11] + b0[3]));
                w1[2] = tn16_relu2(tn16_cvt2(acc[12] + b1[0], acc[13] + b1[1]));
                w1[3] = tn16_relu2(tn16_cvt2(acc[14] + b1[2], acc[15] + b1[3]));
                bout[2 * t + 1] = __builtin_bit_cast(bf16x8, w1);
            }
            if constexpr (TRAIN) {
                const unsigned m16 = tn16_sign_bits(w0, w1);
                if constexpr ((t & 1) == 0) mb[t / 2] = m16; else mb[t / 2] |= m16 << 16;
                tn16_stash_tile(lds, sel_off, p.lane16, st, ft0 + t, bout[2 * t], bout[2 * t + 1], acc);
            }
        }
    });
    if constexpr (TRAIN && KIND != 3) {
        typedef unsigned mvec __attribute__((ext_vector_type(HID / 64)));
        mvec mv;
#pragma unroll
        for (int i = 0; i < HID / 64; ++i) mv[i] = mb[i];
        *reinterpret_cast<mvec*>(st.mask + (int64_t)l * st.mask_lstride + p.lane16 / 16 * (HID / 16)) = mv;
    }
}

// PositionalEncoding(L, include_input=True) of one point as the bf16 B operand of the input k-steps (slot map:
// tnerf_internal.h).  reference src/encoding.py:27-33.  The result is rounded to bf16 (2^-9 relative), so sin/cos come
// from the hardware v_sin_f32 (argument in revolutions, |error| ~ 1e-6): y = 2^k p is exact, y/(2 pi) is formed in two
// terms (c_hi, c_lo, residual by fma) so that the fractional turn is good to ~1e-7 even at 2^9 * 6 / 2 pi = 490 turns,
// and cos(x) = sin(x + 1/4 turn).
__device__ __forceinline__ float tn16_sin_turns(float y, float quarter) {
    const float c_hi = 0.15915494f, c_lo = 6.4206382e-09f;              // 1/(2 pi) = c_hi + c_lo, c_hi = fp32(1/(2 pi))
    const float r_hi = __fmul_rn(y, c_hi);
    const float err = fmaf(y, c_hi, -r_hi);                              // exact residual of the product
    const float lo = fmaf(y, c_lo, err);
    const float fr = __builtin_amdgcn_fractf(r_hi);
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf((fr + quarter) + lo));
}

__device__ __forceinline__ void tn16_encode(float px, float py, float pz, int Lf, int h, bf16x8 (&enc)[TN16_KE]) {
    asm volatile("" : "+s"(Lf));       // opaque: otherwise every `slot < 3 Lf` below is hoisted out of the tile loop as a 64-bit lane mask (tx_encode, mlpx3_core.hpp)
    const float quarter = h ? 0.25f : 0.0f;
    tn_static_for<TN16_KE>([&](auto uc) TN_INLINE_LAMBDA {
        constexpr int u = decltype(uc)::value;
        float v[8];
        tn_static_for<8>([&](auto ec) TN_INLINE_LAMBDA {
            constexpr int e = decltype(ec)::value;
            constexpr int a = 8 * u + e, k = a / 3, c = a % 3;
            const float pc = c == 0 ? px : (c == 1 ? py : pz);
            float r = 0.0f;
            if (a < 3 * Lf) {
                r = tn16_sin_turns(pc * (float)(1u << (k < 31 ? k : 0)), quarter);
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

// Workgroup prologue shared by the bf16 kernels: biases (and the transposition selectors) -> LDS, the first four stages
// of the stream starting at `src` in flight, stage 0 published, the first fragments in registers.
__device__ __forceinline__ void tn16_prologue(Pipe16& p, unsigned char* lds, const unsigned char* packed, const Net16& n,
                                              const unsigned char* src, int n_stage, int lane, int wave, bool selectors) {
    {
        float* bl = reinterpret_cast<float*>(lds + TN16_RING);
        const float* bg = reinterpret_cast<const float*>(packed + n.bias_off);
        for (int i = threadIdx.x; i < n.n_bias; i += 512) bl[i] = bg[i];
        if (selectors && wave == 0) tn16_write_selectors(lds, TN16_SEL_OFF(n.n_bias), lane);
    }
    p.lane16 = lane * 16;
    p.src = src; p.src_off = 0; p.stream_bytes = (uint32_t)n_stage * TN16_SLOT;
    p.dst_off = 0;
    p.lds_dst0 = (uint32_t)(uintptr_t)lds + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) p.voff[i] = lane * 16 + wave * 2048 + i * 1024;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i <= TN16_LEAD; ++i) tn16_issue_stage(p);
    TN16_WAIT_VM(2 * TN16_LEAD);
    __builtin_amdgcn_s_barrier();
    p.va_cur = p.lane16; p.va_nxt = p.lane16; p.nxt_off = 0;                  // the first boundary moves va_cur onto slot 0
#pragma unroll
    for (int i = 0; i < TN16_PF; ++i) p.afr[i] = *reinterpret_cast<const bf16x8*>(lds + p.lane16 + i * 1024);
}

