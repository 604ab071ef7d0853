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
//   * k-step-major order in two HALF-PASSES per layer (output tiles 0..NT/2-1 = half A, then half B), all of a half's
//     accumulators live; the three pieces of the input activation X[s] are dead once half B has passed k-step s, so the layer's
//     output pieces are written back into the same registers: one activation array (3 x HID/16 x 4 registers) instead of an
//     in / out pair.  8x256: 192 + 128 registers — one wave per SIMD (512-register budget), four waves per workgroup.
//   * with one wave per SIMD nothing else fills the matrix pipe while a wave runs an epilogue (bias, ReLU, sign bits, stash
//     store, split: ~8 VALU per value), so the epilogue of one half is cut into per-pair MICRO-STEPS that are issued in the
//     shadows of the other half's MFMAs (tx_pass's hook): half A's epilogue rides on the second half of pass B (X[0..KH/2)
//     is dead there), half B's on the first half of the next layer's pass A.  The order is pinned with sched_barrier.
//   * the weight stream carries three pieces per fragment (tnerf_internal.h, NetX3): per (half, k-step) record NT/2 x 3 KB
//     through an LDS ring of 24 KB stages (LDS-DMA, counted vmcnt, one raw barrier per stage), shared by the four waves.
#pragma once
#include "mlp16_core.hpp"

#define TX_SLOT (TX_STAGE * 1024)              // bytes per stage
#define TX_NS 5                                // ring slots
#define TX_RING (TX_NS * TX_SLOT)              // 120 KB
#define TX_LEAD 3                              // stages in flight behind the published one (LEAD + 2 <= NS)
// Waves per workgroup: 256-wide nets need the 512-register budget of one wave per SIMD; a 128-wide wave fits 256 registers, so
// two of them share a SIMD and fill each other's gaps (encoder, compositing, barriers).
// Diagnostic builds only (-DTX_SPLIT_ACC=1, tools/x3_accuracy_probe.py): the five correction products of a k-step accumulate in a
// second accumulator set (acc[NH + tile]) that never holds the large a1.b1 sums, and the two are added in the epilogue — what the
// chain would deliver if the matrix pipe did not drop addends below 1/8 ulp of its accumulator (accuracy).  Costs
// 128 registers the 256-wide kernels do not have (it spills): a measurement, not a product option.
#ifndef TX_SPLIT_ACC
#define TX_SPLIT_ACC 0
#endif
#define TX_ACCN(HID) ((HID) / 64 * (1 + TX_SPLIT_ACC))
#ifndef TX_NW128
#define TX_NW128 8
#endif
// (The 128-wide TRAINING forward keeps four: its stash pointers and sign words do not fit 256 registers — 91 spilled.)
template <int HID, bool TRAIN_FWD = false> struct TxCfg {
    static constexpr int NW = (HID == 128 && !TRAIN_FWD) ? TX_NW128 : 4;
    static constexpr int DPW = TX_STAGE / NW;                     // DMA pieces per wave and stage
    static_assert(TX_STAGE % NW == 0 && DPW <= 8, "ring budget");
};
static_assert(TX_LEAD + 2 <= TX_NS, "ring budget");

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
    uint32_t lds_dst0;           // absolute LDS address of ring + wave * DPW KB
    uint32_t voff;               // lane * 16 + wave * DPW * 1024: this wave's first piece of a stage
    const unsigned char* pend_src; uint32_t pend_dst;      // the stage whose pieces are being issued behind MFMAs (tx_defer_stage)
};

// This wave's DPW pieces of a stage are 1 KB each, consecutive in the stream and in the slot: piece i is the pending
// stage's base (+ 4 KB for i >= 4) with the instruction's immediate offset (i & 3) KB — the offset applies to the global AND
// the LDS address.  M0 (the DMA's LDS base) is compiler-reserved: saved and restored around the load.
template <int I>
__device__ __forceinline__ void tx_issue_piece(const unsigned char* src, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(src + (I >> 2) * 4096), "s"(lds_dst + (I >> 2) * 4096), "n"((I & 3) * 1024) : "memory");
}
// the stage `dst_off` / `src_off` point at: all pieces now (prologue, heads^T) ...
template <int DPW>
__device__ __forceinline__ void tx_issue_stage(PipeX& p) {
    const unsigned char* s = p.src + p.src_off;
    tn_static_for<DPW>([&](auto ic) TN_INLINE_LAMBDA { tx_issue_piece<decltype(ic)::value>(s, p.voff, p.lds_dst0 + p.dst_off); });
    p.src_off += TX_SLOT; if (p.src_off == p.stream_bytes) p.src_off = 0;
    p.dst_off += TX_SLOT; if (p.dst_off == TX_RING) p.dst_off = 0;
}
// ... or piece by piece behind the MFMAs of the stage that has just been published (tx_pass): back-to-back DMA instructions
// cost the wave more issue time than the same six spread over as many MFMA groups.
__device__ __forceinline__ void tx_defer_stage(PipeX& p) {
    p.pend_src = p.src + p.src_off; p.pend_dst = p.lds_dst0 + p.dst_off;
    p.src_off += TX_SLOT; if (p.src_off == p.stream_bytes) p.src_off = 0;
    p.dst_off += TX_SLOT; if (p.dst_off == TX_RING) p.dst_off = 0;
}

// Start of a stage: wait for this wave's DMA of the stage (LEAD-1 younger ones may stay in flight; the training kernels' stash
// stores, which are younger still, make this wait stricter than it needs to be, never laxer: vmcnt retires in issue order),
// barrier (the stage is readable by everyone, the slot of the previous one is free), issue stage + LEAD (DEFER: the caller
// issues its pieces with tx_issue_piece before the next boundary).
template <int DPW, bool DEFER>
__device__ __forceinline__ void tx_boundary(PipeX& p) {
    TN16_WAIT_VM(DPW * (TX_LEAD - 1));
    __builtin_amdgcn_s_barrier();
    if constexpr (DEFER) tx_defer_stage(p); else tx_issue_stage<DPW>(p);
    p.cur += TX_SLOT; if (p.cur == TX_RING) p.cur = 0;
}

// Workgroup prologue: biases -> LDS, LEAD stages in flight, the first one landed; the first tx_boundary publishes stage 0.
// `src` / `n_stage`: the stream this kernel walks (forward: packed, n.n_stage; dgrad: the backward stream behind it).
template <int NW>
__device__ __forceinline__ void tx_prologue(PipeX& p, unsigned char* lds, const unsigned char* packed, const NetX3& n,
                                            const unsigned char* src, int n_stage, int lane, int wave) {
    constexpr int DPW = TX_STAGE / NW;
    {
        float* bl = reinterpret_cast<float*>(lds + TX_RING);
        const float* bg = reinterpret_cast<const float*>(packed + n.bias_off);
        for (int i = threadIdx.x; i < n.n_bias; i += NW * 64) bl[i] = bg[i];
    }
    p.lane16 = lane * 16;
    p.src = src; p.src_off = 0; p.stream_bytes = (uint32_t)n_stage * TX_SLOT;
    p.dst_off = 0;
    p.lds_dst0 = (uint32_t)(uintptr_t)lds + wave * (DPW * 1024);
    p.voff = lane * 16 + wave * DPW * 1024;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < TX_LEAD; ++i) tx_issue_stage<DPW>(p);
    p.cur = TX_RING - TX_SLOT;                   // the first boundary moves it onto slot 0
}

// The activation of a wave's 32-sample tile: three bf16 pieces of every k-step operand.
template <int HID>
struct ActX { u32x4 p1[HID / 16], p2[HID / 16], p3[HID / 16]; };          // packed bf16 pairs; dword q = values 2q, 2q+1 of the k-step
#define TX_BF(x) __builtin_bit_cast(bf16x8, (x))
struct EncX { bf16x8 p1[TN16_KE], p2[TN16_KE], p3[TN16_KE]; };

typedef float f32x2 __attribute__((ext_vector_type(2)));
struct FragX { bf16x8 a1, a2, a3; };
__device__ __forceinline__ FragX tx_frag_load(const unsigned char* base, int tl) {
    FragX f;
    f.a1 = *reinterpret_cast<const bf16x8*>(base + (tl * 3 + 0) * 1024);
    f.a2 = *reinterpret_cast<const bf16x8*>(base + (tl * 3 + 1) * 1024);
    f.a3 = *reinterpret_cast<const bf16x8*>(base + (tl * 3 + 2) * 1024);
    return f;
}
#define TX_PIN() __builtin_amdgcn_sched_barrier(0)

// One half-pass of a layer: NK k-step records, each NTU tiles x 6 MFMAs into acc[tile slot].
// KIND 0: the TN16_KE input k-steps (B operand from E)   1: the hidden k-steps   3: heads — ONE output tile (acc[0]), so a
// record's NH tile slots carry NH consecutive k-steps of it instead (KH / NH records).
// A skip layer's half is a KIND 1 pass followed by a KIND 0 pass that accumulates (ZERO = false) — one copy of the long pass
// in the instruction cache instead of two.
// RPS = records per stage (2 for 256-wide, 4 for 128-wide nets); every pass is a whole number of stages, so the stage
// phase of record k is k % RPS.  ZERO: the accumulators start at zero.
// hook(integral_constant<slot>) is called behind MFMA number slot = (k * NTU + tile) * 6 + j: work to issue in its shadow.
// The next group's A fragments are read from LDS behind the first MFMA of a group (not across a stage boundary).
template <int HID, int KIND, bool ZERO, int NW, typename Hook>
__device__ __forceinline__ void tx_pass(PipeX& p, const unsigned char* lds, const ActX<HID>& X, const EncX& E,
                                        f32x16 (&acc)[TX_ACCN(HID)], Hook&& hook) {
    constexpr int NH = HID / 64, KH = HID / 16, RPS = TX_STAGE / (NH * 3), DPW = TX_STAGE / NW;
    constexpr int NK = KIND == 0 ? TN16_KE : (KIND == 3 ? KH / NH : KH);
    constexpr int NTU = NH;
    static_assert(NK % RPS == 0, "a half-pass must be a whole number of stages");
    FragX cur;
    tn_static_for<NK>([&](auto kc) TN_INLINE_LAMBDA {
        constexpr int k = decltype(kc)::value;
        // The stage boundary (wait, barrier, next DMA stage named) of every stage but the pass's first is taken one MFMA group
        // EARLY — in front of the last group of the stage before, whose A fragments are already in registers — so that the new
        // stage's first fragments are read behind that group's MFMAs instead of in front of an idle matrix pipe.  (The ring has
        // the spare slot this needs: TX_LEAD + 2 <= TX_NS.)
        if constexpr (k == 0) tx_boundary<DPW, true>(p);
        const unsigned char* base = lds + p.cur + (k % RPS) * (NH * 3 * 1024) + p.lane16;
        if constexpr (k == 0) cur = tx_frag_load(base, 0);
        tn_static_for<NTU>([&](auto tc) TN_INLINE_LAMBDA {
            constexpr int tl = decltype(tc)::value;
            constexpr int ks = KIND == 3 ? k * NH + tl : k;          // the k-step of this group's B operand
            constexpr int ta = KIND == 3 ? 0 : tl;                   // ... and its accumulator
            bf16x8 b1, b2, b3;
            if constexpr (KIND == 0) { b1 = E.p1[ks]; b2 = E.p2[ks]; b3 = E.p3[ks]; }
            else { b1 = TX_BF(X.p1[ks]); b2 = TX_BF(X.p2[ks]); b3 = TX_BF(X.p3[ks]); }
            constexpr bool more_tile = tl + 1 < NTU;
            constexpr bool more_rec = !more_tile && (k + 1) % RPS != 0 && k + 1 < NK;
            constexpr bool early = !more_tile && (k + 1) % RPS == 0 && k + 1 < NK;      // last group of a stage, another follows in this pass
            constexpr int s0 = (k * NTU + tl) * 6;
            FragX nxt;
            if constexpr (early) tx_boundary<DPW, true>(p);
            constexpr int tc_ = TX_SPLIT_ACC ? ta + NH : ta;         // where the correction products go
            if constexpr (ZERO && ks == 0) { const f32x16 z = {}; acc[tc_] = TN16_MFMA(cur.a3, b1, z); }
            else                          acc[tc_] = TN16_MFMA(cur.a3, b1, acc[tc_]);
            if constexpr (more_tile)     nxt = tx_frag_load(base, tl + 1);
            else if constexpr (more_rec) nxt = tx_frag_load(base + NH * 3 * 1024, 0);
            else if constexpr (early)    nxt = tx_frag_load(lds + p.cur + p.lane16, 0);
            hook(std::integral_constant<int, s0>{});     TX_PIN();
            acc[tc_] = TN16_MFMA(cur.a2, b2, acc[tc_]); hook(std::integral_constant<int, s0 + 1>{}); TX_PIN();
            acc[tc_] = TN16_MFMA(cur.a1, b3, acc[tc_]); hook(std::integral_constant<int, s0 + 2>{}); TX_PIN();
            acc[tc_] = TN16_MFMA(cur.a2, b1, acc[tc_]); hook(std::integral_constant<int, s0 + 3>{});
            {   // this group's share of the pending stage's DMA pieces
                constexpr int GPS = RPS * NTU, PPS = (DPW + GPS - 2) / (GPS - 1), g = (k % RPS) * NTU + tl;
                tn_static_for<PPS>([&](auto uc) TN_INLINE_LAMBDA {
                    constexpr int i = g * PPS + decltype(uc)::value;
                    if constexpr (i < DPW && g < GPS - 1) tx_issue_piece<i>(p.pend_src, p.voff, p.pend_dst);
                });
            }
            TX_PIN();
            acc[tc_] = TN16_MFMA(cur.a1, b2, acc[tc_]); hook(std::integral_constant<int, s0 + 4>{}); TX_PIN();
            if constexpr (TX_SPLIT_ACC && ZERO && ks == 0) { const f32x16 z = {}; acc[ta] = TN16_MFMA(cur.a1, b1, z); }
            else                                            acc[ta] = TN16_MFMA(cur.a1, b1, acc[ta]);
            hook(std::integral_constant<int, s0 + 5>{}); TX_PIN();
            if constexpr (more_tile || more_rec || early) cur = nxt;
        });
    });
}

// heads^T of the backward stream: ONE k-step (B operand = Z.p*[0]) into both halves' accumulators; the stage holds record A,
// record B and padding.
template <int HID, int NW>
__device__ __forceinline__ void tx_pass_headsT(PipeX& p, const unsigned char* lds, const EncX& Z, f32x16 (&accA)[TX_ACCN(HID)], f32x16 (&accB)[TX_ACCN(HID)]) {
    constexpr int NH = HID / 64;
    tx_boundary<TX_STAGE / NW, false>(p);
    const unsigned char* base = lds + p.cur + p.lane16;
    const f32x16 z = {};
    tn_static_for<2 * NH>([&](auto tc) TN_INLINE_LAMBDA {
        constexpr int t = decltype(tc)::value;
        const FragX f = tx_frag_load(base, t);                                    // record B follows record A: slot t = half * NH + tl
        f32x16 a = TN16_MFMA(f.a3, Z.p1[0], z);
        a = TN16_MFMA(f.a2, Z.p2[0], a); a = TN16_MFMA(f.a1, Z.p3[0], a);
        a = TN16_MFMA(f.a2, Z.p1[0], a); a = TN16_MFMA(f.a1, Z.p2[0], a); a = TN16_MFMA(f.a1, Z.p1[0], a);
        if constexpr (t < NH) accA[t] = a; else accB[t - NH] = a;
        if constexpr (TX_SPLIT_ACC) { if constexpr (t < NH) accA[t + NH] = z; else accB[t] = z; }
    });
}

// ---- epilogues as micro-steps.  A half has NP = NH*8 register PAIRS (values 2pr, 2pr+1 of a tile); pair i of the order
// below is tile 2(i/16) + 1 - (i%16)/8 of the half, pair 7 - i%8: the sign words of the training stash are built by shifting
// (alignbit), i.e. bit 31 first = the odd tile of the word, register 15 downwards.
template <int HID, int HALF, int I> struct TxPair {
    static constexpr int NH = HID / 64;
    static constexpr int tl = 2 * (I / 16) + (1 - (I % 16) / 8);      // accumulator slot in the half
    static constexpr int t = HALF * NH + tl;                          // n-tile
    static constexpr int pr = 7 - I % 8;
    static constexpr int r0 = 2 * pr, r1 = 2 * pr + 1;
    static constexpr int xs = 2 * t + pr / 4, xq = pr % 4;            // activation k-step and dword the pair lands in
    static constexpr int row0 = 32 * t + (r0 & 3) + 8 * (r0 >> 2), row1 = 32 * t + (r1 & 3) + 8 * (r1 >> 2);   // feature rows (+ 4h) of the stash
};
struct TxEpi { float v0[4], v1[4], r0[4], r1[4], s0[4], s1[4]; f32x2 b[4]; uint32_t msk; };      // up to 4 pairs in flight

// steps 3..5 of any epilogue: the exact split of the pair into the activation registers
template <int HID, int HALF, int I, int J>
__device__ __forceinline__ void tx_epi_split(ActX<HID>& X, TxEpi& e) {
    using P = TxPair<HID, HALF, I>;
    constexpr int u = I % 4;
    if constexpr (J == 3) {
        e.r0[u] = e.v0[u] - __uint_as_float(__float_as_uint(e.v0[u]) & 0xFFFF0000u);
        e.r1[u] = e.v1[u] - __uint_as_float(__float_as_uint(e.v1[u]) & 0xFFFF0000u);
    } else if constexpr (J == 4) {
        e.s0[u] = e.r0[u] - __uint_as_float(__float_as_uint(e.r0[u]) & 0xFFFF0000u);
        e.s1[u] = e.r1[u] - __uint_as_float(__float_as_uint(e.r1[u]) & 0xFFFF0000u);
    } else if constexpr (J == 5) {
        X.p1[P::xs][P::xq] = __builtin_amdgcn_perm(__float_as_uint(e.v1[u]), __float_as_uint(e.v0[u]), 0x07060302u);
        X.p2[P::xs][P::xq] = __builtin_amdgcn_perm(__float_as_uint(e.r1[u]), __float_as_uint(e.r0[u]), 0x07060302u);
        X.p3[P::xs][P::xq] = __builtin_amdgcn_perm(__float_as_uint(e.s1[u]), __float_as_uint(e.s0[u]), 0x07060302u);
    }
}

// Forward: bias (fp32, LDS byte offset vb + 16 h: this layer's biases for rows 4h..), ReLU, [training: fp32 value to the stash,
// sign bit], split.  The bias pair is READ one micro-step before it is used (step 0 of the pair; step 1 adds), so that the
// compiler's lgkmcnt wait lands a whole MFMA later.
// srow: per-lane stash pointer of the layer's activation rows (row 4h, this sample); mword: per-lane pointer of its sign words.
template <int HID, int HALF, int I, int J, bool TRAIN, int PPG>
__device__ __forceinline__ void tx_epi_fwd(const f32x16 (&acc)[TX_ACCN(HID)], ActX<HID>& X, TxEpi& e, const unsigned char* lds, uint32_t vb,
                                           float* __restrict__ srow, uint32_t* __restrict__ mword) {
    using P = TxPair<HID, HALF, I>;
    constexpr int u = I % 4;
    // PPG: pairs per MFMA group of the window this runs in; pair I + PPG is the one that takes this register slot next
    if constexpr (J == 0) {
        if constexpr (I < PPG) e.b[u] = *reinterpret_cast<const f32x2*>(lds + vb + P::row0 * 4);      // the window's first group
        e.v0[u] = acc[P::tl][P::r0]; e.v1[u] = acc[P::tl][P::r1];
        if constexpr (TX_SPLIT_ACC) { e.v0[u] += acc[P::tl + P::NH][P::r0]; e.v1[u] += acc[P::tl + P::NH][P::r1]; }
    } else if constexpr (J == 1) {
        e.v0[u] = fmaxf(e.v0[u] + e.b[u][0], 0.0f); e.v1[u] = fmaxf(e.v1[u] + e.b[u][1], 0.0f);
    } else if constexpr (J == 5 && I + PPG < P::NH * 8) {          // (also runs the split's step 5 below)
        e.b[(I + PPG) % 4] = *reinterpret_cast<const f32x2*>(lds + vb + TxPair<HID, HALF, I + PPG>::row0 * 4);
        tx_epi_split<HID, HALF, I, J>(X, e);
    } else if constexpr (J == 2) {
        if constexpr (TRAIN) {
            TN_STASH_STORE(&srow[P::row0 * 32], e.v0[u]); TN_STASH_STORE(&srow[P::row1 * 32], e.v1[u]);
            if constexpr (I % 16 == 0) e.msk = 0u;
            e.msk = __builtin_amdgcn_alignbit(e.msk, __float_as_uint(e.v1[u]) + 0x7FFFFFFFu, 31);
            e.msk = __builtin_amdgcn_alignbit(e.msk, __float_as_uint(e.v0[u]) + 0x7FFFFFFFu, 31);
            if constexpr (I % 16 == 15) mword[P::t / 2] = e.msk;
        }
    } else tx_epi_split<HID, HALF, I, J>(X, e);
}

// Backward: ReLU backward with the forward's sign bits (mw: the words of the layer this activation gradient belongs to),
// dZ to the stash, split.
template <int HID, int HALF, int I, int J>
__device__ __forceinline__ void tx_epi_bwd(const f32x16 (&acc)[TX_ACCN(HID)], ActX<HID>& X, TxEpi& e, const uint32_t (&mw)[HID / 64], float* __restrict__ zrow) {
    using P = TxPair<HID, HALF, I>;
    constexpr int u = I % 4;
    if constexpr (J == 0) {
        e.v0[u] = acc[P::tl][P::r0]; e.v1[u] = acc[P::tl][P::r1];
        if constexpr (TX_SPLIT_ACC) { e.v0[u] += acc[P::tl + P::NH][P::r0]; e.v1[u] += acc[P::tl + P::NH][P::r1]; }
    } else if constexpr (J == 1) {
        e.v0[u] = __int_as_float(__float_as_int(e.v0[u]) & __builtin_amdgcn_sbfe((int)mw[P::t / 2], (P::t & 1) * 16 + P::r0, 1));
        e.v1[u] = __int_as_float(__float_as_int(e.v1[u]) & __builtin_amdgcn_sbfe((int)mw[P::t / 2], (P::t & 1) * 16 + P::r1, 1));
    } else if constexpr (J == 2) {
        TN_STASH_STORE(&zrow[P::row0 * 32], e.v0[u]); TN_STASH_STORE(&zrow[P::row1 * 32], e.v1[u]);
    } else tx_epi_split<HID, HALF, I, J>(X, e);
}

// The slots [W0, W0 + 6 G) of a pass as an epilogue window: group g = (slot - W0) / 6 carries micro-step (slot - W0) % 6 of the
// pairs g*PPG .. g*PPG + PPG - 1 (PPG = ceil(NP / G) <= 4).  f(integral_constant<I>, integral_constant<J>, integral_constant<PPG>).
template <int W0, int G, int NP, typename F>
__device__ __forceinline__ auto tx_window(F&& f) {
    return [&f](auto sc) TN_INLINE_LAMBDA {
        constexpr int s = decltype(sc)::value;
        constexpr int PPG = (NP + G - 1) / G;
        static_assert(PPG <= 4, "epilogue window too short");
        if constexpr (s >= W0 && s < W0 + 6 * G) {
            constexpr int g = (s - W0) / 6, j = (s - W0) % 6;
            tn_static_for<PPG>([&](auto uc) TN_INLINE_LAMBDA {
                constexpr int i = g * PPG + decltype(uc)::value;
                if constexpr (i < NP) f(std::integral_constant<int, i>{}, std::integral_constant<int, j>{}, std::integral_constant<int, PPG>{});
            });
        }
    };
}
// A whole epilogue with nothing to hide behind.  NJ = 6: all micro-steps; 3: without the split (nothing consumes the pieces).
template <int NP, int NJ, typename F>
__device__ __forceinline__ void tx_drain(F&& f) {
    tn_static_for<NP>([&](auto ic) TN_INLINE_LAMBDA {
        tn_static_for<NJ>([&](auto jc) TN_INLINE_LAMBDA { f(ic, jc, std::integral_constant<int, 1>{}); });
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
