// "x3" chain: the fp32 MLP chain on the fp16 matrix pipe with every fp32 product formed from THREE partial products
// (the arithmetic of the matrix pipe it is built around: tools/microbench/mfma_sim.c).
//
// An operand x (fp32) is scaled by a power of two into the fp16 range and carried as two fp16 pieces, x 2^s = p1 + p2, both
// rounded to nearest: 22-23 significant bits.  a b 2^(s+t) = a1 b1 + (a1 b2 + a2 b1) [+ a2 b2 <= 2^-22 |ab|: dropped].  Three
// v_mfma_f32_32x32x16_f16 per (32-feature tile, 16-wide k-step): the LEADING products accumulate in one fp32 accumulator, the
// two CORRECTION products in a second one, and the two are added once, in the epilogue.  The split accumulators are not a
// luxury: the matrix pipe cuts its accumulator to the window of the group's largest product by FLOOR (toward -inf) before
// adding, so an accumulator that carries fine correction bits and then meets large leading products is biased downwards,
// identically for every sample — measured as 5x the reference's error on single weight-gradient tensors.  An accumulator that
// only ever receives leading products has no bits below that window; the correction accumulator's cuts are 2^-11 smaller.
// Measured (model of the pipe fitted to hardware dumps, 8x256 fixture): activations 1.1e-7 from fp64 after 8 layers (an fp32 fma
// chain: 2.2e-7), every weight-gradient tensor within 1.3x of the reference's own CPU fp32 error.
//
// Scales.  Weights: one power of two per layer, max|W| 2^s in (2^11, 2^12] (k_x3stats / the finishing kernel keep it, the
// stream's TX_META records carry it).  Activations: one power of two PER SAMPLE and layer, chosen from a bound known before the
// layer's epilogue starts:  max_i |H_l[i]| <= max|W_l| ||X_l||_1 + max|b_l|  with X_l the layer's input, whose L1 norm the
// previous epilogue summed;  t = 14 - exponent(bound)  puts every scaled activation below 2^14 (fp16 overflows at 2^16).  Values
// more than 2^-17 below the bound keep less than 22 bits, but never less than 2^-39 of the bound absolutely.
//
// Orientation as everywhere in this library: weights = A operand, the wave's 32 samples on the lanes, the 32x32 fp32
// accumulator of n-tile t = the next layer's B operand for k-steps 2t, 2t+1 — after bias / ReLU in fp32 and one split.
//   * k-step-major order in two HALF-PASSES per layer (output tiles 0..NT/2-1 = half A, then half B), all of a half's
//     accumulators live; the pieces of the input activation X[s] are dead once half B has passed k-step s, so the layer's
//     output pieces are written back into the same registers.  8x256: 128 (pieces) + 256 (two halves x main / correction)
//     registers — one wave per SIMD, four waves per workgroup; 4x128: 64 + 128 — two waves per SIMD, eight per workgroup (TxCfg).
//   * the epilogue of one half (descale, bias, ReLU, sign bits, stash store, L1 norm, split) is cut into per-pair MICRO-STEPS
//     issued in the shadows of the other half's MFMAs (tx_pass's hook); the order is pinned with sched_barrier.
//   * the weight stream: per (half, k-step) record NT/2 x 2 KB through an LDS ring of 16 KB stages (LDS-DMA, counted vmcnt,
//     one raw barrier per two stages), shared by the workgroup's waves.
#pragma once
#include "mlp16_core.hpp"

// Ablation knobs for timing (tools/x3_stamp_probe.py): WRONG numerics, diagnostic builds only (tools/build_variant.sh passes -DTN_DIAG).
#if (defined(TX_NO_EPI) || defined(TX_NO_FRAG) || defined(TX_NO_DMA) || defined(TX_NO_STASH) || defined(TX_NO_SIGN) || defined(TX_SERIAL_EPI) || defined(TX_NO_BARRIER)) && !defined(TN_DIAG)
#error "TX_NO_EPI / TX_NO_FRAG / TX_NO_DMA are diagnostic knobs with wrong numerics: build with -DTN_DIAG (tools/build_variant.sh)"
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define TX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

#define TX_SLOT (TX_STAGE * 1024)              // bytes per stage
// The ring by the workgroup's wave count.  4 waves (one per SIMD, 512 registers each: the 256-wide kernels): 7 slots = 112 KB, 5 stages
// in flight behind the published one.  8 waves (two per SIMD, 256 registers each: the 128-wide fused kernels, TxCfg): the eight
// waves' input pieces take 64 KB of LDS, so 5 slots = 80 KB and 3 stages in flight — the two waves of a SIMD share its matrix pipe,
// a stage lasts twice as long on the wall clock and a DMA still has ~3 k cycles to land.  NS = LEAD + 2 always (tx_boundary).
#define TX_NS_OF(nw) ((nw) == 8 ? 5 : 7)                        // ring slots
#define TX_LEAD_OF(nw) (TX_NS_OF(nw) - 2)                       // stages in flight behind the published one
#define TX_RING_OF(nw) (TX_NS_OF(nw) * TX_SLOT)                 // 112 KB / 80 KB
#ifndef TX_WAIT_EXTRA
#define TX_WAIT_EXTRA 0                            // TN_DIAG timing experiments only: a non-zero value makes the stage wait too lax (wrong results)
#elif !defined(TN_DIAG)
#error "TX_WAIT_EXTRA is a diagnostic knob: build with -DTN_DIAG"
#endif
#define TX_TOP 14                              // scaled activations stay below 2^TX_TOP
#define TX_BND_N (TN_MAXD + 1)                 // row groups whose magnitude bounds a training kernel keeps (see TX_BND_OFF)
// Waves per workgroup, by kernel.  256-wide: one wave per SIMD everywhere (pieces 128 + split accumulators 256 registers).  128-wide:
// pieces 64 + accumulators 128 registers fit a 256-register wave, so the kernels that carry little besides the tile run TWO waves per
// SIMD.  Why: one wave gets ~3 non-MFMA instructions per MFMA for free and pays ~4 cycles for each further one — a 128-wide epilogue
// carries 8 to 9 per MFMA; a second wave's MFMAs fill exactly those gaps (tools/microbench/two_waves.hip: the training mixture at 6
// per MFMA runs at 102 cycles per 3 MFMAs with two waves, 136 with one; measured on the kernels: DESIGN.md).
//   TILE    k_tilex3_fwd / k_tilex3_bwd / k_mlpx3_bwd                 8 at 128-wide
//   RENDER  k_renderx3<.., false> (inference)                         8 at 128-wide
//   RAY     k_renderx3<.., true> / k_dgradx3 (a ray's compositing state rides through the walk: 10-15 values too many for 256
//           registers) and k_mlpx3_fwd (its in-memory input rows)     4: the launchers send 128-wide TRAINING through the tile kernels
template <int HID> struct TxCfg {
#ifdef TX_NW128                                                   // diagnostic builds: the 128-wide kernels at 4 waves again (A/B timing)
    static constexpr int NW2 = HID == 128 ? TX_NW128 : 4;
#else
    static constexpr int NW2 = HID == 128 ? 8 : 4;
#endif
    static constexpr int TILE = NW2, RENDER = NW2, RAY = 4;
    static_assert(TX_STAGE % NW2 == 0 && (NW2 == 4 || NW2 == 8), "ring budget");
};
static_assert(TX_LEAD_OF(8) >= 3 && TX_LEAD_OF(4) >= 3, "ring budget");

// ---- fp16 pieces
__device__ __forceinline__ unsigned tx_cvt2(float lo, float hi) {             // v_cvt_pk_f16_f32: two fp32 -> one dword of two fp16 (RNE)
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ float tx_lo2f(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[0]; }
__device__ __forceinline__ float tx_hi2f(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[1]; }
// (v_fma_mix_f32 would do x - (float)half in one instruction, but only inline asm reaches it and hipcc then pads every MFMA in
// front of such an asm with s_nop 11: measured 2x slower.)
// (x0, x1) already scaled -> the two packed pieces of the pair
__device__ __forceinline__ void tx_split2(float x0, float x1, unsigned& p1, unsigned& p2) {
    p1 = tx_cvt2(x0, x1);
    p2 = tx_cvt2(x0 - tx_lo2f(p1), x1 - tx_hi2f(p1));
}
// 2^e as a float, e clamped to the normal range
__device__ __forceinline__ float tx_exp2i(int e) { return __uint_as_float((uint32_t)(min(max(e, -126), 127) + 127) << 23); }
// t with bound * 2^t < 2^TX_TOP (bound >= 0; 0 and non-finite bounds give a harmless finite scale)
__device__ __forceinline__ int tx_scale_exp(float bound) {
    const int e = __builtin_amdgcn_frexp_expf(bound);           // bound = m 2^e, m in [0.5, 1)   (0 for bound = 0 / inf / nan)
    return min(max(TX_TOP - e, -100), 100);
}
// the other lane of this sample (lane ^ 32)
__device__ __forceinline__ float tx_partner(float v) { return __shfl_xor(v, 32, 64); }

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
    uint32_t par;                // parity of the stage boundaries taken: the odd ones carry the wait and the barrier (tx_boundary)
#ifdef TN_STAGE_STAMPS           // diagnostic build (tools/x3_stage_probe.py): s_memtime at every stage boundary of workgroup 0 / wave 0
    unsigned long long* smarks; int sn;
#endif
};

// This wave's DPW pieces of a stage are 1 KB each, consecutive in the stream and in the slot: piece i is the pending
// stage's base (+ 4 KB for i >= 4) with the instruction's immediate offset (i & 3) KB — the offset applies to the global AND
// the LDS address.  M0 (the DMA's LDS base) is compiler-reserved: saved and restored around the load.
template <int I>
__device__ __forceinline__ void tx_issue_piece(const unsigned char* src, uint32_t voff, uint32_t lds_dst) {
#ifdef TX_NO_DMA
    return;
#endif
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
    p.dst_off += TX_SLOT; if (p.dst_off == TX_RING_OF(TX_STAGE / DPW)) p.dst_off = 0;
}
// M0 for a whole deferred stage.  hipcc emits no M0 use of its own in these kernels (tests/test_kernel_resources.py checks the
// ISA for that), and every other DMA (tx_issue_piece, tn_glds16) saves and restores it, so M0 written at the stage boundary still
// holds the slot's LDS base when the stage's pieces are issued behind the MFMAs: ONE instruction per piece instead of five
// (s_mov x3, s_nop, load) — 320 fewer of the ~3500 instructions a wave issues per 256-wide layer, at one wave per SIMD.
__device__ __forceinline__ void tx_m0_set(uint32_t lds_dst) { asm volatile("s_mov_b32 m0, %0" :: "s"(lds_dst) : "memory"); }
template <int I>
__device__ __forceinline__ void tx_issue_piece_m0(const unsigned char* src, uint32_t voff) {
    static_assert(I < 4, "one M0 value per stage: pieces 0..3 (immediate offsets 0..3 KB)");
#ifdef TX_NO_DMA
    return;
#endif
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" :: "v"(voff), "s"(src), "n"(I * 1024) : "memory");
}
// ... or piece by piece behind the MFMAs of the stage that has just been published (tx_pass): back-to-back DMA instructions
// cost the wave more issue time than the same pieces spread over as many MFMA groups.
template <int DPW>
__device__ __forceinline__ void tx_defer_stage(PipeX& p) {
    p.pend_src = p.src + p.src_off; p.pend_dst = p.lds_dst0 + p.dst_off;
    tx_m0_set(p.pend_dst);
    p.src_off += TX_SLOT; if (p.src_off == p.stream_bytes) p.src_off = 0;
    p.dst_off += TX_SLOT; if (p.dst_off == TX_RING_OF(TX_STAGE / DPW)) p.dst_off = 0;
}

// Start of a stage: wait for this wave's DMA of the stage (LEAD-1 younger ones may stay in flight; the training kernels' stash
// stores, which are younger still, make this wait stricter than it needs to be, never laxer: vmcnt retires in issue order),
// barrier (the stage is readable by everyone, the slot of the previous one is free), issue stage + LEAD (DEFER: the caller
// issues its pieces with tx_issue_piece before the next boundary).
template <int DPW, bool DEFER>
__device__ __forceinline__ void tx_boundary(PipeX& p) {
    constexpr int LEAD = TX_LEAD_OF(TX_STAGE / DPW), RING = TX_RING_OF(TX_STAGE / DPW);
#ifdef TN_STAGE_STAMPS
    if (p.smarks && p.sn < 500) p.smarks[p.sn++] = __builtin_amdgcn_s_memtime();
#endif
#ifndef TX_BAR1
    // ONE barrier per TWO stages.  At an even boundary s the wave waits for its own DMA of stages s AND s+1 (stages s+2 .. s+4 may
    // stay in flight), then the barrier: both stages are readable by everyone, and everyone has issued its last read of stage s-1
    // (a boundary sits in front of a stage's first fragment read).  The odd boundary s+1 needs neither: stage s+1 has landed, and
    // the slot its DMA (stage s+6) overwrites held stage s-1, free since barrier s.  The even boundary's DMA (stage s+5) goes to the
    // slot of stage s-2.  What it costs: a stage's DMA has 4 stages instead of 5 to land; what it saves: half of the ~90 cycles a
    // wave loses at every barrier (stamps with and without barriers: 8.1 k / 9.0 k against 7.3 k / 8.3 k cycles per pass).
    p.par ^= 1u;
    if (p.par) {                                 // wave-uniform
        TN16_WAIT_VM(DPW * (LEAD - 2) + TX_WAIT_EXTRA);
#ifndef TX_NO_BARRIER    // diagnostic (races): what the stage barriers cost
        __builtin_amdgcn_s_barrier();
#endif
    }
#else
    TN16_WAIT_VM(DPW * (LEAD - 1) + TX_WAIT_EXTRA);
#ifndef TX_NO_BARRIER    // diagnostic (races): what the stage barriers cost
    __builtin_amdgcn_s_barrier();
#endif
#endif
    if constexpr (DEFER) tx_defer_stage<DPW>(p); else tx_issue_stage<DPW>(p);
    p.cur += TX_SLOT; if (p.cur == RING) p.cur = 0;
}

// Workgroup prologue: biases + scale records -> LDS, LEAD stages in flight, the first one landed; the first tx_boundary
// publishes stage 0.  `src` / `n_stage`: the stream this kernel walks (forward: packed, n.n_stage; dgrad: the backward stream).
// NEGB (training forward): the hidden layers' biases are stored as nb = 0 - b (b = +-0 -> +0), see tx_epi_fwd_value.
template <int NW, bool NEGB = false>
__device__ __forceinline__ void tx_prologue(PipeX& p, unsigned char* lds, const unsigned char* packed, const NetX3& n,
                                            const unsigned char* src, int n_stage, int lane, int wave, unsigned char* lds_bnd = nullptr) {
    constexpr int DPW = TX_STAGE / NW;
    {
        float* bl = reinterpret_cast<float*>(lds + TX_RING_OF(NW));
        const float* bg = reinterpret_cast<const float*>(packed + n.bias_off);
        const int nf = n.n_bias + (n.depth + 1) * TX_META;            // the scale records follow the biases
        const int nneg = NEGB ? n.depth * n.hidden : 0;
        for (int i = threadIdx.x; i < nf; i += NW * 64) bl[i] = i < nneg ? 0.0f - bg[i] : bg[i];
    }
    if (lds_bnd) {                                                   // the bound words start at zero
        for (int i = threadIdx.x; i < TX_BND_N * 64; i += NW * 64) reinterpret_cast<unsigned*>(lds_bnd)[i] = 0u;
        __syncthreads();
    }
    p.lane16 = lane * 16;
    p.src = src; p.src_off = 0; p.stream_bytes = (uint32_t)n_stage * TX_SLOT;
    p.dst_off = 0;
    p.lds_dst0 = (uint32_t)(uintptr_t)lds + wave * (DPW * 1024);
    p.voff = lane * 16 + wave * DPW * 1024;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < TX_LEAD_OF(NW); ++i) tx_issue_stage<DPW>(p);
    p.cur = TX_RING_OF(NW) - TX_SLOT;                   // the first boundary moves it onto slot 0
    p.par = 0u;                                  // ... and is a barrier boundary
#ifdef TN_STAGE_STAMPS
    p.smarks = nullptr; p.sn = 0;
#endif
}
#define TX_CONST_BYTES(n) ((uint32_t)(((n).n_bias + ((n).depth + 1) * TX_META + 3) / 4 * 4) * 4)
// Behind the ring and the constants: the network-input pieces of every wave's tile (forward kernels; [piece][k-step][lane] x 16 B).
// They are needed by layer 0 and again by the skip layer: parked in LDS they do not hold 32 registers through the layers between.
#define TX_ELDS_WAVE (TX_NP * TN16_KE * 1024)
// ... and the workgroup's running maxima of the per-sample bounds (training kernels): what the weight-gradient kernel scales its
// operand rows by.  TX_BND_N row groups (forward: H_0.., the input; dgrad: dZ_0.., the head gradient — TNB_* minus the kernel's
// first index) x 64 lanes: every lane keeps its own word (ds_max_u32 on per-lane addresses: one instruction, no conflict between
// the lanes of a wave; a wave-uniform address would make hipcc emit a 64-step scalar reduction loop and an EXEC-masked atomic in
// the middle of the MFMA stream), reduced over lanes and added to the stash's bound words once, at the end of the kernel.
#define TX_BND_OFF(n, nw, fwd) (TX_RING_OF(nw) + TX_CONST_BYTES(n) + ((fwd) ? (uint32_t)(nw) * TX_ELDS_WAVE : 0u))
#define TX_LDS_BYTES(n, nw, fwd) ((size_t)TX_BND_OFF(n, nw, fwd) + TX_BND_N * 64 * 4)
// lds_bnd: this LANE's word of row group 0; idx: local row-group index
__device__ __forceinline__ void tx_bound_note(unsigned char* lds_bnd, int idx, float bound) {
#ifdef TX_NO_BOUND_NOTE
    return;
#endif
    atomicMax(reinterpret_cast<unsigned*>(lds_bnd) + idx * 64, __float_as_uint(bound));      // bounds are >= 0: their bit patterns order like the values
}
// end of a training kernel: the workgroup's maxima -> the stash's bound words bounds[first + idx]
template <int NW>
__device__ __forceinline__ void tx_bound_flush(const unsigned char* lds_bnd0, float* __restrict__ bounds, int first, int lane, int wave) {
    __syncthreads();
    for (int idx = wave; idx < TX_BND_N; idx += NW) {
        unsigned v = reinterpret_cast<const unsigned*>(lds_bnd0)[idx * 64 + lane];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
        if (lane == 0 && v) atomicMax(reinterpret_cast<unsigned*>(bounds) + first + idx, v);
    }
}
// scale record of layer l (l = depth: heads) in the LDS copy: {2^-s, max|W|, max|b|, -}
template <int NW>
__device__ __forceinline__ f32x4 tx_meta(const unsigned char* lds, const NetX3& n, int l) {
    return *reinterpret_cast<const f32x4*>(lds + TX_RING_OF(NW) + (n.n_bias + l * TX_META) * 4);
}

// The activation of a wave's 32-sample tile: two fp16 pieces of every k-step operand.
template <int HID>
struct ActX { u32x4 p1[HID / 16], p2[HID / 16]; };          // packed fp16 pairs; dword q = values 2q, 2q+1 of the k-step
#define TX_H8(x) __builtin_bit_cast(f16x8, (x))
struct EncX { u32x4 p1[TN16_KE], p2[TN16_KE]; };

struct FragX { f16x8 a1, a2; };
__device__ __forceinline__ FragX tx_frag_load(const unsigned char* base, int tl) {
    FragX f;
    f.a1 = *reinterpret_cast<const f16x8*>(base + (tl * TX_NP + 0) * 1024);
    f.a2 = *reinterpret_cast<const f16x8*>(base + (tl * TX_NP + 1) * 1024);
    return f;
}
#define TX_PIN() __builtin_amdgcn_sched_barrier(0)
// sched_barrier pins the machine scheduler, not the IR: an epilogue step whose results are only consumed by the NEXT layer's
// passes (half A's, computed in pass B) is otherwise sunk to the end of the loop body, out of every MFMA shadow (seen in the
// ISA: 1100 vector instructions in one block behind pass B).  An empty volatile asm that "modifies" a step's results keeps the
// step where it was written.
#define TX_KEEP(x) asm volatile("" : "+v"(x))
#define TX_ACCN(HID) ((HID) / 64 * 2)             // a half's accumulators: [tile slot] leading products, [NH + tile slot] corrections
#define TX_SPG 3                                  // MFMAs (= hook slots) per (tile, k-step) group

// One half-pass of a layer: NK k-step records, each NTU tiles x 3 MFMAs into acc[tile slot] / acc[NH + tile slot].
// KIND 0: the TN16_KE input k-steps (B operand from E)   1: the hidden k-steps   3: heads — ONE output tile (acc[0], acc[NH]), so a
// record's NH tile slots carry NH consecutive k-steps of it instead (KH / NH records).
// A skip layer's half is a KIND 1 pass followed by a KIND 0 pass that accumulates (ZERO = false) — one copy of the long pass
// in the instruction cache instead of two.
// RPS = records per stage (2 for 256-wide, 4 for 128-wide nets); every pass is a whole number of stages.  ZERO: the
// accumulators start at zero.
// hook(integral_constant<slot>) is called behind MFMA number slot = (k * NTU + tile) * 3 + j: work to issue in its shadow.
// The A fragments of a (tile, k-step) GROUP (three MFMAs = 96 cycles) are read TX_FD groups ahead (behind the first MFMA of a
// group), and the stage boundary (wait, barrier, next DMA stage named) of every stage but the pass's first is taken TX_FD groups
// EARLY, so that the new stage's first fragments are read behind MFMAs as well.  TX_FD = 2 measured the same as 1 (621 / 815 /
// 722 us against 622 / 810 / 726 for render / training forward / dgrad) and costs eight registers: 1.  (The ring has the spare slot this needs: LEAD + 2 = NS; the
// slot a boundary hands to the DMA held the stage before the one whose last TX_FD groups are still running.)
// elds: this lane's slot of the wave's network-input pieces in LDS (KIND 0; tx_store_input), read one k-step ahead.
#ifndef TX_FD
#define TX_FD 1
#endif
template <int HID, int KIND, bool ZERO, int NW, typename Hook>
__device__ __forceinline__ void tx_pass(PipeX& p, const unsigned char* lds, const ActX<HID>& X, const unsigned char* elds,
                                        f32x16 (&acc)[TX_ACCN(HID)], Hook&& hook) {
    constexpr int NH = HID / 64, KH = HID / 16, RPS = TX_STAGE / (NH * TX_NP), DPW = TX_STAGE / NW;
    constexpr int NK = KIND == 0 ? TN16_KE : (KIND == 3 ? KH / NH : KH);
    constexpr int NTU = NH, NG = NK * NTU, GPS = RPS * NTU;          // groups of the pass / per stage
    static_assert(NK % RPS == 0, "a half-pass must be a whole number of stages");
    static_assert(TX_FD < GPS && TX_FD <= NG, "fragment prefetch distance");
    FragX fr[TX_FD + 1];
    f16x8 e1, e2, en1, en2;                                          // KIND 0: the B operand of this / the next k-step
    if constexpr (KIND == 0) { e1 = *reinterpret_cast<const f16x8*>(elds); e2 = *reinterpret_cast<const f16x8*>(elds + TN16_KE * 1024); }
    tx_boundary<DPW, true>(p);
    tn_static_for<TX_FD>([&](auto ic) TN_INLINE_LAMBDA {
        constexpr int i = decltype(ic)::value;
        fr[i] = tx_frag_load(lds + p.cur + p.lane16, i);
    });
    tn_static_for<NG>([&](auto fc) TN_INLINE_LAMBDA {
        constexpr int f = decltype(fc)::value, k = f / NTU, tl = f % NTU;
        constexpr int ks = KIND == 3 ? k * NH + tl : k;              // the k-step of this group's B operand
        constexpr int ta = KIND == 3 ? 0 : tl;                       // ... and its accumulator
        f16x8 b1, b2;
        if constexpr (KIND == 0) { b1 = e1; b2 = e2; }
        else { b1 = TX_H8(X.p1[ks]); b2 = TX_H8(X.p2[ks]); }
        constexpr int s0 = f * TX_SPG, fn = f + TX_FD;               // fn: the group whose fragments are read behind this one's first MFMA
        const FragX& cur = fr[f % (TX_FD + 1)];
        if constexpr (fn < NG && fn % GPS == 0) tx_boundary<DPW, true>(p);            // fn opens a stage: publish it first
        if constexpr (ZERO && ks == 0) { const f32x16 z = {}; acc[ta + NH] = TX_MFMA(cur.a2, b1, z); }
        else                          acc[ta + NH] = TX_MFMA(cur.a2, b1, acc[ta + NH]);
#ifdef TX_NO_FRAG
        fr[fn % (TX_FD + 1)] = cur;
#else
        if constexpr (fn < NG) fr[fn % (TX_FD + 1)] = tx_frag_load(lds + p.cur + p.lane16, fn % GPS);
#endif
        if constexpr (KIND == 0 && tl == 0 && k + 1 < NK) {
            en1 = *reinterpret_cast<const f16x8*>(elds + (k + 1) * 1024); en2 = *reinterpret_cast<const f16x8*>(elds + (TN16_KE + k + 1) * 1024);
        }
        hook(std::integral_constant<int, s0>{});     TX_PIN();
        acc[ta + NH] = TX_MFMA(cur.a1, b2, acc[ta + NH]); hook(std::integral_constant<int, s0 + 1>{});
        {   // this group's share of the DMA pieces of the pending stage (named by the last boundary), all issued before the next one
            constexpr int NS_ = NG / GPS;                                             // stages of the pass
            constexpr int sidx = (f + TX_FD) / GPS < NS_ - 1 ? (f + TX_FD) / GPS : NS_ - 1;
            constexpr int last_b = sidx == 0 ? 0 : sidx * GPS - TX_FD;               // the group in front of which the last boundary was taken
            constexpr int next_b = sidx + 1 <= NS_ - 1 ? (sidx + 1) * GPS - TX_FD : NG;
            constexpr int since = f - last_b, span = next_b - last_b;
            constexpr int PPS = (DPW + span - 2) / (span - 1);
            tn_static_for<PPS>([&](auto uc) TN_INLINE_LAMBDA {
                constexpr int i = since * PPS + decltype(uc)::value;
                if constexpr (i < DPW && since < span - 1) {
                    if constexpr (DPW <= 4) tx_issue_piece_m0<i>(p.pend_src, p.voff);
                    else                    tx_issue_piece<i>(p.pend_src, p.voff, p.pend_dst);
                }
            });
        }
        TX_PIN();
        if constexpr (ZERO && ks == 0) { const f32x16 z = {}; acc[ta] = TX_MFMA(cur.a1, b1, z); }
        else                          acc[ta] = TX_MFMA(cur.a1, b1, acc[ta]);
        hook(std::integral_constant<int, s0 + 2>{}); TX_PIN();
        if constexpr (KIND == 0 && tl == NTU - 1 && k + 1 < NK) { e1 = en1; e2 = en2; }
    });
#ifdef TX_NO_EPI      // ablation: nothing consumes the accumulators — keep the MFMAs alive
    tn_static_for<TX_ACCN(HID)>([&](auto ic) TN_INLINE_LAMBDA { const float d_ = acc[decltype(ic)::value][0]; asm volatile("" :: "v"(d_)); });
#endif
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
        f32x16 c = TX_MFMA(f.a2, TX_H8(Z.p1[0]), z);
        c = TX_MFMA(f.a1, TX_H8(Z.p2[0]), c);
        const f32x16 m = TX_MFMA(f.a1, TX_H8(Z.p1[0]), z);
        if constexpr (t < NH) { accA[t] = m; accA[t + NH] = c; } else { accB[t - NH] = m; accB[t] = c; }
    });
}

// ---- epilogues as micro-steps.  A half has NP = NH*8 register PAIRS (values 2pr, 2pr+1 of a tile), walked in ASCENDING order of
// the activation k-step they land in: pair I = tile slot I / 8, register pair I % 8 -> k-step xs = 2 t + pr / 4.  That order is what
// lets an epilogue ride on (almost) a whole pass: half A's pieces overwrite X[xs] right behind pass B's last read of it, half B's
// pieces are ready right before the next pass A's first read — no parking of values, no second window (round 3 walked the pairs
// in the order its sign-word shifts dictated and had to park half A's values in AGPRs: two extra moves per value).
// Sign words of the x3 stash (written here, read by the x3 dgrad epilogue only): the bit of register r of tile t is bit
// 31 - ((t & 1) * 16 + r) of word t / 2 — the first value shifted in ends up at the top.
template <int HID, int HALF, int I> struct TxPair {
    static constexpr int NH = HID / 64;
    static constexpr int tl = I / 8;                                  // accumulator slot in the half
    static constexpr int t = HALF * NH + tl;                          // n-tile
    static constexpr int pr = I % 8;
    static constexpr int r0 = 2 * pr, r1 = 2 * pr + 1;
    static constexpr int xs = 2 * t + pr / 4, xq = pr % 4;            // activation k-step and dword the pair lands in
    static constexpr int row0 = 32 * t + (r0 & 3) + 8 * (r0 >> 2), row1 = 32 * t + (r1 & 3) + 8 * (r1 >> 2);   // feature rows (+ 4h) of the stash
    static constexpr int bit0 = 31 - ((t & 1) * 16 + r0), bit1 = 31 - ((t & 1) * 16 + r1);                     // sign-word bits
};
// Per-sample scalars of the epilogue that is running: dsc = 2^-(s + t_in) turns the accumulator sum into the layer's output
// (NEGATIVE in the training forward, see tx_epi_fwd_value), osc = 2^t_out scales that output into the fp16 range for the split;
// l1 sums |output| pairwise (the next bound); kk = one dword the compiler cannot see through (tx_konst): as fp32 the largest
// finite multiple of ... (0x7f7fbc00 = 3.39e38, the ReLU's upper clamp), its low half the fp16 value -1.0 (the split's multiplier).
struct TxScale { float dsc, osc; f32x2 l1; float kk; };
// x - (float)half in ONE instruction: v_fma_mix_f32 (f16 * f16 + f32 -> f32) runs in an MFMA's shadow like a plain v_fma_f32
// (tools/microbench/epi_mix.hip), where round 3's v_cvt_f32_f16 + packed subtract cost three to four times as much.  hipcc selects
// it for fma(fpext(a), fpext(b), c) — but only if b is not a visible constant (it folds * -1 into a subtraction first) and only
// in functions compiled without packed fp32 (TX_PLAIN_F32: otherwise the SLP vectoriser forms v_pk_fma_f32 first).
__device__ __forceinline__ float tx_konst() { float k = __uint_as_float(0x7f7fbc00u); asm volatile("" : "+s"(k)); return k; }      // (an SGPR: wave-uniform, and the kernels have no VGPR to spare)
__device__ __forceinline__ float tx_sub_half(float x, _Float16 hv, float kk) {
    return __builtin_fmaf((float)hv, (float)__builtin_bit_cast(f16x2, kk)[0], x);
}
// Where a training kernel's epilogues put a tile's fp32 rows.  A wave that is alone on its SIMD pays for every store instruction
// it issues (tools/microbench/store_issue.hip: behind three MFMAs a global_store_dword with a 64-bit VGPR address holds the wave's
// issue for ~36 cycles, whatever the width; the buffer form — resource in SGPRs, ONE 32-bit VGPR offset — for ~14), so the 128
// stores per layer go through a per-tile buffer resource: base = the 32-sample block of the tile's first lane, the lane's offset =
// (its block - that block, 0 or 1) x block bytes + sample x 4 + 4 h rows, and padding lanes get an offset far behind num_records:
// the hardware drops their stores (no dump block, no branch, no EXEC games).
struct TxDst { __amdgpu_buffer_rsrc_t rs; uint32_t off; };
#define TX_DST_DROP 0x80000000u                     // + any row offset stays >= num_records and does not wrap
// m: this lane's (clamped, in-range) sample index; lane 0's is the tile's smallest.  rows: stash rows per block.
__device__ __forceinline__ TxDst tx_dst_tile(float* stash, int64_t rows, int64_t m, bool valid, int h) {
    const int blk = (int)(m >> 5), blk0 = __builtin_amdgcn_readfirstlane(blk);
    const uint32_t blk_bytes = (uint32_t)rows * 128u;
    TxDst d;
    d.rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(stash) + (int64_t)blk0 * rows * 128, 0, (int)(2u * blk_bytes), 0x00020000);
    d.off = valid ? (uint32_t)(blk - blk0) * blk_bytes + (uint32_t)(m & 31) * 4u + (uint32_t)h * 512u : TX_DST_DROP;
    return d;
}
__device__ __forceinline__ void tx_dst_store(const TxDst& d, uint32_t row_bytes, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), d.rs, (int)(d.off + row_bytes), 0, 2);      // aux 2 = nt, as TN_STASH_STORE
}
struct TxEpi { f32x2 v[4], c[4]; unsigned p1[4]; f32x2 b[4]; uint32_t msk; };      // up to 4 pairs in flight

// An epilogue is a chain of TX_NSTEP fine STEPS per register pair, each step one or two instructions per value that depend only
// on the step before.  With one wave per SIMD a dependent instruction cannot issue until its predecessor has left the pipeline
// (~8 cycles), and the wave issues in order — so an epilogue whose chain runs within one MFMA gap stretches the gap to 50
// cycles (measured: stamps per pass, PMC issue-stall counters).  The steps are therefore SOFTWARE-PIPELINED over the MFMA
// gaps: pair i runs step k behind MFMA number W0 + 3 start(i) + k, so that a gap carries steps of three different pairs — mutually
// independent instructions — and a pair's next step is a whole gap away.
//   steps 0..3 (part V): accumulators -> the layer's fp32 output (leading + correction, descale, bias + ReLU / sign-bit mask)
//   steps 4..8 (part S): [stash] L1 norm, scale, first pieces, residuals (v_fma_mix_f32), second pieces into the activation registers
// What an instruction costs a lone wave behind its MFMAs (tools/microbench/epi_mix.hip, pk_mfma.hip): about three plain VALU
// instructions per MFMA are free, every further one ~3.4 cycles; v_accvgpr_read ~2x, v_cvt_pk_f16_f32 ~2x, v_fma_mixlo/hi_f16 ~2.5x,
// packed fp32 ~3x (does not overlap the MFMA at all).  The chain below is the cheapest found: 19 instructions per pair (inference).
#define TX_NSTEP 9
#ifndef TX_GB256
#define TX_GB256 56                              // 256-wide: groups of the next pass A over which half B's epilogue is spread
#endif
#ifndef TX_GA256
#define TX_GA256 58                              // 256-wide: groups of pass B over which half A's epilogue is spread (from group TX_WA / 3 on)
#endif
#define TX_WA 6                                  // first slot of half A's window in pass B: X[0] has been read by then
#define TX_VSTEPS 4

// steps 4..8 (part S).  FWD: the values are ReLU outputs (the L1 norm needs no abs).
template <int HID, int HALF, int I, int K, bool TRAIN, bool FWD>
__device__ __forceinline__ void tx_epi_split(ActX<HID>& X, TxEpi& e, TxScale& sc, const TxDst& srow) {
    using P = TxPair<HID, HALF, I>;
    constexpr int u = I % 4;
    if constexpr (K == 4) {
#ifndef TX_NO_STASH   // ablation (wrong results): what the stash stores cost
        if constexpr (TRAIN) { tx_dst_store(srow, P::row0 * 128, e.v[u][0]); tx_dst_store(srow, P::row1 * 128, e.v[u][1]); }
#endif
        if constexpr (FWD) { sc.l1[0] += e.v[u][0]; sc.l1[1] += e.v[u][1]; }
        else { sc.l1[0] += fabsf(e.v[u][0]); sc.l1[1] += fabsf(e.v[u][1]); }
        TX_KEEP(sc.l1);
    } else if constexpr (K == 5) {
        e.v[u][0] *= sc.osc; e.v[u][1] *= sc.osc; TX_KEEP(e.v[u]);                 // (the unscaled value is dead: stored and summed in step 4)
    } else if constexpr (K == 6) {
        e.p1[u] = tx_cvt2(e.v[u][0], e.v[u][1]); TX_KEEP(e.p1[u]);
    } else if constexpr (K == 7) {
        const f16x2 hp = __builtin_bit_cast(f16x2, e.p1[u]);
        e.v[u][0] = tx_sub_half(e.v[u][0], hp[0], sc.kk); e.v[u][1] = tx_sub_half(e.v[u][1], hp[1], sc.kk); TX_KEEP(e.v[u]);
    } else if constexpr (K == 8) {
        unsigned q2 = tx_cvt2(e.v[u][0], e.v[u][1]);
#ifdef TX_X_AGPR     // experiment (measured: no gain, 9.3 k / 8.05 k cycles per pass against 9.0 k / 8.1 k): the activation pieces written to AGPRs,
        // which the MFMA reads as its B operand directly — instead of the allocator's own parking of pieces in AGPRs with reloads
        unsigned o1, o2;
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(o1) : "v"(e.p1[u]));
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(o2) : "v"(q2));
        X.p1[P::xs][P::xq] = o1;
        X.p2[P::xs][P::xq] = o2;
#else
        TX_KEEP(q2);
        X.p1[P::xs][P::xq] = e.p1[u];
        X.p2[P::xs][P::xq] = q2;
#endif
    }
}

// An accumulator element into a VGPR, from WHERE IT IS: an AGPR.  Left to itself the register allocator gives the two halves' accumulators
// one set of 128 AGPRs and, at every pass boundary, copies the half that has just been finished into VGPRs for the epilogue to read — 128
// v_accvgpr_read in a burst with no MFMA to hide behind, ~900 cycles per half-pass (stage stamps: the stage around a pass boundary took
// 1.8-1.9 k cycles, every other one 0.98 k) — and then runs out of VGPRs and parks activation pieces in AGPRs.  An asm read with an "a"
// operand keeps each half in its own AGPRs until the epilogue step that needs the value, one read per value in an MFMA's shadow.
// ASM = false (tx_drain: the epilogue runs right behind the MFMAs that produce the values): plain C++, the compiler inserts the MFMA ->
// VALU wait states itself; the windows start >= 6 MFMAs (192 cycles) behind the last MFMA of the half they read.
#ifndef TX_ASM_ACC_128
#define TX_ASM_ACC_128 0          // 128-wide kernels: measured below
#endif
template <bool ASM>
__device__ __forceinline__ float tx_acc_get(const f32x16& a, int r) {
    if constexpr (ASM) {
        float x;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a[r]));
        return x;
    } else {
        return a[r];
    }
}

// Forward part V (steps 0..3): leading + correction, descale + bias (fp32, LDS byte offset vb + 16 h: this layer's biases for rows
// 4h..), ReLU.  TRAIN: the ReLU sign bits ride on a NEGATED pre-activation: the LDS table holds nb = 0 - b (so that b = +-0 gives
// +0) and dsc is negative, nz = fma(acc, dsc, nb) = -z; then bit 31 of nz IS "z > 0" (z = +-0 gives nz = +0: not set, as
// torch's relu'(0) = 0), one v_alignbit per value shifts it into the sign word, and ReLU is med3(-nz, 0, 3.39e38) — source modifier,
// ONE instruction: fmaxf behind TX_KEEP needs a canonicalising v_max first, and med3 against a visible inf is folded into exactly
// that fmaxf; the clamp is the opaque constant of TxScale (an activation of 3.4e38 has overflowed every product downstream anyway).  (Round 3: v_add_u32 0x7fffffff + v_alignbit on the ReLU output.)
template <int HID, int HALF, int I, int K, bool TRAIN, bool AR = false>
__device__ __forceinline__ void tx_epi_fwd_value(const f32x16 (&acc)[TX_ACCN(HID)], TxEpi& e, const TxScale& sc, const unsigned char* lds, uint32_t vb,
                                                 uint32_t* __restrict__ mword) {
    using P = TxPair<HID, HALF, I>;
    constexpr int u = I % 4;
    if constexpr (K == 0) {
        e.b[u] = *reinterpret_cast<const f32x2*>(lds + vb + P::row0 * 4);
        e.v[u][0] = tx_acc_get<AR>(acc[P::tl], P::r0); e.v[u][1] = tx_acc_get<AR>(acc[P::tl], P::r1);
        e.c[u][0] = tx_acc_get<AR>(acc[P::tl + P::NH], P::r0); e.c[u][1] = tx_acc_get<AR>(acc[P::tl + P::NH], P::r1);
        TX_KEEP(e.v[u]); TX_KEEP(e.c[u]);
    } else if constexpr (K == 1) {
        e.v[u][0] += e.c[u][0]; e.v[u][1] += e.c[u][1]; TX_KEEP(e.v[u]);
    } else if constexpr (K == 2) {
        e.c[u][0] = __builtin_fmaf(e.v[u][0], sc.dsc, e.b[u][0]); e.c[u][1] = __builtin_fmaf(e.v[u][1], sc.dsc, e.b[u][1]); TX_KEEP(e.c[u]);
    } else if constexpr (K == 3) {
        if constexpr (TRAIN) {
#ifndef TX_NO_SIGN    // ablation (wrong results): what the sign words cost
            if constexpr (I % 16 == 0) e.msk = 0u;
            e.msk = __builtin_amdgcn_alignbit(e.msk, __float_as_uint(e.c[u][0]), 31);
            e.msk = __builtin_amdgcn_alignbit(e.msk, __float_as_uint(e.c[u][1]), 31);
            if constexpr (I % 16 == 15) mword[P::t / 2] = e.msk;
#endif
            e.v[u][0] = __builtin_amdgcn_fmed3f(-e.c[u][0], 0.0f, sc.kk);
            e.v[u][1] = __builtin_amdgcn_fmed3f(-e.c[u][1], 0.0f, sc.kk);
        } else {
            // ReLU as a signed-integer max: negative floats (and -0) are negative integers.
            e.v[u][0] = __int_as_float(max(__float_as_int(e.c[u][0]), 0)); e.v[u][1] = __int_as_float(max(__float_as_int(e.c[u][1]), 0));
        }
        TX_KEEP(e.v[u]);
    }
}
// Backward part V: leading + correction, descale, ReLU backward with the forward's sign bits (mw: the words of the layer this
// activation gradient belongs to).
template <int HID, int HALF, int I, int K, bool AR = false>
__device__ __forceinline__ void tx_epi_bwd_value(const f32x16 (&acc)[TX_ACCN(HID)], TxEpi& e, const TxScale& sc, const uint32_t (&mw)[HID / 64]) {
    using P = TxPair<HID, HALF, I>;
    constexpr int u = I % 4;
    if constexpr (K == 0) {
        e.v[u][0] = tx_acc_get<AR>(acc[P::tl], P::r0); e.v[u][1] = tx_acc_get<AR>(acc[P::tl], P::r1);
        e.c[u][0] = tx_acc_get<AR>(acc[P::tl + P::NH], P::r0); e.c[u][1] = tx_acc_get<AR>(acc[P::tl + P::NH], P::r1);
        TX_KEEP(e.v[u]); TX_KEEP(e.c[u]);
    } else if constexpr (K == 1) {
        e.v[u][0] += e.c[u][0]; e.v[u][1] += e.c[u][1]; TX_KEEP(e.v[u]);
        e.b[u][0] = __int_as_float(__builtin_amdgcn_sbfe((int)mw[P::t / 2], P::bit0, 1));       // the mask: 0 / all ones
        e.b[u][1] = __int_as_float(__builtin_amdgcn_sbfe((int)mw[P::t / 2], P::bit1, 1));
        TX_KEEP(e.b[u]);
    } else if constexpr (K == 2) {
        e.v[u][0] *= sc.dsc; e.v[u][1] *= sc.dsc; TX_KEEP(e.v[u]);
    } else if constexpr (K == 3) {
        e.v[u][0] = __int_as_float(__float_as_int(e.v[u][0]) & __float_as_int(e.b[u][0]));
        e.v[u][1] = __int_as_float(__float_as_int(e.v[u][1]) & __float_as_int(e.b[u][1]));
        TX_KEEP(e.v[u]);
    }
}

// Epilogue steps [K0, K1) of NP pairs over the slots of a pass from W0 on, SPS consecutive steps per slot.  G = groups in which
// pairs start: G <= NP: PPG = ceil(NP / G) pairs start per group; G > NP ("spread"): pair i starts in group floor(i G / NP), some
// groups start none.  Pair i runs steps K0 + q SPS .. behind MFMA slot W0 + 3 start(i) + q.
// SPS = 1 is the software pipeline of the comment above (three pairs in flight); SPS = 4 runs a pair's chain inside its
// own group (several pairs per group interleave instead).  f(integral_constant<I>, integral_constant<K>).
// The caller checks the window against the pass that carries it: tx_half_a_ok / tx_half_b_ok below.
template <int W0, int G, int NP, int K0, int K1, int SPS, typename F>
__device__ __forceinline__ auto tx_window(F&& f) {
    return [&f](auto sc) TN_INLINE_LAMBDA {
        constexpr int s = decltype(sc)::value;
        constexpr bool SPREAD = G > NP;
        constexpr int PPG = SPREAD ? 1 : (NP + G - 1) / G, NQ = (K1 - K0 + SPS - 1) / SPS, DEPTH = (NQ + TX_SPG - 1) / TX_SPG;
        static_assert(PPG * DEPTH <= 4, "more pairs in flight than TxEpi holds");
#ifdef TX_NO_EPI
        return;
#endif
        if constexpr (s >= W0 && s < W0 + TX_SPG * (G - 1) + NQ) {
            constexpr int rel = s - W0;
            tn_static_for<DEPTH>([&](auto dc) TN_INLINE_LAMBDA {                 // the group that started d groups ago is at slot-step rel % 3 + 3 d
                constexpr int g = rel / TX_SPG - decltype(dc)::value, q = rel % TX_SPG + TX_SPG * decltype(dc)::value;
                if constexpr (g >= 0 && g < G && q < NQ)
                    tn_static_for<SPS>([&](auto kc) TN_INLINE_LAMBDA {
                        constexpr int k = K0 + q * SPS + decltype(kc)::value;
                        if constexpr (k < K1) {
                            if constexpr (SPREAD) {
                                constexpr int i = (g * NP + G - 1) / G;             // the pair that starts in group g, if any
                                if constexpr (i < NP && (i * G) / NP == g) f(std::integral_constant<int, i>{}, std::integral_constant<int, k>{});
                            } else {
                                tn_static_for<PPG>([&](auto uc) TN_INLINE_LAMBDA {
                                    constexpr int i = g * PPG + decltype(uc)::value;
                                    if constexpr (i < NP) f(std::integral_constant<int, i>{}, std::integral_constant<int, k>{});
                                });
                            }
                        }
                    });
            });
        }
    };
}
// slot (relative to W0) behind which pair i of a window runs its LAST step
template <int G, int NP, int SPS> constexpr int tx_last_slot(int i) {
    const int start = G > NP ? (i * G) / NP : i / ((NP + G - 1) / G);
    return TX_SPG * start + (TX_NSTEP - 1) / SPS;
}
// Half A's epilogue rides on pass B of its own layer and writes X[xs], xs = pair / 4, which pass B reads until the last tile of
// k-step xs (slot 3 (NH xs + NH - 1) + 2): the pair's last step must come later — and inside the pass (NG groups).
template <int HID, int W0, int G, int SPS> constexpr bool tx_half_a_ok() {
    constexpr int NH = HID / 64, NP = NH * 8, NG = HID / 16 * NH;
    for (int i = 0; i < NP; ++i) {
        const int last = W0 + tx_last_slot<G, NP, SPS>(i), xs = i / 4;
        if (last <= TX_SPG * (NH * xs + NH - 1) + 2 || last >= TX_SPG * NG) return false;
    }
    return true;
}
// Half B's epilogue rides on the NEXT pass A, which reads X[xs], xs = 2 NH + pair / 4, from slot 3 NH xs on: the pair must be
// through one group earlier.
template <int HID, int G, int SPS> constexpr bool tx_half_b_ok() {
    constexpr int NH = HID / 64, NP = NH * 8;
    for (int i = 0; i < NP; ++i) {
        const int last = tx_last_slot<G, NP, SPS>(i), xs = 2 * NH + i / 4;
        if (last + TX_SPG >= TX_SPG * NH * xs) return false;
    }
    return true;
}
// A whole epilogue with nothing to hide behind (steps 0 .. NJ-1 of every pair).
template <int NP, int NJ, typename F>
__device__ __forceinline__ void tx_drain(F&& f) {
    tn_static_for<NP>([&](auto ic) TN_INLINE_LAMBDA {
        tn_static_for<NJ>([&](auto jc) TN_INLINE_LAMBDA { f(ic, jc); });
    });
}

// The network input of one sample as the B operand of the input k-steps (slot map: tnerf_internal.h), scaled by `esc` and split.
// value(integral_constant<a>) is the fp32 value of input step a = 8u + e (the step numbering of the fp32 path's pairing: the
// training stash).
template <typename Val>
__device__ __forceinline__ void tx_split_input(EncX& E, float esc, Val&& value) {
    tn_static_for<TN16_KE>([&](auto uc) TN_INLINE_LAMBDA {
        constexpr int u = decltype(uc)::value;
        u32x4 w1, w2;
        tn_static_for<4>([&](auto qc) TN_INLINE_LAMBDA {
            constexpr int q = decltype(qc)::value;
            unsigned a_, b_;
            tx_split2(value(std::integral_constant<int, 8 * u + 2 * q>{}) * esc, value(std::integral_constant<int, 8 * u + 2 * q + 1>{}) * esc, a_, b_);
            w1[q] = a_; w2[q] = b_;
        });
        E.p1[u] = w1; E.p2[u] = w2;
    });
}
// The input pieces to / in this lane's LDS slots (only this lane ever touches them: no barrier).
__device__ __forceinline__ void tx_store_input(unsigned char* elds, const EncX& E) {
#pragma unroll
    for (int u = 0; u < TN16_KE; ++u) {
        *reinterpret_cast<u32x4*>(elds + u * 1024) = E.p1[u];
        *reinterpret_cast<u32x4*>(elds + (TN16_KE + u) * 1024) = E.p2[u];
    }
}
// ... scaled by 2^d (d <= 0) in place, as two fp16 factors so that neither is subnormal
__device__ __forceinline__ void tx_rescale_input(unsigned char* elds, int d) {
    const int d1 = max(d, -14), d2 = max(d - d1, -14);
    const _Float16 f1 = (_Float16)tx_exp2i(d1), f2 = (_Float16)tx_exp2i(d2);
    const f16x8 m1 = {f1, f1, f1, f1, f1, f1, f1, f1}, m2 = {f2, f2, f2, f2, f2, f2, f2, f2};
#pragma unroll
    for (int i = 0; i < TX_NP * TN16_KE; ++i) {
        f16x8* q = reinterpret_cast<f16x8*>(elds + i * 1024);
        *q = (*q * m1) * m2;
    }
}

// PositionalEncoding(L, include_input=True) of one point, fp32-accurate (tn_sincos, as the fp32 kernels), in the step numbering
// of the training stash: encf[a], a = 8u + e.                                                reference src/encoding.py:27-33
__device__ __forceinline__ void tx_encode(float px, float py, float pz, int Lf, int h, float (&encf)[8 * TN16_KE]) {
    // Lf is a kernel argument: every `slot < 3 Lf` below is a loop-invariant wave-uniform condition, and hipcc hoists all ~60 of them out of
    // the tile loop as 64-bit lane masks — 120 SGPRs, spilled to VGPR lanes.  Opaque here, they are evaluated per tile (scalar compares).
    asm volatile("" : "+s"(Lf));
    // every slot is DEFINED before the conditional writes below (a slot written only under run-time conditions the compiler cannot
    // prove exhaustive would be an undefined value on the paths it cannot rule out)
    tn_static_for<8 * TN16_KE>([&](auto ac) TN_INLINE_LAMBDA { encf[decltype(ac)::value] = 0.0f; });
    // six arguments (two frequencies x three coordinates) per tn_sincos_n call: independent chains that fill each other's latency
    constexpr int NK = (8 * TN16_KE + 2) / 3;                      // frequencies that can appear in the slots
    tn_static_for<(NK + 1) / 2>([&](auto kc) TN_INLINE_LAMBDA {
        constexpr int k0 = 2 * decltype(kc)::value;
        if (k0 < Lf) {                                             // wave-uniform
            const float f0 = (float)(1u << (k0 < 31 ? k0 : 0)), f1 = (float)(1u << (k0 + 1 < 31 ? k0 + 1 : 0));
            const float x[6] = {px * f0, py * f0, pz * f0, px * f1, py * f1, pz * f1};
            float sn[6], cs[6];
            tn_sincos_n<6>(x, sn, cs);
            tn_static_for<6>([&](auto ic) TN_INLINE_LAMBDA {
                constexpr int i = decltype(ic)::value, a = 3 * k0 + i;
                if constexpr (a < 8 * TN16_KE) encf[a] = h ? cs[i] : sn[i];
            });
        }
    });
    // behind the 3 Lf sin / cos slots: the input itself (x | y in the two lane halves, then z | 0), zeros after it
    tn_static_for<8 * TN16_KE>([&](auto ac) TN_INLINE_LAMBDA {
        constexpr int a = decltype(ac)::value;
        if (a >= 3 * Lf) encf[a] = a == 3 * Lf ? (h ? py : px) : (a == 3 * Lf + 1 ? (h ? 0.0f : pz) : 0.0f);
    });
}
