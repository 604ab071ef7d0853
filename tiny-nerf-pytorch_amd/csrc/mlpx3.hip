// "x3" chain kernels: the fused ray -> sample -> encode -> MLP -> composite path (reference src/train.py:46-56, :114-121)
// with the MLP's fp32 products formed from three partial products on the fp16 matrix pipe (mlpx3_core.hpp).  Same inputs, outputs, sample bins,
// encoder arithmetic (fp32-accurate sin/cos), compositing and — when training — the same block-major fp32 stash (activations,
// ReLU sign bits, head outputs, loss gradient) as the fp32-MFMA kernels of mlp_fwd.hip, so the dgrad / weight-gradient kernels
// and every caller are unchanged.  One persistent workgroup per CU — 4 waves (one per SIMD, 512-register budget) at 256 wide, 8 waves (two
// per SIMD, 256 registers each) in the 128-wide tile / inference kernels (TxCfg, mlpx3_core.hpp);
// a wave owns one ray (or one 32-sample tile) at a time and marches it 32 samples per pass over the record stream.  A layer is two half-passes whose
// epilogues ride in each other's MFMA shadows (mlpx3_core.hpp).
#include "mlpx3_core.hpp"
#include "mlp_args.hpp"

// The backward kernels are compiled WITHOUT packed fp32 VALU instructions.  A v_pk_mul / v_pk_add / v_pk_fma_f32 does not run in the
// shadow of the wave's MFMA: behind three v_mfma_f32_32x32x16_f16 (96 cycles) two of them per MFMA make the loop 152 cycles, the four
// plain instructions they stand for 104 (tools/microbench/pk_mfma.hip).  hipcc's post-RA peephole unpacks the ones it believes to be in a
// shadow and leaves the rest (72 per layer in the dgrad loop); with the feature off the 8x256 dgrad kernel runs 2.7 % faster.  The
// forward kernels keep it: there the extra instructions cost as much as the stalls (measured: +-0).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TX_ALLOW_PK)      // (hipcc's host pass of this file does not know the feature and would warn; TX_ALLOW_PK: diagnostic A/B builds)
#define TX_PLAIN_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define TX_PLAIN_F32
#endif

// Diagnostic builds (-DTN_STAMPS, tools/x3_stamp_probe.py): cycles a wave spends in the layer walks and in the epilogues, and the
// shader clock (s_memtime / s_memrealtime).  The values go to a.f.stamps only; the product build has none of this.
#ifdef TN_STAMPS
struct TxProf { unsigned long long walk = 0, epi = 0, t = 0; unsigned long long* marks = nullptr; int n = 0; };
#define TX_PROF_BEGIN(pf) (pf).t = __builtin_amdgcn_s_memtime()
#define TX_PROF_ADD(pf, field) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); (pf).field += n_ - (pf).t; (pf).t = n_; } while (0)
// a time stamp per pass of the FIRST tile of workgroup 0 / wave 0 (marks != nullptr there only)
#ifdef TN_STAGE_STAMPS      // (the stage stamps of tx_boundary are taken instead: no marks between the passes)
#define TX_PROF_MARK(pf) do {} while (0)
#else
#define TX_PROF_MARK(pf) do { if ((pf).marks && (pf).n < 60) { (pf).marks[(pf).n++] = __builtin_amdgcn_s_memtime(); } } while (0)
#endif
#else
struct TxProf {};
#define TX_PROF_BEGIN(pf) do {} while (0)
#define TX_PROF_ADD(pf, field) do {} while (0)
#define TX_PROF_MARK(pf) do {} while (0)
#endif

struct FwdX3Args {
    FwdArgs f;                      // layout (stash rows), ray source, sampling, outputs, stash, loss — as the fp32 kernels
    NetX3 n;
    const unsigned char* packed3;   // record stream + biases (tnerf_mlp_pack_x3)
};

// Per-sample facts of the network input: encmax >= every |input value| and >= 1, l1 >= the L1 norm of the input, te = the
// exponent its pieces are scaled by.
struct TxIn { float encmax, l1; int te; };

// The network input rows of the training stash (the fp32 path's pairing: step st = 8u + e), written as soon as the values exist —
// before they are split — so that no copy of them has to outlive the split.
__device__ __forceinline__ void tx_stash_input(const FwdX3Args& a, int h, int lane, const float (&encf)[8 * TN16_KE], int64_t m, bool valid) {
    const MlpLayout& L = a.f.L;
    const int64_t ms = valid ? m : a.f.Mp + (lane & 31);           // padding lanes write to the dump block
    float* __restrict__ q = tn_stash_at(a.f.stash, L.stash_rows, ms) + (int64_t)(L.enc_row0 + h) * 32;      // row enc_row0 + 2 st + h
    // NE is 20 (in_dim <= 40) or 32 input steps: one uniform branch, then straight-line stores with immediate offsets
    int ne = L.NE; asm volatile("" : "+s"(ne));                    // (opaque: see tx_encode — no hoisted lane masks)
    if (ne == 20) {
        tn_static_for<20>([&](auto sc) TN_INLINE_LAMBDA { constexpr int st = decltype(sc)::value; TN_STASH_STORE(&q[2 * st * 32], encf[st]); });
    } else {
        tn_static_for<8 * TN16_KE>([&](auto sc) TN_INLINE_LAMBDA {
            constexpr int st = decltype(sc)::value;
            if (st < ne) TN_STASH_STORE(&q[2 * st * 32], encf[st]);
        });
    }
}

// The network for one 32-sample tile.  m: this lane's sample index in the stash (valid if `valid`).  res[4]: r,g,b after
// sigmoid, sigma after ReLU (lane-half 0).  E: this lane's LDS slots of the input pieces, scaled by 2^in.te (rescaled in place for
// the skip layer).
template <int HID, bool TRAIN, int NW, bool ACC_ASM = true>
__device__ __forceinline__ void tx_mlp_tile(PipeX& p, const unsigned char* lds, const FwdX3Args& a, int h, int lane, unsigned char* E, const TxIn& in,
                                            int64_t m, bool valid, float (&res)[4], TxProf& pf, unsigned char* lds_bnd) {
    constexpr int NT = HID / 32;
    const MlpLayout& L = a.f.L;
    const int depth = a.n.depth, skip_at = a.n.skip_at;
    float* __restrict__ stash = a.f.stash;
    const int64_t Mp = a.f.Mp;
    const int64_t ms = valid ? m : Mp + (lane & 31);               // padding lanes write to the dump block: stores need no branch
    // (this lane's sign words of layer 0; point_at steps the pointer from layer to layer — no second copy of it lives through the walk)
    uint32_t* __restrict__ mword = TRAIN ? reinterpret_cast<uint32_t*>(stash + TN_STASH_BODY_FLOATS(L, Mp)) + (2 * ms + h) * (NT / 2) : nullptr;
    constexpr int NH = NT / 2, KH = HID / 16, NP = NH * 8, G2 = KH / 2 * NH;
    ActX<HID> X;
    f32x16 accA[TX_ACCN(HID)], accB[TX_ACCN(HID)];
    TxEpi es;
    TxScale sc{0.0f, 0.0f, {0.0f, 0.0f}, tx_konst()};
    int t_prev = in.te;                                            // exponent of the scale of the pieces the running passes consume
    // per-lane stash pointers of the layer whose epilogue is running (training)
    TxDst srow{};
    if constexpr (TRAIN) srow = tx_dst_tile(stash, L.stash_rows, m, valid, h);       // (.off: + the layer's first row, stepped by point_at)
    uint32_t vbe = TX_RING_OF(NW) + 16u * h;                       // LDS offset of the biases of the layer whose epilogue is running (+ HID * 4 per layer)
    // Layer l's epilogues are about to start: its input's L1 norm is complete (the previous layer's half B epilogue ended in the
    // middle of pass A), so the bound on its outputs — and with it the scale of its pieces — is known.
    // (called for l = 0, 1, .. depth-1 in this order: the per-layer addresses are STEPPED, so that only one copy of each is live)
    auto point_at = [&](int l) TN_INLINE_LAMBDA {
        if (l > 0) vbe += HID * 4;
        if constexpr (TRAIN) {
            srow.off += (uint32_t)(L.h_row0[l] - (l > 0 ? L.h_row0[l - 1] : 0)) * 128u;
            if (l > 0) mword += (Mp + 32) * NT;
        }
        const f32x4 mt = tx_meta<NW>(lds, a.n, l);                 // {2^-s, max|W|, max|b|}
        const float l1_own = sc.l1[0] + sc.l1[1];
        float l1_in = l == 0 ? in.l1 : l1_own + tx_partner(l1_own);
        if (l > 0 && l == skip_at) l1_in += in.l1;
        float bound = __builtin_fmaf(mt[1], l1_in, mt[2]);
        if (l + 1 == skip_at) bound = fmaxf(bound, in.encmax);     // the skip layer's input pieces share this layer's output scale
        const int t_out = tx_scale_exp(bound);
        if constexpr (TRAIN) tx_bound_note(lds_bnd, TNB_H(l), bound);          // this layer's rows of the stash, for the weight-gradient kernel
        const float dsc = mt[0] * tx_exp2i(-t_prev);
        sc.dsc = TRAIN ? -dsc : dsc;                               // training: negated pre-activations against the negated bias table (tx_epi_fwd_value)
        sc.osc = tx_exp2i(t_out); sc.l1 = f32x2{0.0f, 0.0f};
        t_prev = t_out;
    };
    if constexpr (TRAIN) tx_bound_note(lds_bnd, TNB_ENC, in.encmax);
    // steps 0..3 = part V, 4..8 = part S    (mlpx3_core.hpp)
#ifdef TX_NO_ASM_ACC_READ
    constexpr bool AR = false;
#else
    constexpr bool AR = (HID == 256 || TX_ASM_ACC_128) && ACC_ASM;      // (mlpx3_core.hpp, tx_acc_get)
#endif
    auto epi_full = [&](auto halfc, auto& acc, auto ic, auto kc) TN_INLINE_LAMBDA {
        constexpr int HALF = decltype(halfc)::value, I = decltype(ic)::value, K = decltype(kc)::value;
        if constexpr (K < TX_VSTEPS) tx_epi_fwd_value<HID, HALF, I, K, TRAIN, AR>(acc, es, sc, lds, vbe, mword);
        else tx_epi_split<HID, HALF, I, K, TRAIN, true>(X, es, sc, srow);
    };
    auto epiA = [&](auto ic, auto kc) TN_INLINE_LAMBDA { epi_full(std::integral_constant<int, 0>{}, accA, ic, kc); };
    auto epiB = [&](auto ic, auto kc) TN_INLINE_LAMBDA { epi_full(std::integral_constant<int, 1>{}, accB, ic, kc); };
    // 256-wide: one pair per group at most, the chain pipelined over the gaps (SPS 1) and spread over (almost) the whole pass;
    // 128-wide: two pairs per group, chains inside the group
    constexpr int SPF = HID == 256 ? 1 : 4;
    constexpr int GB = HID == 256 ? TX_GB256 : G2, GA = HID == 256 ? TX_GA256 : G2;
    static_assert(tx_half_b_ok<HID, GB, SPF>(), "half B's epilogue window overruns the first read of its pieces");
    static_assert(tx_half_a_ok<HID, TX_WA, GA, SPF>(), "half A's epilogue window writes an activation slot pass B still reads, or overruns the pass");
    auto none = [](auto) TN_INLINE_LAMBDA {};
    TX_PROF_BEGIN(pf);
    TX_PROF_MARK(pf);
    // layer 0: half A bare, half A's epilogue behind half B
    tx_pass<HID, 0, true, NW>(p, lds, X, E, accA, none);
    point_at(0);
    TX_PROF_MARK(pf);
    tx_pass<HID, 0, true, NW>(p, lds, X, E, accB, tx_window<0, TN16_KE * NH, NP, 0, TX_NSTEP, 4>(epiA));
    TX_PROF_MARK(pf);
    // layer l: half B's epilogue of layer l-1 behind pass A (its pieces are read from the middle of the pass on), half A's of layer l
    // behind pass B (its pieces replace the input k-steps pass B has just read)
    for (int l = 1; l < depth; ++l) {
        if (l == skip_at) tx_rescale_input(E, t_prev - in.te);     // the skip layer consumes the input at its own input's scale
#ifdef TX_SERIAL_EPI   // diagnostic (wrong results): the passes bare, the epilogues behind them with nothing to hide in — marks: pass A | epilogue B | pass B | epilogue A
        tx_pass<HID, 1, true, NW>(p, lds, X, E, accA, none);
        if (l == skip_at) tx_pass<HID, 0, false, NW>(p, lds, X, E, accA, none);
        TX_PROF_MARK(pf);
        tx_drain<NP, TX_NSTEP>(epiB);
        TX_PROF_MARK(pf);
        point_at(l);
        tx_pass<HID, 1, true, NW>(p, lds, X, E, accB, none);
        if (l == skip_at) tx_pass<HID, 0, false, NW>(p, lds, X, E, accB, none);
        TX_PROF_MARK(pf);
        tx_drain<NP, TX_NSTEP>(epiA);
        TX_PROF_MARK(pf);
#else
        tx_pass<HID, 1, true, NW>(p, lds, X, E, accA, tx_window<0, GB, NP, 0, TX_NSTEP, SPF>(epiB));
        if (l == skip_at) tx_pass<HID, 0, false, NW>(p, lds, X, E, accA, none);                    // + W_l[:, hidden:] . encoding
        TX_PROF_MARK(pf);
        point_at(l);
        tx_pass<HID, 1, true, NW>(p, lds, X, E, accB, tx_window<TX_WA, GA, NP, 0, TX_NSTEP, SPF>(epiA));
        if (l == skip_at) tx_pass<HID, 0, false, NW>(p, lds, X, E, accB, none);
        TX_PROF_MARK(pf);
#endif
    }
    // heads (tile slot 0 of accA): the last layer's half B epilogue must be through before k-step KH/2
    tx_pass<HID, 3, true, NW>(p, lds, X, E, accA, tx_window<0, KH / 2, NP, 0, TX_NSTEP, 4>(epiB));
    TX_PROF_MARK(pf);
    TX_PROF_ADD(pf, walk);
    // heads: rows 0..2 = rgb.0 (sigmoid), row 3 = sigma.0 (ReLU)                                   nerf.py:39-40
    const f32x4 hb = *reinterpret_cast<const f32x4*>(lds + TX_RING_OF(NW) + depth * HID * 4);
    const float dh = tx_meta<NW>(lds, a.n, depth)[0] * tx_exp2i(-t_prev);
#pragma unroll
    for (int i = 0; i < 3; ++i) res[i] = 1.0f / (1.0f + expf(-(__builtin_fmaf(accA[0][i] + accA[NH][i], dh, hb[i]))));
    res[3] = fmaxf(__builtin_fmaf(accA[0][3] + accA[NH][3], dh, hb[3]), 0.0f);
}

template <int HID, bool TRAIN>
__global__ TX_PLAIN_F32 __launch_bounds__((TRAIN ? TxCfg<HID>::RAY : TxCfg<HID>::RENDER) * 64, 1) void k_renderx3(FwdX3Args a) {
    constexpr int NW = TRAIN ? TxCfg<HID>::RAY : TxCfg<HID>::RENDER;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
#ifdef TN_STAMPS
    const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    RaySource rs = a.f.rs; SampleArgs sa = a.f.sa;
    if (TRAIN) tn_resolve_step(rs, sa);                            // dataset mode: this step's image and Philox counters
    const int S = sa.S, Lf = a.n.Lf;
    PipeX p;
    unsigned char* lds_bnd0 = lds + TX_BND_OFF(a.n, NW, true);
    unsigned char* lds_bnd = lds_bnd0 + lane * 4;                  // this lane's bound words (forward: TNB_H(l) = l, TNB_ENC)
    tx_prologue<NW, TRAIN>(p, lds, a.packed3, a.n, a.packed3, a.n.n_stage, lane, wave, TRAIN ? lds_bnd0 : nullptr);
    unsigned char* E = lds + TX_RING_OF(NW) + TX_CONST_BYTES(a.n) + wave * TX_ELDS_WAVE + lane * 16;      // this lane's input pieces
    TxProf pf;
#ifdef TN_STAMPS
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    if (a.f.stamps && blockIdx.x == 0 && wave == 0) pf.marks = a.f.stamps + (size_t)gridDim.x * NW * 8;      // behind the per-wave records
#ifdef TN_STAGE_STAMPS
    if (a.f.stamps && blockIdx.x == 0 && wave == 0) p.smarks = a.f.stamps + (size_t)gridDim.x * NW * 8 + 64;
#endif
#endif

    // Every wave of the workgroup runs the same number of network passes (the stage barriers are workgroup-wide): rays beyond
    // R are computed on a clamped index and stored nowhere (training: into the dump block).
    const int64_t R = a.f.R;
    const int64_t n_groups = (R + NW - 1) / NW;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t ray = g * NW + wave;
        const bool rvalid = ray < R;
        const int64_t rayc = rvalid ? ray : R - 1;
        float dn;
        { float ro_[3], rd_[3]; tn_fetch_ray(rs, rayc, ro_, rd_); dn = tn_norm3(rd_[0], rd_[1], rd_[2]); }
        float T_in = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, cd = 0.f, ca = 0.f;
        // March the ray 32 samples per pass; every second pass (or the last one) the 64 lanes composite a segment: lane l <- sample s0 + l.
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int sb = 0; sb < S; sb += 32) {
            {
                const int s = sb + j;
                const bool valid = rvalid && s < S;
                const int sc = s < S ? s : S - 1;
                const float z = tn_depth(sa, rayc, sc);
                // the ray is fetched again for every tile (behind a barrier the loop-invariant-code motion cannot cross): origin and
                // direction are needed HERE only, and six registers held through the layer walk are six the walk does not have
                int64_t rayt = rayc; asm volatile("" : "+s"(rayt));
                float ro_[3], rd_[3];
                tn_fetch_ray(rs, rayt, ro_, rd_);
                const float px = tn_point(ro_[0], rd_[0], z), py = tn_point(ro_[1], rd_[1], z), pz = tn_point(ro_[2], rd_[2], z);
                EncX Er;
                float encf[8 * TN16_KE];
                tx_encode(px, py, pz, Lf, h, encf);
                TxIn in;
                in.encmax = fmaxf(fmaxf(fabsf(px), fabsf(py)), fmaxf(fabsf(pz), 1.0f));      // |sin|, |cos| <= 1
                in.l1 = __builtin_fmaf(3.0f, in.encmax, (float)(6 * Lf));
                in.te = tx_scale_exp(in.encmax);
                if constexpr (TRAIN) tx_stash_input(a, h, lane, encf, rayc * S + sc, valid);
                tx_split_input(Er, tx_exp2i(in.te), [&](auto ac) TN_INLINE_LAMBDA { return encf[decltype(ac)::value]; });
                tx_store_input(E, Er);
                float res[4];
                tx_mlp_tile<HID, TRAIN, NW>(p, lds, a, h, lane, E, in, rayc * S + sc, valid, res, pf, lds_bnd);
                const bool upper = (sb & 32) != 0;                            // wave-uniform
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float mv = __shfl(res[i], lane & 31, 64);           // lanes 32..63 <- lanes 0..31
                    v[i] = upper ? (h ? mv : v[i]) : res[i];
                }
            }
            if ((sb & 32) == 0 && sb + 32 < S) continue;                      // wait for the upper half
            const int s0 = sb & ~63;
            const int s = s0 + lane;
            const bool ok = s < S && (s - s0) < ((sb & 32) ? 64 : 32);
            const int sc = s < S ? s : S - 1;
            const float z = tn_depth(sa, rayc, sc);
            const float zn = (s + 1 < S) ? tn_depth(sa, rayc, s + 1) : z;
            const CompTerms t = tn_comp_terms(ok ? v[3] : 0.0f, z, zn, s == S - 1, dn);       // volume.py:18-31
            const float om = ok ? t.om : 1.0f;
            const float incl = tn_wave_scan_mul(om, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float T = T_in * excl;
            const float w = ok ? t.alpha * T : 0.0f;                                          // volume.py:34
            cr += w * v[0]; cg += w * v[1]; cb += w * v[2]; cd += w * z; ca += w;             // volume.py:36-38
            T_in *= __shfl(incl, 63, 64);
            if (TRAIN && ok && rvalid) {
#pragma unroll
                for (int i = 0; i < 4; ++i) tn_stash_at(a.f.stash, a.f.L.stash_rows, rayc * S + s)[(a.f.L.out_row0 + i) * 32] = v[i];
            }
        }
        cr = tn_wave_sum(cr); cg = tn_wave_sum(cg); cb = tn_wave_sum(cb); cd = tn_wave_sum(cd); ca = tn_wave_sum(ca);
        if (lane == 0 && rvalid) {
            const float bg = a.f.white ? (1.0f - ca) : 0.0f;                                  // volume.py:42
            a.f.comp[3 * ray] = cr + bg; a.f.comp[3 * ray + 1] = cg + bg; a.f.comp[3 * ray + 2] = cb + bg;
            if (a.f.depth) a.f.depth[ray] = cd;
            if (a.f.acc) a.f.acc[ray] = ca;
            if (TRAIN && a.f.loss.ray_ws) tn_ray_loss(a.f.loss, rs, ray, cr + bg, cg + bg, cb + bg);      // train.py:122
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no DMA may still be writing this workgroup's LDS at exit
    if constexpr (TRAIN) {
        tx_bound_flush<NW>(lds_bnd0, a.f.stash + TN_BOUND_OFF(a.f.L, a.f.Mp), 0, lane, wave);
        if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned*>(a.f.stash + TN_BOUND_OFF(a.f.L, a.f.Mp))[TNB_TAG] = TN_TAG_X3;      // this pipe's stash
    }
#ifdef TN_STAMPS
    if (a.f.stamps && lane == 0) {
        unsigned long long* o = a.f.stamps + (blockIdx.x * NW + wave) * 8;
        o[0] = __builtin_amdgcn_s_memtime() - st_c0; o[1] = __builtin_amdgcn_s_memrealtime() - st_r0; o[2] = pf.walk; o[3] = pf.epi;
        o[4] = st_entry; o[5] = st_r0; o[6] = __builtin_amdgcn_s_memrealtime();
        if (pf.marks) pf.marks[63] = pf.n;
    }
#endif
}

// ------------------------------------------------------------------------------------------------ dgrad
// What autograd derives from volume.py:18-42 and nerf.py:34-40 (as mlp_bwd.hip), with the chain dH_{l-1} = W_l^T dZ_l on the
// fp16 matrix pipe: the backward record stream (heads^T, then the transposed hidden layers), dZ_l as two scaled pieces, the
// same k-step-major layer walk.  Reads the forward's sign bits and head outputs from the stash and writes every dZ_l (fp32)
// next to them, exactly where the weight-gradient kernel expects them.
struct BwdX3Args {
    BwdArgs b;                      // layout, stash, ray source, sampling, g_comp (+ stride) — as the fp32 kernel
    NetX3 n;
    const unsigned char* packed3;
};

// The sign words of one layer for this lane: fetched with an asm load so that the wait can be COUNTED (a compiler-placed
// vmcnt(0) would drain the weight ring).  The wait takes the load's own destination registers: nothing may read (or copy)
// them before it.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int NW> struct TxMask;
template <> struct TxMask<4> {
    u32x4 v;
    __device__ __forceinline__ void fetch(const uint32_t* q) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(q) : "memory"); }
    template <int N> __device__ __forceinline__ void wait() { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N) : "memory"); }
};
template <> struct TxMask<2> {
    u32x2 v;
    __device__ __forceinline__ void fetch(const uint32_t* q) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(q) : "memory"); }
    template <int N> __device__ __forceinline__ void wait() { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N) : "memory"); }
};

// dzh[4]: this lane's head gradients (r,g,b,sigma pre-activation) for sample m.
template <int HID, int NW>
__device__ __forceinline__ void tx_bwd_tile(PipeX& p, const unsigned char* lds, const BwdX3Args& a, const float (&dzh)[4], int64_t m, bool valid,
                                            int lane, unsigned char* lds_bnd) {
    constexpr int NT = HID / 32;
    const MlpLayout& L = a.b.L;
    const int h = lane >> 5, depth = a.n.depth;
    float* __restrict__ stash = a.b.stash;
    const int64_t Mp = a.b.Mp;
    const int64_t ms = valid ? m : Mp + (lane & 31);               // padding lanes use the dump block
    float* __restrict__ pl = tn_stash_at(stash, L.stash_rows, ms) + 4 * h * 32;
    if (h == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pl[(L.dzh_row0 + i) * 32] = dzh[i];
    }
    // this lane's sign words of layer depth-1; stepped down a layer with every fetch (as the row offset below: one live copy each)
    const uint32_t* __restrict__ mptr = reinterpret_cast<const uint32_t*>(stash + TN_STASH_BODY_FLOATS(L, Mp)) + (2 * ms + h) * (NT / 2)
                                        + (int64_t)(depth - 1) * (Mp + 32) * NT;
    ActX<HID> X;
    EncX Z;                                                        // the head gradient as the B operand of the heads^T k-step (slot (h=0, e<4) = row e)
    // scale of the head gradient: its largest magnitude; the bound on dH_{depth-1} = W_head^T dZ_head: max|W_head| ||dZ_head||_1
    const float zmax = fmaxf(fmaxf(fabsf(dzh[0]), fabsf(dzh[1])), fmaxf(fabsf(dzh[2]), fabsf(dzh[3])));
    int t_prev = tx_scale_exp(zmax);
    tx_bound_note(lds_bnd, TNB_DZH - TNB_DZ(0), zmax);
    {
        const float zs = tx_exp2i(t_prev);
        unsigned a0, b0, a1, b1;
        tx_split2(h ? 0.0f : dzh[0] * zs, h ? 0.0f : dzh[1] * zs, a0, b0);
        tx_split2(h ? 0.0f : dzh[2] * zs, h ? 0.0f : dzh[3] * zs, a1, b1);
        const u32x4 w1 = {a0, a1, 0u, 0u}, w2 = {b0, b1, 0u, 0u};
        Z.p1[0] = w1; Z.p2[0] = w2;
    }
    constexpr int NH = NT / 2, KH = HID / 16, NP = NH * 8, G2 = KH / 2 * NH, DPW = TX_STAGE / NW;
    f32x16 accA[TX_ACCN(HID)], accB[TX_ACCN(HID)];
    uint32_t mw[NT / 2];
    TxEpi es;
    TxScale sc{0.0f, 0.0f, {0.0f, 0.0f}, tx_konst()};
    TxMask<NT / 2> mk;                                             // asm load + counted wait (above)
    TxDst zrow = tx_dst_tile(stash, L.stash_rows, m, valid, h);
    // The product W_l^T dZ_l (l = depth: the heads) is about to enter its epilogues: the L1 norm of dZ_l is complete.
    auto scale_for = [&](int l, float l1_in) TN_INLINE_LAMBDA {
        const f32x4 mt = tx_meta<NW>(lds, a.n, l);
        const int t_out = tx_scale_exp(mt[1] * l1_in);
        tx_bound_note(lds_bnd, l - 1, mt[1] * l1_in);                          // bounds dH_{l-1}, hence dZ_{l-1} (local index of TNB_DZ(l - 1))
        sc.dsc = mt[0] * tx_exp2i(-t_prev); sc.osc = tx_exp2i(t_out); sc.l1 = f32x2{0.0f, 0.0f};
        t_prev = t_out;
    };
    // steps 0..3 = part V, 4..8 = part S    (mlpx3_core.hpp)
#ifdef TX_NO_ASM_ACC_READ
    constexpr bool AR = false;
#else
    constexpr bool AR = HID == 256 || TX_ASM_ACC_128;              // (mlpx3_core.hpp, tx_acc_get)
#endif
    auto epi_full = [&](auto halfc, auto arc, auto& acc, auto ic, auto kc) TN_INLINE_LAMBDA {
        constexpr int HALF = decltype(halfc)::value, I = decltype(ic)::value, K = decltype(kc)::value;
        if constexpr (K < TX_VSTEPS) tx_epi_bwd_value<HID, HALF, I, K, decltype(arc)::value>(acc, es, sc, mw);
        else tx_epi_split<HID, HALF, I, K, true, false>(X, es, sc, zrow);
    };
    using ArC = std::integral_constant<bool, AR>;
    // drains (right behind the MFMAs that produce the values: compiler-visible reads) ...
    auto epiA = [&](auto ic, auto kc) TN_INLINE_LAMBDA { epi_full(std::integral_constant<int, 0>{}, std::false_type{}, accA, ic, kc); };
    auto epiB = [&](auto ic, auto kc) TN_INLINE_LAMBDA { epi_full(std::integral_constant<int, 1>{}, std::false_type{}, accB, ic, kc); };
    // ... and the windows that ride on the other half's pass (accumulators read from their AGPRs where they are)
    auto epiA_w = [&](auto ic, auto kc) TN_INLINE_LAMBDA { epi_full(std::integral_constant<int, 0>{}, ArC{}, accA, ic, kc); };
    auto epiB_w = [&](auto ic, auto kc) TN_INLINE_LAMBDA { epi_full(std::integral_constant<int, 1>{}, ArC{}, accB, ic, kc); };
    constexpr int SPF = HID == 256 ? 1 : 4;
    constexpr int GB = HID == 256 ? TX_GB256 : G2, GA = HID == 256 ? TX_GA256 : G2;      // see tx_mlp_tile
    static_assert(tx_half_b_ok<HID, GB, SPF>(), "half B's epilogue window overruns the first read of its pieces");
    static_assert(tx_half_a_ok<HID, TX_WA, GA, SPF>(), "half A's epilogue window writes an activation slot pass B still reads, or overruns the pass");
    // dH_{depth-1} = W_head^T dZ_head: both halves; half A's epilogue has nothing to hide behind
    mk.fetch(mptr);
    tx_pass_headsT<HID, NW>(p, lds, Z, accA, accB);
    mk.template wait<DPW>();                           // one boundary (DPW DMAs) was issued behind the fetch
#pragma unroll
    for (int w = 0; w < NT / 2; ++w) mw[w] = mk.v[w];
    zrow.off += (uint32_t)L.dz_row0[depth - 1] * 128u;
    scale_for(depth, (fabsf(dzh[0]) + fabsf(dzh[1])) + (fabsf(dzh[2]) + fabsf(dzh[3])));
    tx_drain<NP, TX_NSTEP>(epiA);
    // layer l (dH_{l-1} = W_l^T dZ_l): dZ_l's half B epilogue behind pass A; the sign words of layer l-1 are fetched before pass A
    // and waited for (>= TX_LEAD boundaries later) at the start of pass B, where dZ_{l-1}'s half A epilogue begins
    for (int l = depth - 1; l >= 1; --l) {
        mptr -= (Mp + 32) * NT;
        mk.fetch(mptr);
        tx_pass<HID, 1, true, NW>(p, lds, X, nullptr, accA, tx_window<0, GB, NP, 0, TX_NSTEP, SPF>(epiB_w));
        auto wa = tx_window<TX_WA, GA, NP, 0, TX_NSTEP, SPF>(epiA_w);
        tx_pass<HID, 1, true, NW>(p, lds, X, nullptr, accB, [&](auto sc_) TN_INLINE_LAMBDA {
            if constexpr (decltype(sc_)::value == 0) {
                mk.template wait<DPW * TX_LEAD_OF(NW)>();
#pragma unroll
                for (int w = 0; w < NT / 2; ++w) mw[w] = mk.v[w];
                zrow.off += (uint32_t)(L.dz_row0[l - 1] - L.dz_row0[l]) * 128u;      // (mod 2^32)
                const float l1_own = sc.l1[0] + sc.l1[1];
                scale_for(l, l1_own + tx_partner(l1_own));
            }
            wa(sc_);
        });
    }
    tx_drain<NP, 5>(epiB);                                         // dZ_0, half B: to the stash only (steps 0 .. 4)
}

template <int HID>
__global__ TX_PLAIN_F32 __launch_bounds__(TxCfg<HID>::RAY * 64, 1) void k_dgradx3(BwdX3Args a) {
    constexpr int NW = TxCfg<HID>::RAY;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    if (!tn_stash_tag_is(a.b.stash, a.b.L, a.b.Mp, TN_TAG_X3)) return;      // not an x3 forward's stash: its sign words mean something else
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    RaySource rs = a.b.rs; SampleArgs sa = a.b.sa;
    tn_resolve_step(rs, sa);
    const int S = sa.S;
    const int nseg = (S + 63) / 64;
    PipeX p;
    unsigned char* lds_bnd0 = lds + TX_BND_OFF(a.n, NW, false);
    unsigned char* lds_bnd = lds_bnd0 + lane * 4;                  // this lane's bound words (dgrad: TNB_DZ(l) .. TNB_DZH, minus TNB_DZ(0))
    tx_prologue<NW>(p, lds, a.packed3, a.n, a.packed3 + (int64_t)a.n.n_rec * a.n.rec_frags * 1024, a.n.n_bw_stage, lane, wave, lds_bnd0);
    const int64_t R = a.b.R;
    const int64_t n_groups = (R + NW - 1) / NW;
    const int orow = a.b.L.out_row0 * 32; const int64_t SR = a.b.L.stash_rows;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t ray = g * NW + wave;
        const bool rvalid = ray < R;
        const int64_t rayc = rvalid ? ray : R - 1;
        float ro_[3], rd_[3];
        tn_fetch_ray(rs, rayc, ro_, rd_);
        const float dn = tn_norm3(rd_[0], rd_[1], rd_[2]);
        const int64_t gi = (int64_t)a.b.g_stride * rayc;
        const float gr = rvalid ? a.b.g_comp[gi] : 0.f, gg = rvalid ? a.b.g_comp[gi + 1] : 0.f, gb = rvalid ? a.b.g_comp[gi + 2] : 0.f;
        const float gbg = a.b.white ? (gr + gg + gb) : 0.0f;
        const int64_t mray = rayc * S;
        auto outv = [&](int i, int sc) TN_INLINE_LAMBDA { return tn_stash_at(a.b.stash, SR, mray + sc)[orow + 32 * i]; };

        float segprod = 1.0f;
        if (nseg > 1) {
            for (int sg_ = 0; sg_ < nseg; ++sg_) {
                const int s = sg_ * 64 + lane; const bool ok = s < S; const int sc = ok ? s : S - 1;
                const float z = tn_depth(sa, rayc, sc);
                const float zn = (s + 1 < S) ? tn_depth(sa, rayc, s + 1) : z;
                const CompTerms t = tn_comp_terms(ok ? outv(3, sc) : 0.f, z, zn, s == S - 1, dn);
                const float pr = tn_wave_prod(ok ? t.om : 1.0f);
                if (lane == sg_) segprod = pr;
            }
        }
        const float seg_incl = tn_wave_scan_mul(segprod, lane);
        float seg_T = __shfl_up(seg_incl, 1, 64);
        if (lane == 0) seg_T = 1.0f;
        float tail = 0.0f;
        for (int sg_ = nseg - 1; sg_ >= 0; --sg_) {
            const int s = sg_ * 64 + lane; const bool ok = s < S; const int sc = ok ? s : S - 1;
            const float c0 = outv(0, sc), c1 = outv(1, sc), c2 = outv(2, sc);
            const float sg = ok ? outv(3, sc) : 0.f;
            const float z = tn_depth(sa, rayc, sc);
            const float zn = (s + 1 < S) ? tn_depth(sa, rayc, s + 1) : z;
            const CompTerms t = tn_comp_terms(sg, z, zn, s == S - 1, dn);
            const float om = ok ? t.om : 1.0f;
            const float incl = tn_wave_scan_mul(om, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float T = __shfl(seg_T, sg_, 64) * excl;
            const float w = ok ? t.alpha * T : 0.f;
            const float dw = gr * c0 + gg * c1 + gb * c2 - gbg;
            const float v = ok ? w * dw : 0.f;
            const float suf = tn_wave_suffix_sum(v, lane);
            const float after = (suf - v) + tail;
            const float da = T * dw - after / om;
            float d4[4];
            d4[0] = ok ? (w * gr) * (c0 * (1.0f - c0)) : 0.f;                          // sigmoid backward of dL/dc = w g
            d4[1] = ok ? (w * gg) * (c1 * (1.0f - c1)) : 0.f;
            d4[2] = ok ? (w * gb) * (c2 * (1.0f - c2)) : 0.f;
            d4[3] = (ok && sg > 0.0f) ? (da * t.e) * t.delta : 0.f;                    // ReLU backward of dL/dsigma
            tail += __shfl(suf, 0, 64);
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                const int sb = sg_ * 64 + 32 * half;
                if (sb >= S) break;                                                    // wave-uniform
                float dzh[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) dzh[i] = __shfl(d4[i], 32 * half + (lane & 31), 64);
                const int st = sb + (lane & 31);
                const bool valid = rvalid && st < S;
                tx_bwd_tile<HID, NW>(p, lds, a, dzh, rayc * S + (st < S ? st : S - 1), valid, lane, lds_bnd);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tx_bound_flush<NW>(lds_bnd0, a.b.stash + TN_BOUND_OFF(a.b.L, a.b.Mp), TNB_DZ(0), lane, wave);
}

// ------------------------------------------------------------------------------------------------ MLP only
// TinyNeRF.forward / its backward on x[M, in_dim] given in memory (reference src/nerf.py:29-41; the per-function call style of
// src/train.py:117-118): the same tiles as the fused kernels with the network input READ instead of computed.  in_dim = 6L+3:
// the k-slot (step a = 8u + e, lane half h) of the record stream carries column 3 + 6 (a/3) + 3h + a%3 of x for a < 3L (the
// sin / cos of frequency a/3, coordinate a%3), columns h and 2 (h = 0) for the raw coordinates at a = 3L, 3L+1.
__device__ __forceinline__ void tx_load_input(const float* __restrict__ xrow, bool valid, int Lf, int h, float (&encf)[8 * TN16_KE]) {
    tn_static_for<8 * TN16_KE>([&](auto ac) TN_INLINE_LAMBDA {
        constexpr int a = decltype(ac)::value;
        int col = -1;
        if (a < 3 * Lf)           col = 3 + 6 * (a / 3) + 3 * h + a % 3;
        else if (a == 3 * Lf)     col = h;
        else if (a == 3 * Lf + 1) col = h ? -1 : 2;
        encf[a] = (valid && col >= 0) ? xrow[col] : 0.0f;
    });
}

template <int HID, bool TRAIN>
__global__ TX_PLAIN_F32 __launch_bounds__(TxCfg<HID>::RAY * 64, 1) void k_mlpx3_fwd(FwdX3Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    constexpr int NW = TxCfg<HID>::RAY;
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    PipeX p;
    unsigned char* lds_bnd0 = lds + TX_BND_OFF(a.n, NW, true);
    unsigned char* lds_bnd = lds_bnd0 + lane * 4;                  // this lane's bound words (forward: TNB_H(l) = l, TNB_ENC)
    tx_prologue<NW, TRAIN>(p, lds, a.packed3, a.n, a.packed3, a.n.n_stage, lane, wave, TRAIN ? lds_bnd0 : nullptr);
    unsigned char* E = lds + TX_RING_OF(NW) + TX_CONST_BYTES(a.n) + wave * TX_ELDS_WAVE + lane * 16;
    TxProf pf;
    const int64_t M = a.f.M, n_tiles = (M + 31) / 32, n_groups = (n_tiles + NW - 1) / NW;
    const int in_dim = a.n.in_dim, Lf = a.n.Lf;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {         // every wave of the workgroup runs every pass (stage barriers)
        const int64_t m = (g * NW + wave) * 32 + j;
        const bool valid = m < M;
        const int64_t mc = valid ? m : M - 1;
        EncX Er;
        float encf[8 * TN16_KE];
        tx_load_input(a.f.x + mc * in_dim, valid, Lf, h, encf);
        TxIn in;
        {   // this lane holds half of the row: largest magnitude and L1 norm over both lane halves
            float mx = 1.0f, l1 = 0.0f;
#pragma unroll
            for (int i = 0; i < 8 * TN16_KE; ++i) { mx = fmaxf(mx, fabsf(encf[i])); l1 += fabsf(encf[i]); }
            in.encmax = fmaxf(mx, tx_partner(mx)); in.l1 = l1 + tx_partner(l1); in.te = tx_scale_exp(in.encmax);
        }
        if constexpr (TRAIN) tx_stash_input(a, h, lane, encf, mc, valid);
        tx_split_input(Er, tx_exp2i(in.te), [&](auto ac) TN_INLINE_LAMBDA { return encf[decltype(ac)::value]; });
        tx_store_input(E, Er);
        float res[4];
        tx_mlp_tile<HID, TRAIN, NW, !TRAIN>(p, lds, a, h, lane, E, in, mc, valid, res, pf, lds_bnd);      // (training: the asm accumulator reads cost this kernel 9 spilled values)
        // the sample index is formed again behind the tile (from a lane id the compiler cannot match with the one above): kept alive
        // across the layer walk, the 64-bit index was what spilled to scratch in the 256-wide training kernel
        int lane2 = lane; asm volatile("" : "+v"(lane2));
        const int64_t m2 = (g * NW + wave) * 32 + (lane2 & 31);
        if (m2 < M && (lane2 >> 5) == 0) {
            a.f.rgb_out[3 * m2 + 0] = res[0]; a.f.rgb_out[3 * m2 + 1] = res[1]; a.f.rgb_out[3 * m2 + 2] = res[2];
            a.f.sigma_out[m2] = res[3];
            if (TRAIN) {
#pragma unroll
                for (int i = 0; i < 4; ++i) tn_stash_at(a.f.stash, a.f.L.stash_rows, m2)[(a.f.L.out_row0 + i) * 32] = res[i];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (TRAIN) {
        tx_bound_flush<NW>(lds_bnd0, a.f.stash + TN_BOUND_OFF(a.f.L, a.f.Mp), 0, lane, wave);
        if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned*>(a.f.stash + TN_BOUND_OFF(a.f.L, a.f.Mp))[TNB_TAG] = TN_TAG_X3;      // this pipe's stash
    }
}

template <int HID>
__global__ TX_PLAIN_F32 __launch_bounds__(TxCfg<HID>::TILE * 64, 1) void k_mlpx3_bwd(BwdX3Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    constexpr int NW = TxCfg<HID>::TILE;
    if (!tn_stash_tag_is(a.b.stash, a.b.L, a.b.Mp, TN_TAG_X3)) return;      // not an x3 forward's stash
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    PipeX p;
    unsigned char* lds_bnd0 = lds + TX_BND_OFF(a.n, NW, false);
    unsigned char* lds_bnd = lds_bnd0 + lane * 4;                  // this lane's bound words (dgrad: TNB_DZ(l) .. TNB_DZH, minus TNB_DZ(0))
    tx_prologue<NW>(p, lds, a.packed3, a.n, a.packed3 + (int64_t)a.n.n_rec * a.n.rec_frags * 1024, a.n.n_bw_stage, lane, wave, lds_bnd0);
    const int64_t M = a.b.M, n_tiles = (M + 31) / 32, n_groups = (n_tiles + NW - 1) / NW;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t m = (g * NW + wave) * 32 + (lane & 31);
        const bool valid = m < M;
        const int64_t mc = valid ? m : M - 1;
        float dzh[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float c = tn_stash_at(a.b.stash, a.b.L.stash_rows, mc)[(a.b.L.out_row0 + i) * 32];
            dzh[i] = valid ? a.b.d_rgb[3 * mc + i] * (c * (1.0f - c)) : 0.0f;            // sigmoid backward
        }
        const float sg = tn_stash_at(a.b.stash, a.b.L.stash_rows, mc)[(a.b.L.out_row0 + 3) * 32];
        dzh[3] = (valid && sg > 0.0f) ? a.b.d_sigma[mc] : 0.0f;                            // ReLU backward
        tx_bwd_tile<HID, NW>(p, lds, a, dzh, mc, valid, lane, lds_bnd);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tx_bound_flush<NW>(lds_bnd0, a.b.stash + TN_BOUND_OFF(a.b.L, a.b.Mp), TNB_DZ(0), lane, wave);
}

// ------------------------------------------------------------------------------------------------ sub-ray work units
// A ray is a wave in the kernels above, so a batch of fewer rays than the chip has waves (4 x 256 = 1024) leaves CUs idle while
// every wave still walks its ray's tiles one after the other: 512 rays x 64 samples occupy half the chip for two tile times.  With
// 32-sample TILES as the unit, 512 rays are 1024 tiles — every wave one tile.  What couples the tiles of a ray is the compositing
// (reference src/volume.py:30-36) and its backward; both are a few hundred flops per sample, so they move into a kernel of
// their own that runs between the tile kernels, one wave per ray, on the head outputs the forward tiles left in the stash:
//   k_tilex3_fwd  tile -> network -> stash (inputs, activations, sign bits, head outputs)
//   k_compx3      ray  -> compositing forward (exactly the arithmetic of k_renderx3: same segment scans, bit-identical colours),
//                         loss gradient, compositing backward (exactly k_dgradx3's) -> the head gradients into the stash
//   k_tilex3_bwd  tile -> head gradients from the stash -> the dgrad chain (tx_bwd_tile)
// The launchers take this route when rays < waves (tnx3_tile_units); results are bit-identical to the ray kernels'.
template <int HID>
__global__ TX_PLAIN_F32 __launch_bounds__(TxCfg<HID>::TILE * 64, 1) void k_tilex3_fwd(FwdX3Args a) {
    constexpr int NW = TxCfg<HID>::TILE;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    RaySource rs = a.f.rs; SampleArgs sa = a.f.sa;
    tn_resolve_step(rs, sa);
    const int S = sa.S, Lf = a.n.Lf, nt = (S + 31) / 32;
    PipeX p;
    unsigned char* lds_bnd0 = lds + TX_BND_OFF(a.n, NW, true);
    unsigned char* lds_bnd = lds_bnd0 + lane * 4;
    tx_prologue<NW, true>(p, lds, a.packed3, a.n, a.packed3, a.n.n_stage, lane, wave, lds_bnd0);
    unsigned char* E = lds + TX_RING_OF(NW) + TX_CONST_BYTES(a.n) + wave * TX_ELDS_WAVE + lane * 16;
    TxProf pf;
    const int64_t R = a.f.R, units = R * nt, n_groups = (units + NW - 1) / NW;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {         // every wave of the workgroup runs every pass (stage barriers)
        const int64_t u = g * NW + wave;
        const bool uvalid = u < units;
        const int64_t uc = uvalid ? u : units - 1;
        const int64_t ray = uc / nt;
        const int sb = (int)(uc - ray * nt) * 32;
        float ro_[3], rd_[3];
        tn_fetch_ray(rs, ray, ro_, rd_);
        const int s = sb + j;
        const bool valid = uvalid && s < S;
        const int sc = s < S ? s : S - 1;
        const float z = tn_depth(sa, ray, sc);
        const float px = tn_point(ro_[0], rd_[0], z), py = tn_point(ro_[1], rd_[1], z), pz = tn_point(ro_[2], rd_[2], z);
        EncX Er;
        float encf[8 * TN16_KE];
        tx_encode(px, py, pz, Lf, h, encf);
        TxIn in;
        in.encmax = fmaxf(fmaxf(fabsf(px), fabsf(py)), fmaxf(fabsf(pz), 1.0f));
        in.l1 = __builtin_fmaf(3.0f, in.encmax, (float)(6 * Lf));
        in.te = tx_scale_exp(in.encmax);
        const int64_t m = ray * S + sc;
        tx_stash_input(a, h, lane, encf, m, valid);
        tx_split_input(Er, tx_exp2i(in.te), [&](auto ac) TN_INLINE_LAMBDA { return encf[decltype(ac)::value]; });
        tx_store_input(E, Er);
        float res[4];
        tx_mlp_tile<HID, true, NW>(p, lds, a, h, lane, E, in, m, valid, res, pf, lds_bnd);
        int lane2 = lane; asm volatile("" : "+v"(lane2));             // (as k_mlpx3_fwd: the index is formed again behind the tile)
        const int s2 = sb + (lane2 & 31);
        if (uvalid && s2 < S && (lane2 >> 5) == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) tn_stash_at(a.f.stash, a.f.L.stash_rows, ray * S + s2)[(a.f.L.out_row0 + i) * 32] = res[i];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tx_bound_flush<NW>(lds_bnd0, a.f.stash + TN_BOUND_OFF(a.f.L, a.f.Mp), 0, lane, wave);
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned*>(a.f.stash + TN_BOUND_OFF(a.f.L, a.f.Mp))[TNB_TAG] = TN_TAG_X3;
}

// One wave per ray.  FWD: compositing forward from the stash's head outputs -> comp (+ depth, acc), the loss gradient if a target is
// given.  BWD: the compositing backward for g = dL/dcomp (the loss gradient just formed, or g_comp) -> head gradients into the stash.
struct CompX3Args {
    MlpLayout L; float* stash; int64_t Mp;
    RaySource rs; SampleArgs sa; int64_t R; int32_t white;
    float* comp; float* depth; float* acc; LossArgs loss;          // FWD
    const float* g_comp; int32_t g_stride;                         // BWD without FWD
};
template <bool FWD, bool BWD>
__global__ __launch_bounds__(256) void k_compx3(CompX3Args a) {
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
    if (ray >= a.R) return;
    RaySource rs = a.rs; SampleArgs sa = a.sa;
    tn_resolve_step(rs, sa);
    const int S = sa.S, nseg = (S + 63) / 64;
    float ro_[3], rd_[3];
    tn_fetch_ray(rs, ray, ro_, rd_);
    const float dn = tn_norm3(rd_[0], rd_[1], rd_[2]);
    const int orow = a.L.out_row0 * 32; const int64_t SR = a.L.stash_rows;
    const int64_t mray = ray * S;
    auto outv = [&](int i, int sc) TN_INLINE_LAMBDA { return tn_stash_at(a.stash, SR, mray + sc)[orow + 32 * i]; };
    float gr = 0.f, gg = 0.f, gb = 0.f;
    if constexpr (FWD) {       // k_renderx3's compositing, segment by segment                       reference src/volume.py:18-42
        float T_in = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, cd = 0.f, ca = 0.f;
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int s = s0 + lane;
            const bool ok = s < S;
            const int sc = ok ? s : S - 1;
            const float v0 = outv(0, sc), v1 = outv(1, sc), v2 = outv(2, sc), v3 = outv(3, sc);
            const float z = tn_depth(sa, ray, sc);
            const float zn = (s + 1 < S) ? tn_depth(sa, ray, s + 1) : z;
            const CompTerms t = tn_comp_terms(ok ? v3 : 0.0f, z, zn, s == S - 1, dn);
            const float om = ok ? t.om : 1.0f;
            const float incl = tn_wave_scan_mul(om, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float T = T_in * excl;
            const float w = ok ? t.alpha * T : 0.0f;
            cr += w * v0; cg += w * v1; cb += w * v2; cd += w * z; ca += w;
            T_in *= __shfl(incl, 63, 64);
        }
        cr = tn_wave_sum(cr); cg = tn_wave_sum(cg); cb = tn_wave_sum(cb); cd = tn_wave_sum(cd); ca = tn_wave_sum(ca);
        const float bg = a.white ? (1.0f - ca) : 0.0f;
        if (lane == 0) {
            a.comp[3 * ray] = cr + bg; a.comp[3 * ray + 1] = cg + bg; a.comp[3 * ray + 2] = cb + bg;
            if (a.depth) a.depth[ray] = cd;
            if (a.acc) a.acc[ray] = ca;
            if (a.loss.ray_ws) tn_ray_loss(a.loss, rs, ray, cr + bg, cg + bg, cb + bg);      // train.py:122
        }
        if constexpr (BWD) {   // every lane needs the gradient lane 0 has just stored: the same three operations, redone in registers
            int64_t row = ray;
            if (rs.c2w) { const int64_t pp = tn_ray_pixel(rs, ray); row = rs.step ? (int64_t)rs.image * rs.H * rs.W + pp : pp; }
            else if (a.loss.target_index) row = a.loss.target_index[ray];
            gr = (2.0f * ((cr + bg) - a.loss.target[3 * row])) * a.loss.inv_denom;
            gg = (2.0f * ((cg + bg) - a.loss.target[3 * row + 1])) * a.loss.inv_denom;
            gb = (2.0f * ((cb + bg) - a.loss.target[3 * row + 2])) * a.loss.inv_denom;
        }
    } else {
        const int64_t gi = (int64_t)a.g_stride * ray;
        gr = a.g_comp[gi]; gg = a.g_comp[gi + 1]; gb = a.g_comp[gi + 2];
    }
    if constexpr (BWD) {       // k_dgradx3's compositing backward; the head gradients go to the stash instead of into the tile walk
        const float gbg = a.white ? (gr + gg + gb) : 0.0f;
        float segprod = 1.0f;
        if (nseg > 1) {
            for (int sg_ = 0; sg_ < nseg; ++sg_) {
                const int s = sg_ * 64 + lane; const bool ok = s < S; const int sc = ok ? s : S - 1;
                const float z = tn_depth(sa, ray, sc);
                const float zn = (s + 1 < S) ? tn_depth(sa, ray, s + 1) : z;
                const CompTerms t = tn_comp_terms(ok ? outv(3, sc) : 0.f, z, zn, s == S - 1, dn);
                const float pr = tn_wave_prod(ok ? t.om : 1.0f);
                if (lane == sg_) segprod = pr;
            }
        }
        const float seg_incl = tn_wave_scan_mul(segprod, lane);
        float seg_T = __shfl_up(seg_incl, 1, 64);
        if (lane == 0) seg_T = 1.0f;
        float tail = 0.0f;
        const int zrow = a.L.dzh_row0 * 32;
        for (int sg_ = nseg - 1; sg_ >= 0; --sg_) {
            const int s = sg_ * 64 + lane; const bool ok = s < S; const int sc = ok ? s : S - 1;
            const float c0 = outv(0, sc), c1 = outv(1, sc), c2 = outv(2, sc);
            const float sg = ok ? outv(3, sc) : 0.f;
            const float z = tn_depth(sa, ray, sc);
            const float zn = (s + 1 < S) ? tn_depth(sa, ray, s + 1) : z;
            const CompTerms t = tn_comp_terms(sg, z, zn, s == S - 1, dn);
            const float om = ok ? t.om : 1.0f;
            const float incl = tn_wave_scan_mul(om, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float T = __shfl(seg_T, sg_, 64) * excl;
            const float w = ok ? t.alpha * T : 0.f;
            const float dw = gr * c0 + gg * c1 + gb * c2 - gbg;
            const float v = ok ? w * dw : 0.f;
            const float suf = tn_wave_suffix_sum(v, lane);
            const float after = (suf - v) + tail;
            const float da = T * dw - after / om;
            float d4[4];
            d4[0] = ok ? (w * gr) * (c0 * (1.0f - c0)) : 0.f;
            d4[1] = ok ? (w * gg) * (c1 * (1.0f - c1)) : 0.f;
            d4[2] = ok ? (w * gb) * (c2 * (1.0f - c2)) : 0.f;
            d4[3] = (ok && sg > 0.0f) ? (da * t.e) * t.delta : 0.f;
            tail += __shfl(suf, 0, 64);
            if (ok) {
                float* q = tn_stash_at(a.stash, SR, mray + s) + zrow;
#pragma unroll
                for (int i = 0; i < 4; ++i) q[32 * i] = d4[i];
            }
        }
    }
}

template <int HID>
__global__ TX_PLAIN_F32 __launch_bounds__(TxCfg<HID>::TILE * 64, 1) void k_tilex3_bwd(BwdX3Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    constexpr int NW = TxCfg<HID>::TILE;
    if (!tn_stash_tag_is(a.b.stash, a.b.L, a.b.Mp, TN_TAG_X3)) return;
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    PipeX p;
    unsigned char* lds_bnd0 = lds + TX_BND_OFF(a.n, NW, false);
    unsigned char* lds_bnd = lds_bnd0 + lane * 4;
    tx_prologue<NW>(p, lds, a.packed3, a.n, a.packed3 + (int64_t)a.n.n_rec * a.n.rec_frags * 1024, a.n.n_bw_stage, lane, wave, lds_bnd0);
    const int S = a.b.sa.S, nt = (S + 31) / 32;
    const int64_t R = a.b.R, units = R * nt, n_groups = (units + NW - 1) / NW;
    const int zrow = a.b.L.dzh_row0 * 32;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t u = g * NW + wave;
        const bool uvalid = u < units;
        const int64_t uc = uvalid ? u : units - 1;
        const int64_t ray = uc / nt;
        const int s = (int)(uc - ray * nt) * 32 + (lane & 31);
        const bool valid = uvalid && s < S;
        const int64_t mc = ray * S + (s < S ? s : S - 1);
        const float* q = tn_stash_at(a.b.stash, a.b.L.stash_rows, mc) + zrow;
        float dzh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dzh[i] = valid ? q[32 * i] : 0.0f;
        tx_bwd_tile<HID, NW>(p, lds, a, dzh, mc, valid, lane, lds_bnd);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tx_bound_flush<NW>(lds_bnd0, a.b.stash + TN_BOUND_OFF(a.b.L, a.b.Mp), TNB_DZ(0), lane, wave);
}

#ifdef TX_SKIP256      // compile-time experiments on the 128-wide kernels only (never in a library build: -DTN_DIAG)
#ifndef TN_DIAG
#error "TX_SKIP256 is a diagnostic knob"
#endif
#define TX_IF256(...)
#else
#define TX_IF256(...) __VA_ARGS__
#endif
// rays < waves: the tile route (above).  TNERF_X3_UNITS=rays / tiles overrides (A/B runs, the bitwise test of the two routes).
// 128-wide networks train through the tile kernels whenever a ray has more than one tile: those run two waves per SIMD (TxCfg), which the
// ray kernels' compositing state does not leave the registers for.
static bool tnx3_tile_units(int64_t R, int32_t S, int64_t n_waves, int hidden) {
    const char* e = getenv("TNERF_X3_UNITS");                        // (read per launch: a test flips it between two calls)
    const int force = !e ? 0 : (e[0] == 't' ? 1 : (e[0] == 'r' ? -1 : 0));
    if (force) return force > 0 && S > 0;
    return S > 32 && (R < n_waves || (hidden == 128 && TxCfg<128>::TILE > TxCfg<128>::RAY));
}
static int tnx3_waves(int hidden, int kind) {      // kind 0: ray training kernels + module forward, 1: tile kernels + module backward, 2: inference render
    if (hidden == 256) return kind == 0 ? TxCfg<256>::RAY : (kind == 1 ? TxCfg<256>::TILE : TxCfg<256>::RENDER);
    return kind == 0 ? TxCfg<128>::RAY : (kind == 1 ? TxCfg<128>::TILE : TxCfg<128>::RENDER);
}

// units: rays (fused) or 32-sample tiles (mlp_only) — one per wave and pass
// heads_done: the forward's compositing kernel has already written the head gradients (k_compx3<true, true>, tnx3_launch_fwd)
int tnx3_launch_dgrad(const BwdX3Args& a, bool mlp_only, hipStream_t stream, const char* who, bool heads_done = false) {
    const int dev = tn_stream_device(stream), n_cu = tn_device_cus(dev);
    const int nw = tnx3_waves(a.n.hidden, mlp_only ? 1 : 0);        // k_mlpx3_bwd | k_dgradx3
    const int64_t units = mlp_only ? (a.b.M + 31) / 32 : a.b.R, groups = (units + nw - 1) / nw;
    const dim3 grid((unsigned)(groups < n_cu ? groups : n_cu)), block(nw * 64);
    const size_t lds_bytes = TX_LDS_BYTES(a.n, nw, false);
    if (!mlp_only && tnx3_tile_units(a.b.R, a.b.sa.S, (int64_t)nw * n_cu, a.n.hidden)) {       // tiles as the unit (k_tilex3_bwd)
        const int tw = tnx3_waves(a.n.hidden, 1);
        const dim3 tblock(tw * 64);
        const size_t tlds = TX_LDS_BYTES(a.n, tw, false);
        if (!heads_done) {
            CompX3Args c{};
            c.L = a.b.L; c.stash = a.b.stash; c.Mp = a.b.Mp; c.rs = a.b.rs; c.sa = a.b.sa; c.R = a.b.R; c.white = a.b.white;
            c.g_comp = a.b.g_comp; c.g_stride = a.b.g_stride;
            hipLaunchKernelGGL((k_compx3<false, true>), dim3((unsigned)((a.b.R + 3) / 4)), dim3(256), 0, stream, c);
            TN_HIP_CHECK_LAUNCH(who);
        }
        const int64_t tiles = a.b.R * ((a.b.sa.S + 31) / 32), tg = (tiles + tw - 1) / tw;
        const dim3 tgrid((unsigned)(tg < n_cu ? tg : n_cu));
#define TX_TCASE(H_)                                                                                                          \
        if (a.n.hidden == H_) {                                                                                               \
            static std::atomic<uint32_t> seen_[TN_MAX_DEVICES];                                                               \
            if (int rc_ = tn_grant_dyn_lds(reinterpret_cast<const void*>(&k_tilex3_bwd<H_>), tlds, dev, seen_, who)) return rc_; \
            hipLaunchKernelGGL((k_tilex3_bwd<H_>), tgrid, tblock, tlds, stream, a);                                           \
            TN_HIP_CHECK_LAUNCH(who);                                                                                         \
            return TNERF_OK;                                                                                                  \
        }
        TX_IF256(TX_TCASE(256)) TX_TCASE(128)
#undef TX_TCASE
    }
#define TX_CASE(H_, K_, M_)                                                                                                  \
    if (a.n.hidden == H_ && mlp_only == M_) {                                                                                 \
        static std::atomic<uint32_t> seen_[TN_MAX_DEVICES];                                                                   \
        if (int rc_ = tn_grant_dyn_lds(reinterpret_cast<const void*>(&K_<H_>), lds_bytes, dev, seen_, who)) return rc_;       \
        hipLaunchKernelGGL((K_<H_>), grid, block, lds_bytes, stream, a);                                                      \
        TN_HIP_CHECK_LAUNCH(who);                                                                                             \
        return TNERF_OK;                                                                                                      \
    }
    if (heads_done) { tn_set_error("%s: the forward took the tile route, the backward does not (TNERF_X3_UNITS changed in between?)", who); return TNERF_EINVAL; }
    TX_IF256(TX_CASE(256, k_dgradx3, false)) TX_CASE(128, k_dgradx3, false) TX_IF256(TX_CASE(256, k_mlpx3_bwd, true)) TX_CASE(128, k_mlpx3_bwd, true)
#undef TX_CASE
    tn_set_error("%s: no x3 kernel for hidden=%d", who, a.n.hidden);
    return TNERF_EUNSUPPORTED;
}

// dgrad of a train step on the x3 kernel (train_api.hip calls this instead of the fp32-MFMA dgrad when packed3 is given).
int tnx3_train_dgrad(const char* who, const BwdArgs& b, const tnerf_mlp_desc* d, const void* packed3, hipStream_t stream, bool heads_done) {
    BwdX3Args a{};
    int rc = tn_build_netx3(d, &a.n); if (rc) return rc;
    a.b = b; a.packed3 = static_cast<const unsigned char*>(packed3);
    return tnx3_launch_dgrad(a, false, stream, who, heads_done);
}

// Clears a few words.  A kernel, NOT hipMemsetAsync: these launches are captured into the train step's hipGraph, and on ROCm 7.0 a
// captured memset node writes a stale 16-byte pattern instead of its value from the second replay on (measured: tests/probes/
// bound_words_probe.py; eager launches and the first replay are fine) — the bound words then carry garbage, NaN bit patterns included.
__global__ void k_zero_words(unsigned* __restrict__ w, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) w[i] = 0u;
}

// heads_done (a train step: the backward follows at once, on the loss gradient this forward forms): where the tile route is taken, ONE
// compositing kernel does the forward, the loss gradient and the compositing backward (k_compx3<true, true>) and *heads_done is set —
// tnx3_launch_dgrad then goes straight to the tile kernel.  One launch (~7 us) and one pass over the head outputs less per step.
int tnx3_launch_fwd(const FwdX3Args& a, bool train, hipStream_t stream, const char* who, bool mlp_only = false, bool* heads_done = nullptr) {
    const int dev = tn_stream_device(stream), n_cu = tn_device_cus(dev);
    if (train) {       // the stash's magnitude bounds start from zero with every training forward (the dgrad kernel adds its own)
        hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, reinterpret_cast<unsigned*>(a.f.stash + TN_BOUND_OFF(a.f.L, a.f.Mp)), TN_BOUND_FLOATS);
        TN_HIP_CHECK_LAUNCH(who);
    }
    const int nw = tnx3_waves(a.n.hidden, (mlp_only || train) ? 0 : 2);      // k_mlpx3_fwd, k_renderx3<.., true> | k_renderx3<.., false>
    const int64_t units = mlp_only ? (a.f.M + 31) / 32 : a.f.R, groups = (units + nw - 1) / nw;
    const dim3 grid((unsigned)(groups < n_cu ? groups : n_cu)), block(nw * 64);
    const size_t lds_bytes = TX_LDS_BYTES(a.n, nw, true);
    if (train && !mlp_only && tnx3_tile_units(a.f.R, a.f.sa.S, (int64_t)nw * n_cu, a.n.hidden)) {      // tiles as the unit, then the rays' compositing
        const int tw = tnx3_waves(a.n.hidden, 1);
        const dim3 tblock(tw * 64);
        const size_t tlds = TX_LDS_BYTES(a.n, tw, true);
        const int64_t tiles = a.f.R * ((a.f.sa.S + 31) / 32), tg = (tiles + tw - 1) / tw;
        const dim3 tgrid((unsigned)(tg < n_cu ? tg : n_cu));
        int launched = 0;
#define TX_TCASE(H_)                                                                                                          \
        if (a.n.hidden == H_) {                                                                                               \
            static std::atomic<uint32_t> seen_[TN_MAX_DEVICES];                                                               \
            if (int rc_ = tn_grant_dyn_lds(reinterpret_cast<const void*>(&k_tilex3_fwd<H_>), tlds, dev, seen_, who)) return rc_; \
            hipLaunchKernelGGL((k_tilex3_fwd<H_>), tgrid, tblock, tlds, stream, a);                                           \
            TN_HIP_CHECK_LAUNCH(who);                                                                                         \
            launched = 1;                                                                                                     \
        }
        TX_IF256(TX_TCASE(256)) TX_TCASE(128)
#undef TX_TCASE
        if (!launched) { tn_set_error("%s: no x3 kernel for hidden=%d", who, a.n.hidden); return TNERF_EUNSUPPORTED; }
        CompX3Args c{};
        c.L = a.f.L; c.stash = a.f.stash; c.Mp = a.f.Mp; c.rs = a.f.rs; c.sa = a.f.sa; c.R = a.f.R; c.white = a.f.white;
        c.comp = a.f.comp; c.depth = a.f.depth; c.acc = a.f.acc; c.loss = a.f.loss;
        if (heads_done && c.loss.ray_ws && c.loss.target) {
            hipLaunchKernelGGL((k_compx3<true, true>), dim3((unsigned)((a.f.R + 3) / 4)), dim3(256), 0, stream, c);
            *heads_done = true;
        } else {
            hipLaunchKernelGGL((k_compx3<true, false>), dim3((unsigned)((a.f.R + 3) / 4)), dim3(256), 0, stream, c);
        }
        TN_HIP_CHECK_LAUNCH(who);
        return TNERF_OK;
    }
#define TX_CASE(H_, T_)                                                                                                      \
    if (a.n.hidden == H_ && train == T_) {                                                                                    \
        static std::atomic<uint32_t> seen_[TN_MAX_DEVICES];                                                                   \
        if (int rc_ = tn_grant_dyn_lds(reinterpret_cast<const void*>(&k_renderx3<H_, T_>), lds_bytes, dev, seen_, who)) return rc_; \
        hipLaunchKernelGGL((k_renderx3<H_, T_>), grid, block, lds_bytes, stream, a);                                          \
        TN_HIP_CHECK_LAUNCH(who);                                                                                             \
        return TNERF_OK;                                                                                                      \
    }
    if (!mlp_only) { TX_IF256(TX_CASE(256, false) TX_CASE(256, true)) TX_CASE(128, false) TX_CASE(128, true) }
#undef TX_CASE
#define TX_CASE(H_, T_)                                                                                                      \
    if (a.n.hidden == H_ && train == T_) {                                                                                    \
        static std::atomic<uint32_t> seen_[TN_MAX_DEVICES];                                                                   \
        if (int rc_ = tn_grant_dyn_lds(reinterpret_cast<const void*>(&k_mlpx3_fwd<H_, T_>), lds_bytes, dev, seen_, who)) return rc_; \
        hipLaunchKernelGGL((k_mlpx3_fwd<H_, T_>), grid, block, lds_bytes, stream, a);                                         \
        TN_HIP_CHECK_LAUNCH(who);                                                                                             \
        return TNERF_OK;                                                                                                      \
    }
    if (mlp_only) { TX_IF256(TX_CASE(256, false) TX_CASE(256, true)) TX_CASE(128, false) TX_CASE(128, true) }
#undef TX_CASE
    tn_set_error("%s: no x3 kernel for hidden=%d", who, a.n.hidden);
    return TNERF_EUNSUPPORTED;
}

// TinyNeRF.forward on the x3 chain (in_dim = 6L+3; tnerf_mlp_fwd otherwise).  Same outputs and stash as tnerf_mlp_fwd.
extern "C" int tnerf_mlp_fwd_x3(const tnerf_mlp_desc* d, const void* packed3, const float* x, int64_t M, float* rgb, float* sigma,
                                float* stash, int64_t Mp, tnerf_stream_t stream) {
    const char* who = "tnerf_mlp_fwd_x3";
    FwdX3Args a{};
    int rc = tn_build_netx3(d, &a.n); if (rc) return rc;
    rc = tn_build_layout(d, &a.f.L); if (rc) return rc;
    if (M == 0) return TNERF_OK;
    if (M < 0 || !packed3 || !x || !rgb || !sigma || (stash && Mp < M)) {
        tn_set_error("%s: M=%lld packed3=%p x=%p rgb=%p sigma=%p Mp=%lld", who, (long long)M, packed3, (const void*)x, (void*)rgb, (void*)sigma, (long long)Mp);
        return TNERF_EINVAL;
    }
    a.packed3 = static_cast<const unsigned char*>(packed3);
    a.f.x = x; a.f.M = M; a.f.rgb_out = rgb; a.f.sigma_out = sigma; a.f.stash = stash; a.f.Mp = Mp;
    return tnx3_launch_fwd(a, stash != nullptr, (hipStream_t)stream, who, true);
}

// dgrad of tnerf_mlp_bwd on the x3 chain (train_api.hip adds the weight-gradient kernel and the slab reduction).
int tnx3_mlp_dgrad(const char* who, const BwdArgs& b, const tnerf_mlp_desc* d, const void* packed3, hipStream_t stream) {
    BwdX3Args a{};
    int rc = tn_build_netx3(d, &a.n); if (rc) return rc;
    a.b = b; a.packed3 = static_cast<const unsigned char*>(packed3);
    return tnx3_launch_dgrad(a, true, stream, who);
}

// ----------------------------------------------------------------------------------- entry points
static int x3_args(const char* who, FwdX3Args& a, const tnerf_mlp_desc* d, const void* packed3, const RaySource& rs, int64_t R, int32_t S,
                   const float* ztab, int32_t randomized, const float* t_rand, uint64_t seed, uint64_t offset, int32_t white) {
    int rc = tn_build_netx3(d, &a.n); if (rc) return rc;
    // tn_fused_args validates the common arguments; the fp32 packed pointer is not used by these kernels
    rc = tn_fused_args(who, a.f, d, reinterpret_cast<const float*>(packed3), rs, R, S, ztab, randomized, t_rand, seed, offset, white);
    if (rc) return rc;
    a.packed3 = static_cast<const unsigned char*>(packed3);
    return TNERF_OK;
}

static int renderx3_impl(const char* who, const tnerf_mlp_desc* d, const void* packed3, const RaySource& rs, int64_t R, int32_t S,
                         const float* ztab, int32_t randomized, const float* t_rand, uint64_t seed, uint64_t offset, int32_t white,
                         float* comp, float* depth, float* acc, tnerf_stream_t stream) {
    FwdX3Args a{};
    int rc = x3_args(who, a, d, packed3, rs, R, S, ztab, randomized, t_rand, seed, offset, white); if (rc) return rc;
    if (R == 0) return TNERF_OK;
    if (!comp) { tn_set_error("%s: comp_rgb is NULL", who); return TNERF_EINVAL; }
    a.f.comp = comp; a.f.depth = depth; a.f.acc = acc;
    return tnx3_launch_fwd(a, false, (hipStream_t)stream, who);
}

extern "C" int tnerf_render_fused_x3(const tnerf_mlp_desc* d, const void* packed3, const float* rays_o, const float* rays_d,
                                     int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                                     uint64_t seed, uint64_t offset, int32_t white, float* comp, float* depth, float* acc,
                                     tnerf_stream_t stream) {
    return renderx3_impl("tnerf_render_fused_x3", d, packed3, tn_table_source(rays_o, rays_d), R, S, ztab, randomized, t_rand, seed,
                         offset, white, comp, depth, acc, stream);
}

extern "C" int tnerf_render_fused_cam_x3(const tnerf_mlp_desc* d, const void* packed3, const tnerf_camera* cam, int64_t R, int32_t S,
                                         const float* ztab, int32_t randomized, const float* t_rand, uint64_t seed, uint64_t offset,
                                         int32_t white, float* comp, float* depth, float* acc, tnerf_stream_t stream) {
    RaySource rs;
    int rc = tn_camera_source("tnerf_render_fused_cam_x3", cam, R, &rs); if (rc) return rc;
    return renderx3_impl("tnerf_render_fused_cam_x3", d, packed3, rs, R, S, ztab, randomized, t_rand, seed, offset, white, comp, depth,
                         acc, stream);
}

#ifdef TN_STAMPS
extern "C" int tnerf_debug_renderx3_stamps(const tnerf_mlp_desc* d, const void* packed3, const float* rays_o, const float* rays_d, int64_t R, int32_t S,
                                           const float* ztab, float* comp, float* stash, int64_t Mp, long long* stamps, tnerf_stream_t stream) {
    FwdX3Args a{};
    int rc = x3_args("tnerf_debug_renderx3_stamps", a, d, packed3, tn_table_source(rays_o, rays_d), R, S, ztab, 0, nullptr, 0, 0, 1);
    if (rc) return rc;
    a.f.comp = comp; a.f.stash = stash; a.f.Mp = Mp; a.f.stamps = reinterpret_cast<unsigned long long*>(stamps);
    return tnx3_launch_fwd(a, stash != nullptr, (hipStream_t)stream, "tnerf_debug_renderx3_stamps");
}
#endif

// Training forward into the fp32 stash (tn_step32_core and tnerf_train_fwd_fused_x3 call this instead of the fp32-MFMA forward).
int tnx3_train_fwd(const char* who, const tnerf_mlp_desc* d, const void* packed3, const RaySource& rs, const TnStepRef& sr,
                   const LossArgs& loss, int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                   uint64_t seed, uint64_t offset, int32_t white, float* comp, float* stash, int64_t Mp, hipStream_t stream, bool* heads_done) {
    FwdX3Args a{};
    int rc = x3_args(who, a, d, packed3, rs, R, S, ztab, randomized, t_rand, seed, offset, white); if (rc) return rc;
    if (R < 1 || !comp || !stash || Mp < R * S) { tn_set_error("%s: comp=%p stash=%p Mp=%lld < R*S=%lld", who, (void*)comp, (void*)stash, (long long)Mp, (long long)(R * S)); return TNERF_EINVAL; }
    a.f.comp = comp; a.f.stash = stash; a.f.Mp = Mp; a.f.loss = loss;
    a.f.sa.step = sr.step; a.f.sa.per_step = sr.per_step;
    return tnx3_launch_fwd(a, true, stream, who, false, heads_done);
}

extern "C" int tnerf_train_fwd_fused_x3(const tnerf_mlp_desc* d, const void* packed3, const float* rays_o, const float* rays_d,
                                        int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                                        uint64_t seed, uint64_t offset, int32_t white, float* comp, float* stash, int64_t Mp,
                                        tnerf_stream_t stream) {
    return tnx3_train_fwd("tnerf_train_fwd_fused_x3", d, packed3, tn_table_source(rays_o, rays_d), TnStepRef{}, LossArgs{}, R, S, ztab,
                          randomized, t_rand, seed, offset, white, comp, stash, Mp, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------------- packing
// max|W| and max|b| of every layer (index depth = the heads) through the pack table (forward stream, piece-0 fragments: every
// weight of the layer exactly once), TX_SCAN_NB workgroups per layer, into the running maxima of the scale records.  Maxima are
// order-independent: deterministic.
#define TX_SCAN_NB 16
__global__ __launch_bounds__(256) void k_x3stats_scan(const float* __restrict__ params, const int32_t* __restrict__ table, NetX3 n,
                                                      float* __restrict__ meta) {
    const int l = blockIdx.x / TX_SCAN_NB, part = blockIdx.x % TX_SCAN_NB;
    const int64_t fe = (int64_t)n.rec_frags * 512;
    const int64_t e0 = n.fw_rec0[l] * fe, e1 = n.fw_rec0[l + 1] * fe, n_w = (int64_t)(n.n_rec + n.n_bw_rec) * fe;
    float mw = 0.0f, mb = 0.0f;
    for (int64_t e = e0 + part * 256 + threadIdx.x; e < e1; e += TX_SCAN_NB * 256) {
        if (((e >> 9) % TX_NP) != 0) continue;
        const int32_t s = table[e];
        if (s >= 0) mw = fmaxf(mw, fabsf(params[s]));
    }
    if (part == 0) {
        const int nb = l < n.depth ? n.hidden : 4, b0 = l < n.depth ? l * n.hidden : n.depth * n.hidden;
        for (int j = threadIdx.x; j < nb; j += 256) {
            const int32_t s = table[n_w + b0 + j];
            if (s >= 0) mb = fmaxf(mb, fabsf(params[s]));
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { mw = fmaxf(mw, __shfl_xor(mw, o, 64)); mb = fmaxf(mb, __shfl_xor(mb, o, 64)); }
    if ((threadIdx.x & 63) == 0) {
        unsigned* a = reinterpret_cast<unsigned*>(meta + l * TX_META + 4);
        atomicMax(a, __float_as_uint(mw));
        if (part == 0) atomicMax(a + 1, __float_as_uint(mb));
    }
}
// Running maxima -> scale records (one thread per layer), and the maxima cleared for the next round.
//   post = 0 (a full pack follows): {1 / 2^s, max|W|, max|b|, 2^s} with max|W| 2^s in [2^11, 2^12)
//   post = 1 (the finishing kernel has just re-scattered every weight with the scale the record's [3] named, accumulating the
//            maxima of the updated parameters): [0] <- 1 / [3], then [1], [2], [3] from the maxima — the chain kernels of the next
//            step read the scale that is in the stream and bounds that hold for the weights that are in the stream.
//   floor: the scale is chosen for max(max|W|, floor).  The step path passes 16 lr: the finishing kernel re-scatters the UPDATED weights
//            with the scale chosen BEFORE the update, Adam moves a weight by at most ~lr per step, so a layer whose weights are smaller
//            than the step (a near-zero initialised layer) can no longer outgrow the fp16 range of its stream in one step (tx_piece_bits
//            would clamp silently).  Costs nothing: a weight 2^-15 below the scale's maximum still has all its 22 bits.
__global__ __launch_bounds__(64) void k_x3stats_final(int n_layers, float* __restrict__ meta, int post, float floor) {
    tx_stats_final(meta, (int)threadIdx.x, n_layers, post, floor);
}

int tnx3_launch_stats(const NetX3& n, const float* params, const int32_t* table, void* packed3, int post, hipStream_t stream, float scale_floor) {
    float* meta = reinterpret_cast<float*>(static_cast<unsigned char*>(packed3) + n.meta_off);
    if (!post) {                                                               // full scan (the buffer may be fresh memory: clear first)
        hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, reinterpret_cast<unsigned*>(meta), (n.depth + 1) * TX_META);
        TN_HIP_CHECK_LAUNCH("x3 weight statistics (clear)");
        hipLaunchKernelGGL(k_x3stats_scan, dim3((unsigned)((n.depth + 1) * TX_SCAN_NB)), dim3(256), 0, stream, params, table, n, meta);
        TN_HIP_CHECK_LAUNCH("x3 weight statistics (scan)");
    }
    hipLaunchKernelGGL(k_x3stats_final, dim3(1), dim3(64), 0, stream, n.depth + 1, meta, post, scale_floor);
    TN_HIP_CHECK_LAUNCH("x3 weight statistics");
    return TNERF_OK;
}

// ---- the x3 pipe's DOMAIN.  Its scales are one power of two per LAYER (weights) and per SAMPLE (activations); the fp16 pieces' own
// exponents carry an element down to 2^-15 of its block's maximum with all 22 bits, below that the second piece goes subnormal and
// bits are lost.  A layer in which a sizeable share of the weights sits that far below max|W_l| (one outlier 2^14+ above the rest),
// or whose largest bias lifts the activation bound 2^14 above the other biases, is outside the domain: measured (tests/
// test_gpu_round2.py::test_in_layer_outliers_*) as 3-13x the reference's own fp32 error in single gradient tensors.
// counts[4 l + {0,1,2,3}] = nonzero weights, weights below 2^-13 max|W_l|, nonzero biases, biases below 2^-14 max|b_l| of layer l
// (index depth = the heads); the maxima are the scale records' (tnerf_mlp_pack_x3 / the finishing kernel keep them).
__global__ __launch_bounds__(256) void k_x3domain(const float* __restrict__ params, const int32_t* __restrict__ table, NetX3 n,
                                                  const float* __restrict__ meta, unsigned* __restrict__ counts) {
    const int l = blockIdx.x / TX_SCAN_NB, part = blockIdx.x % TX_SCAN_NB;
    const int64_t fe = (int64_t)n.rec_frags * 512;
    const int64_t e0 = n.fw_rec0[l] * fe, e1 = n.fw_rec0[l + 1] * fe, n_w = (int64_t)(n.n_rec + n.n_bw_rec) * fe;
    const float wlim = meta[l * TX_META + 1] * 0x1p-13f, blim = meta[l * TX_META + 2] * 0x1p-14f;
    unsigned nz = 0, small = 0, bnz = 0, bsmall = 0;
    for (int64_t e = e0 + part * 256 + threadIdx.x; e < e1; e += TX_SCAN_NB * 256) {
        if (((e >> 9) % TX_NP) != 0) continue;
        const int32_t s = table[e];
        if (s < 0) continue;
        const float a = fabsf(params[s]);
        nz += a > 0.0f; small += (a > 0.0f && a < wlim);
    }
    if (part == 0) {
        const int nb = l < n.depth ? n.hidden : 4, b0 = l < n.depth ? l * n.hidden : n.depth * n.hidden;
        for (int j = threadIdx.x; j < nb; j += 256) {
            const int32_t s = table[n_w + b0 + j];
            if (s < 0) continue;
            const float a = fabsf(params[s]);
            bnz += a > 0.0f; bsmall += (a > 0.0f && a < blim);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        nz += __shfl_xor((int)nz, o, 64); small += __shfl_xor((int)small, o, 64); bnz += __shfl_xor((int)bnz, o, 64); bsmall += __shfl_xor((int)bsmall, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(counts + 4 * l, nz); atomicAdd(counts + 4 * l + 1, small);
        if (part == 0) { atomicAdd(counts + 4 * l + 2, bnz); atomicAdd(counts + 4 * l + 3, bsmall); }
    }
}

extern "C" int tnerf_x3_domain_counts(const tnerf_mlp_desc* d, const float* params, const int32_t* table, const void* packed3,
                                      uint32_t* counts, tnerf_stream_t stream) {
    NetX3 n; int rc = tn_build_netx3(d, &n); if (rc) return rc;
    if (!params || !table || !packed3 || !counts) {
        tn_set_error("tnerf_x3_domain_counts: params=%p table=%p packed3=%p counts=%p", (const void*)params, (const void*)table, packed3, (void*)counts);
        return TNERF_EINVAL;
    }
    hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, (hipStream_t)stream, counts, 4 * (n.depth + 1));
    TN_HIP_CHECK_LAUNCH("tnerf_x3_domain_counts (clear)");
    const float* meta = reinterpret_cast<const float*>(static_cast<const unsigned char*>(packed3) + n.meta_off);
    hipLaunchKernelGGL(k_x3domain, dim3((unsigned)((n.depth + 1) * TX_SCAN_NB)), dim3(256), 0, (hipStream_t)stream, params, table, n, meta, counts);
    TN_HIP_CHECK_LAUNCH("tnerf_x3_domain_counts");
    return TNERF_OK;
}

// stream element i (fp16) = piece ((i >> 9) mod TX_NP) of params[table[i]] 2^s (s: the layer's scale record, tx_piece_bits);
// the entries behind the stream are the fp32 biases.
__global__ __launch_bounds__(256) void k_packx3(const float* __restrict__ params, const int32_t* __restrict__ table, NetX3 n, int64_t n_w,
                                                unsigned short* __restrict__ out16, float* __restrict__ out32, const float* __restrict__ meta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n.pack_entries) return;
    const int32_t s = table[i];
    const float x = s >= 0 ? params[s] : 0.0f;
    if (i >= n_w) { out32[i - n_w] = x; return; }
    const int l = tx_record_layer(&n, (int)(i / ((int64_t)n.rec_frags * 512)));
    out16[i] = tx_piece_bits(x, meta[l * TX_META + 3], (int)((i >> 9) % TX_NP));
}

extern "C" int tnerf_mlp_pack_x3(const tnerf_mlp_desc* d, const float* params, const int32_t* table, void* packed3,
                                 tnerf_stream_t stream) {
    return tnerf_mlp_pack_x3_floor(d, params, table, packed3, 0.0f, stream);
}

extern "C" int tnerf_mlp_pack_x3_floor(const tnerf_mlp_desc* d, const float* params, const int32_t* table, void* packed3,
                                       float scale_floor, tnerf_stream_t stream) {
    NetX3 n; int rc = tn_build_netx3(d, &n); if (rc) return rc;
    if (!params || !table || !packed3 || !(scale_floor >= 0.0f)) {
        tn_set_error("tnerf_mlp_pack_x3: params=%p table=%p packed3=%p scale_floor=%g", (const void*)params, (const void*)table, packed3, scale_floor);
        return TNERF_EINVAL;
    }
    if ((rc = tnx3_launch_stats(n, params, table, packed3, 0, (hipStream_t)stream, scale_floor))) return rc;
    const int64_t n_w = (int64_t)(n.n_rec + n.n_bw_rec) * n.rec_frags * 512;
    unsigned char* base = static_cast<unsigned char*>(packed3);
    hipLaunchKernelGGL(k_packx3, dim3((unsigned)((n.pack_entries + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, table, n, n_w,
                       reinterpret_cast<unsigned short*>(base), reinterpret_cast<float*>(base + n.bias_off),
                       reinterpret_cast<const float*>(base + n.meta_off));
    TN_HIP_CHECK_LAUNCH("tnerf_mlp_pack_x3");
    return TNERF_OK;
}
