// bf16 mode, backward of the train step (what autograd derives from volume.py:18-42 and nerf.py:34-40):
//   k_dgrad16 : composite backward (fp32) + the dgrad chain dH_{l-1} = W_l^T dZ_l on bf16 MFMA with the transposed
//               weight stream shared through LDS (mlp16_core.hpp), dZ_l rounded to bf16 and stashed as the
//               K = samples operand of the weight-gradient MFMAs;
//   k_wgrad16 : dW_l = dZ_l^T H_{l-1} over all sample tiles: both operands stream HBM -> LDS (DMA) -> MFMA with no
//               transposition anywhere (the forward / dgrad kernels already wrote them sample-major per lane); fp32
//               accumulators -> per-workgroup slabs -> the deterministic slab reduction of the fp32 path.
#include "mlp16_core.hpp"
#include "mlp16_args.hpp"

// ---------------------------------------------------------------------------------------------- dgrad
// One transposed layer for the wave's tile.  KINDB 0: heads^T (one k-step, B = zh) 1: hidden^T (B = zin[0..KH)).
// mw: ReLU sign words of the layer whose activation gradient is produced.  ft0: its dZ feature tiles in the stash.
template <int HID, int KINDB>
__device__ __forceinline__ void tn16_layer_bwd(Pipe16& p, const unsigned char* lds, const bf16x8 (&zin)[HID / 16], const bf16x8& zh,
                                               bf16x8 (&zout)[HID / 16], f32x16& acc, const uint32_t (&mw)[HID / 64],
                                               const Stash16& st, uint32_t sel_off, int ft0) {
    constexpr int NT = HID / 32, KH = HID / 16;
    constexpr int KPT = KINDB == 0 ? 1 : KH;
    constexpr int NF = KINDB == 0 ? TN16_STAGE : NT * KH;        // fragments of the stream this layer consumes
    static_assert(NF % TN16_STAGE == 0 && NT <= TN16_STAGE, "a layer must be a whole number of stages");
    tn_static_for<NF / KPT>([&](auto tc) TN_INLINE_LAMBDA {
        constexpr int t = decltype(tc)::value;
        tn_static_for<KPT>([&](auto sc) TN_INLINE_LAMBDA {
            constexpr int s = decltype(sc)::value;
            constexpr int F = t * KPT + s;
            if constexpr (F % TN16_STAGE == 0) tn16_boundary<true>(p);
            const bf16x8 afrag = p.afr[F % TN16_PF];
            {
                constexpr int o = (F % TN16_STAGE) + TN16_PF;
                if constexpr (o < TN16_STAGE) p.afr[F % TN16_PF] = *reinterpret_cast<const bf16x8*>(lds + p.va_cur + o * 1024);
                else                          p.afr[F % TN16_PF] = *reinterpret_cast<const bf16x8*>(lds + p.va_nxt + (o - TN16_STAGE) * 1024);
            }
            if constexpr (t < NT) {
                if constexpr (s == 0) { const f32x16 z = {}; acc = TN16_MFMA(afrag, KINDB == 0 ? zh : zin[0], z); }
                else                  acc = TN16_MFMA(afrag, zin[s], acc);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (t < NT) {
            // ReLU backward: keep dH where the forward activation was positive, round to bf16
            const int m16 = (int)(mw[t / 2] >> ((t & 1) * 16));
            u32x4 w0, w1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k0 = __builtin_amdgcn_sbfe(m16, 2 * q, 1), k1 = __builtin_amdgcn_sbfe(m16, 2 * q + 1, 1);
                const int k2 = __builtin_amdgcn_sbfe(m16, 8 + 2 * q, 1), k3 = __builtin_amdgcn_sbfe(m16, 8 + 2 * q + 1, 1);
                const float a0 = acc[2 * q], a1 = acc[2 * q + 1], a2 = acc[8 + 2 * q], a3 = acc[8 + 2 * q + 1];
                w0[q] = tn16_cvt2(__int_as_float(__float_as_int(a0) & k0), __int_as_float(__float_as_int(a1) & k1));
                w1[q] = tn16_cvt2(__int_as_float(__float_as_int(a2) & k2), __int_as_float(__float_as_int(a3) & k3));
            }
            zout[2 * t] = __builtin_bit_cast(bf16x8, w0);
            zout[2 * t + 1] = __builtin_bit_cast(bf16x8, w1);
            tn16_stash_tile(lds, sel_off, p.lane16, st, ft0 + t, zout[2 * t], zout[2 * t + 1], acc);
        }
    });
}

template <int HID>
__device__ __forceinline__ void tn16_load_mask(uint32_t (&mw)[HID / 64], const Stash16& st, int l, uint32_t lane16) {
    typedef unsigned mvec __attribute__((ext_vector_type(HID / 64)));
    const mvec v = *reinterpret_cast<const mvec*>(st.mask + (int64_t)l * st.mask_lstride + lane16 / 16 * (HID / 16));
#pragma unroll
    for (int i = 0; i < HID / 64; ++i) mw[i] = v[i];
}

// dzh[4]: this lane's head gradients (r,g,b,sigma pre-activation); zero for slots past S.
template <int HID>
__device__ __forceinline__ void tn16_bwd_tile(Pipe16& p, const unsigned char* lds, const Net16& n, int h, const float (&dzh)[4],
                                              const Stash16& st, uint32_t sel_off) {
    constexpr int KH = HID / 16;
    const int depth = n.depth;
    bf16x8 X[KH], Y[KH];
    f32x16 acc;
    uint32_t mw[HID / 64], mwn[HID / 64];
    tn16_load_mask<HID>(mw, st, depth - 1, p.lane16);
    u32x4 zw = {0u, 0u, 0u, 0u};
    if (h == 0) { zw[0] = tn16_cvt2(dzh[0], dzh[1]); zw[1] = tn16_cvt2(dzh[2], dzh[3]); }
    const bf16x8 zh = __builtin_bit_cast(bf16x8, zw);
    {
        const u32x4 zero = {0u, 0u, 0u, 0u};
        tn16_stash_tile(lds, sel_off, p.lane16, st, n.ft_dzh, zh, __builtin_bit_cast(bf16x8, zero), acc);
    }
    // heads^T: dZ_{depth-1} = (W_head^T dZ_head) * (H_{depth-1} > 0)
    if (depth > 1) tn16_load_mask<HID>(mwn, st, depth - 2, p.lane16);
    tn16_layer_bwd<HID, 0>(p, lds, X, zh, X, acc, mw, st, sel_off, n.ft_dz[depth - 1]);
    // hidden layers, last to first: dZ_l (X) -> dZ_{l-1} (Y) and back
    int l = depth - 1;
    while (l >= 1) {
#pragma unroll
        for (int i = 0; i < HID / 64; ++i) mw[i] = mwn[i];
        if (l >= 2) tn16_load_mask<HID>(mwn, st, l - 2, p.lane16);
        tn16_layer_bwd<HID, 1>(p, lds, X, zh, Y, acc, mw, st, sel_off, n.ft_dz[l - 1]);
        if (--l < 1) break;
#pragma unroll
        for (int i = 0; i < HID / 64; ++i) mw[i] = mwn[i];
        if (l >= 2) tn16_load_mask<HID>(mwn, st, l - 2, p.lane16);
        tn16_layer_bwd<HID, 1>(p, lds, Y, zh, X, acc, mw, st, sel_off, n.ft_dz[l - 1]);
        --l;
    }
}

template <int HID>
__global__ __launch_bounds__(512, 2) void k_dgrad16(Fwd16Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    RaySource rs = a.rs; SampleArgs sa = a.sa;
    tn_resolve_step(rs, sa);
    const int S = sa.S;
    const uint32_t sel_off = TN16_SEL_OFF(a.n.n_bias);
    Pipe16 p;
    tn16_prologue(p, lds, a.packed, a.n, a.packed + (int64_t)a.n.n_frag * 1024, a.n.n_bw_stage, lane, wave, true);

    const int64_t n_groups = (a.R + 7) / 8;
    const int TPR = (S + 31) / 32;
    const int nseg = (S + 63) / 64;
    const unsigned char* mask0 = a.stash + TN16_STASH_FRAG_BYTES(a.n, a.n_tiles);
    const f32x4* out4 = reinterpret_cast<const f32x4*>(mask0 + TN16_STASH_MASK_BYTES(a.n, a.n_tiles));
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t ray = g * 8 + wave;
        const bool rvalid = ray < a.R;
        const int64_t rayc = rvalid ? ray : a.R - 1;
        float ro_[3], rd_[3];
        tn_fetch_ray(rs, rayc, ro_, rd_);
        const float dn = tn_norm3(rd_[0], rd_[1], rd_[2]);
        const int64_t gi = (int64_t)a.g_stride * rayc;
        const float gr = rvalid ? a.g_comp[gi] : 0.f, gg = rvalid ? a.g_comp[gi + 1] : 0.f, gb = rvalid ? a.g_comp[gi + 2] : 0.f;
        const float gbg = a.white ? (gr + gg + gb) : 0.0f;
        const int64_t tile0 = rayc * TPR;                                   // head outputs are read from the real tiles
        auto outv = [&](int sc) TN_INLINE_LAMBDA { return out4[(tile0 + (sc >> 5)) * 32 + (sc & 31)]; };

        float segprod = 1.0f;
        if (nseg > 1) {
            for (int sg_ = 0; sg_ < nseg; ++sg_) {
                const int s = sg_ * 64 + lane; const bool ok = s < S; const int sc = ok ? s : S - 1;
                const float z = tn_depth(sa, rayc, sc);
                const float zn = (s + 1 < S) ? tn_depth(sa, rayc, s + 1) : z;
                const CompTerms t = tn_comp_terms(ok ? outv(sc)[3] : 0.f, z, zn, s == S - 1, dn);
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
            const f32x4 o4 = outv(sc);
            const float c0 = o4[0], c1 = o4[1], c2 = o4[2];
            const float sg = ok ? o4[3] : 0.f;
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
                for (int i = 0; i < 4; ++i) dzh[i] = __shfl(d4[i], 32 * half + j, 64);
                const int64_t tile = rvalid ? tile0 + (sb >> 5) : a.n_tiles;
                Stash16 st;
                st.frag = a.stash + (tile * a.n.n_ft) * TN16_FT_BYTES;
                st.mask = const_cast<unsigned char*>(mask0) + tile * (64 * (HID / 64) * 4);
                st.mask_lstride = (a.n_tiles + 1) * (64 * (HID / 64) * 4);
                tn16_bwd_tile<HID>(p, lds, a.n, h, dzh, st, sel_off);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int tn16_launch_dgrad(const Fwd16Args& a, hipStream_t stream, const char* who) {
    const int dev = tn_stream_device(stream), n_cu = tn_device_cus(dev);
    const int64_t groups = (a.R + 7) / 8;
    const dim3 grid((unsigned)(groups < n_cu ? groups : n_cu)), block(512);
    const size_t lds_bytes = TN16_SEL_OFF(a.n.n_bias) + 2048;
    if (a.n.hidden == 256) {
        static std::atomic<uint32_t> seen_[TN_MAX_DEVICES];
        if (int rc = tn_grant_dyn_lds(reinterpret_cast<const void*>(&k_dgrad16<256>), lds_bytes, dev, seen_, who)) return rc;
        hipLaunchKernelGGL((k_dgrad16<256>), grid, block, lds_bytes, stream, a);
    } else {
        static std::atomic<uint32_t> seen_[TN_MAX_DEVICES];
        if (int rc = tn_grant_dyn_lds(reinterpret_cast<const void*>(&k_dgrad16<128>), lds_bytes, dev, seen_, who)) return rc;
        hipLaunchKernelGGL((k_dgrad16<128>), grid, block, lds_bytes, stream, a);
    }
    TN_HIP_CHECK_LAUNCH(who);
    return TNERF_OK;
}

// ---------------------------------------------------------------------------------------------- wgrad
#define TN16W_NS 4
#define TN16W_SLOT 32768                 // A fragments at +0 (<= 16 KB), B fragments at +16 KB
struct Wgrad16Args {
    const unsigned char* stash; const int32_t* jobs; float* slabs; int32_t n_ft;
    int64_t* step_inc;      // dataset mode: the device step counter advances in this kernel (see k_wgrad)
};

// (A non-temporal DMA for this once-read stream measured 5 % slower.)
// Per-wave DMA of one sample tile's operands: fragment f of the job's 2 (n_at + n_bt) goes to slot + (A: f, B: 16 + f') KB.
template <int PER>
__device__ __forceinline__ void tn16w_issue(const unsigned char* tile_base, int a_ft0, int b_ft0, int nfa, int nf, int wave, uint32_t lane16,
                                            uint32_t lds_slot) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        int f = wave * PER + k; f = f < nf ? f : nf - 1;
        const bool isa = f < nfa;
        const int fo = isa ? f : f - nfa;
        const unsigned char* src = tile_base + ((int64_t)(isa ? a_ft0 : b_ft0) * 2 + fo) * 1024;
        tn_glds16(src, lane16, lds_slot + (isa ? 0u : 16384u) + (uint32_t)fo * 1024u);
    }
}

template <int TA, int TB, int PER>
__device__ __forceinline__ void tn16w_body(const Wgrad16Args& a, const int32_t* job, unsigned char* lds, int lane, int wave) {
    const int a_ft0 = job[JOB_A_ROW0], b_ft0 = job[JOB_B_ROW0], n_at = job[JOB_N_AT], n_bt = job[JOB_N_BT], WA = job[JOB_WA];
    const int64_t t0 = job[JOB_MBLK0]; const int nt = job[JOB_MBLKN];
    const bool has_bias = job[JOB_HAS_BIAS] != 0;
    const int nfa = 2 * n_at, nf = 2 * (n_at + n_bt);
    const int wa = wave % WA, wb = wave / WA;
    const int at0 = wa * TA, bt0 = wb * TB;
    const bool active = at0 < n_at && bt0 < n_bt;
    const uint32_t lane16 = lane * 16;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    const int64_t tile_bytes = (int64_t)a.n_ft * TN16_FT_BYTES;
    auto tile_ptr = [&](int k) TN_INLINE_LAMBDA { const int kk = k < nt ? k : nt - 1; return a.stash + (t0 + kk) * tile_bytes; };

    f32x16 acc[TA][TB], accb[TA];
    const f32x16 zero = {};
#pragma unroll
    for (int i = 0; i < TA; ++i) { accb[i] = zero;
#pragma unroll
        for (int jx = 0; jx < TB; ++jx) acc[i][jx] = zero; }
    const u32x4 ones_w = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_w);

    // tiles 0..3 in flight; tile 0 landed and published
#pragma unroll
    for (int k = 0; k < TN16W_NS; ++k) tn16w_issue<PER>(tile_ptr(k), a_ft0, b_ft0, nfa, nf, wave, lane16, lds0 + k * TN16W_SLOT);
    TN16_WAIT_VM(3 * PER);
    __builtin_amdgcn_s_barrier();
    for (int k = 0; k < nt; ++k) {
        const unsigned char* slot = lds + (k & (TN16W_NS - 1)) * TN16W_SLOT + lane16;
        if (active) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 af[TA], bfr[TB];
#pragma unroll
                for (int i = 0; i < TA; ++i) af[i] = *reinterpret_cast<const bf16x8*>(slot + ((at0 + i) * 2 + u) * 1024);
#pragma unroll
                for (int jx = 0; jx < TB; ++jx) bfr[jx] = *reinterpret_cast<const bf16x8*>(slot + 16384 + ((bt0 + jx) * 2 + u) * 1024);
#pragma unroll
                for (int i = 0; i < TA; ++i) {
#pragma unroll
                    for (int jx = 0; jx < TB; ++jx) acc[i][jx] = TN16_MFMA(af[i], bfr[jx], acc[i][jx]);
                    if (has_bias && wb == 0) accb[i] = TN16_MFMA(af[i], ones, accb[i]);      // row sums: every column = sum over the samples
                }
            }
        }
        // own DMA of tile k+1 done (k+2, k+3 stay in flight); after the barrier everyone may read it and slot k is free
        TN16_WAIT_VM(2 * PER);
        __builtin_amdgcn_s_barrier();
        tn16w_issue<PER>(tile_ptr(k + TN16W_NS), a_ft0, b_ft0, nfa, nf, wave, lane16, lds0 + (k & (TN16W_NS - 1)) * TN16W_SLOT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!active) return;
    // slab: [n_at*32 rows][n_bt*32 cols] fp32, then the bias gradients [n_at*32]
    float* slab = a.slabs + job[JOB_SLAB_OFF];
    const int ld = n_bt * 32, c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < TA; ++i) {
#pragma unroll
        for (int jx = 0; jx < TB; ++jx)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[(int64_t)(32 * (at0 + i) + TN_ACC_ROW(r, h)) * ld + 32 * (bt0 + jx) + c] = acc[i][jx][r];
        if (has_bias && wb == 0 && c == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(int64_t)n_at * 32 * ld + 32 * (at0 + i) + TN_ACC_ROW(r, h)] = accb[i][r];
        }
    }
}

__global__ __launch_bounds__(512, 2) void k_wgrad16(Wgrad16Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int32_t* job = a.jobs + (int64_t)blockIdx.x * TN_JOB_INTS;
    if (a.step_inc && blockIdx.x == 0 && threadIdx.x == 0) *a.step_inc += 1;
    const int n_at = job[JOB_N_AT], n_bt = job[JOB_N_BT], WA = job[JOB_WA];
    const int ta = (n_at + WA - 1) / WA, tb = (n_bt + 8 / WA - 1) / (8 / WA);
    if (job[JOB_MBLKN] <= 0) return;
#ifdef TN_STAMPS   // diagnostic build: per-workgroup duration into the unused tail of the job record (tools/bf16_time_probe.py)
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
    // PER = DMA pieces per wave and sample tile = ceil(2 (n_at + n_bt) / 8)
    const int nf = 2 * (n_at + n_bt);
    if (ta == 2 && tb == 4)      tn16w_body<2, 4, 4>(a, job, lds, lane, wave);      // 8 x 8 tiles: 32 fragments per sample tile
    else if (ta == 1 && tb == 2) { if (nf <= 16) tn16w_body<1, 2, 2>(a, job, lds, lane, wave);       // 4 x 4 (128-wide layers): 16
                                   else          tn16w_body<1, 2, 3>(a, job, lds, lane, wave); }     // 8 x 2 (input): 20
    else if (ta == 2 && tb == 1) tn16w_body<2, 1, 4>(a, job, lds, lane, wave);
    else { if (nf <= 16)         tn16w_body<1, 1, 2>(a, job, lds, lane, wave);      // 4 x 2 input / 1 x 4 heads of 128-wide nets: 12 / 10
           else                  tn16w_body<1, 1, 3>(a, job, lds, lane, wave); }    // heads 1 x 8: 18
#ifdef TN_STAMPS
    if (threadIdx.x == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_start;
        int32_t* rec = const_cast<int32_t*>(a.jobs) + (int64_t)blockIdx.x * TN_JOB_INTS;
        rec[14] = (int32_t)(dt & 0xffffffffu); rec[15] = (int32_t)(dt >> 32);
    }
#endif
}

int tn16_launch_wgrad(const Net16& n, const unsigned char* stash, const int32_t* jobs, int64_t n_jobs, float* slabs, int64_t* step_inc, hipStream_t stream) {
    Wgrad16Args a{stash, jobs, slabs, n.n_ft, step_inc};
    const size_t lds_bytes = TN16W_NS * TN16W_SLOT;
    static std::atomic<uint32_t> seen_[TN_MAX_DEVICES];
    if (int rc = tn_grant_dyn_lds(reinterpret_cast<const void*>(&k_wgrad16), lds_bytes, tn_stream_device(stream), seen_, "tnerf_wgrad_bf16")) return rc;
    hipLaunchKernelGGL(k_wgrad16, dim3((unsigned)n_jobs), dim3(512), lds_bytes, stream, a);
    TN_HIP_CHECK_LAUNCH("tnerf_wgrad_bf16");
    return TNERF_OK;
}

// ---------------------------------------------------------------------------------------------- entry points
static int train16_args(const char* who, Fwd16Args& a, const tnerf_mlp_desc* d, const void* packed16, const RaySource& rs, int64_t R, int32_t S,
                        const float* ztab, int32_t randomized, const float* t_rand, uint64_t seed, uint64_t offset, int32_t white,
                        void* stash16) {
    int rc = tn16_fused_args(who, a, d, packed16, rs, R, S, ztab, randomized, t_rand, seed, offset, white); if (rc) return rc;
    if (R < 1 || !stash16) { tn_set_error("%s: R=%lld stash16=%p", who, (long long)R, stash16); return TNERF_EINVAL; }
    a.stash = static_cast<unsigned char*>(stash16);
    a.n_tiles = R * ((S + 31) / 32);
    return TNERF_OK;
}

static int train16_fwd_impl(const char* who, const tnerf_mlp_desc* d, const void* packed16, const RaySource& rs, const TnStepRef& sr,
                            const LossArgs& loss, int64_t R, int32_t S,
                            const float* ztab, int32_t randomized, const float* t_rand, uint64_t seed, uint64_t offset, int32_t white,
                            float* comp, void* stash16, hipStream_t stream) {
    Fwd16Args a{};
    int rc = train16_args(who, a, d, packed16, rs, R, S, ztab, randomized, t_rand, seed, offset, white, stash16); if (rc) return rc;
    if (!comp) { tn_set_error("%s: comp_rgb is NULL", who); return TNERF_EINVAL; }
    a.comp = comp; a.loss = loss;
    a.sa.step = sr.step; a.sa.per_step = sr.per_step;
    return tn16_launch_fwd(a, true, stream, who);
}

static int train16_dgrad_impl(const char* who, const tnerf_mlp_desc* d, const void* packed16, const RaySource& rs, const TnStepRef& sr,
                              int64_t R, int32_t S,
                              const float* ztab, int32_t randomized, const float* t_rand, uint64_t seed, uint64_t offset, int32_t white,
                              const float* g_comp, int32_t g_stride, void* stash16, hipStream_t stream) {
    Fwd16Args a{};
    int rc = train16_args(who, a, d, packed16, rs, R, S, ztab, randomized, t_rand, seed, offset, white, stash16); if (rc) return rc;
    if (!g_comp) { tn_set_error("%s: g_comp is NULL", who); return TNERF_EINVAL; }
    a.g_comp = g_comp; a.g_stride = g_stride;
    a.sa.step = sr.step; a.sa.per_step = sr.per_step;
    return tn16_launch_dgrad(a, stream, who);
}

extern "C" int tnerf_train_fwd_fused_bf16(const tnerf_mlp_desc* d, const void* packed16, const float* rays_o, const float* rays_d,
                                          int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                                          uint64_t seed, uint64_t offset, int32_t white, float* comp, void* stash16, tnerf_stream_t stream) {
    return train16_fwd_impl("tnerf_train_fwd_fused_bf16", d, packed16, tn_table_source(rays_o, rays_d), TnStepRef{}, LossArgs{}, R, S, ztab, randomized, t_rand, seed,
                            offset, white, comp, stash16, (hipStream_t)stream);
}

extern "C" int tnerf_train_dgrad_fused_bf16(const tnerf_mlp_desc* d, const void* packed16, const float* rays_o, const float* rays_d,
                                            int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                                            uint64_t seed, uint64_t offset, int32_t white, const float* g_comp, void* stash16,
                                            tnerf_stream_t stream) {
    return train16_dgrad_impl("tnerf_train_dgrad_fused_bf16", d, packed16, tn_table_source(rays_o, rays_d), TnStepRef{}, R, S, ztab, randomized, t_rand,
                              seed, offset, white, g_comp, 3, stash16, (hipStream_t)stream);
}

extern "C" int tnerf_wgrad_bf16(const tnerf_mlp_desc* d, const void* stash16, int64_t n_tiles, const int32_t* job_table, int64_t n_jobs,
                                float* slabs, tnerf_stream_t stream) {
    Net16 n; int rc = tn_build_net16(d, &n); if (rc) return rc;
    if (!stash16 || n_tiles < 1 || !job_table || n_jobs < 1 || !slabs) { tn_set_error("tnerf_wgrad_bf16: bad arguments"); return TNERF_EINVAL; }
    return tn16_launch_wgrad(n, static_cast<const unsigned char*>(stash16), job_table, n_jobs, slabs, nullptr, (hipStream_t)stream);
}

// forward (+ loss gradient per ray) -> dgrad -> wgrad: the bf16 step up to the slabs (see tn_step32_core).
int tn_step16_core(const char* who, const tnerf_mlp_desc* d, const void* packed16, const RaySource& rs, const TnStepRef& sr,
                   const LossArgs& loss, int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                   uint64_t seed, uint64_t offset, int32_t white, float* comp_rgb, void* stash16,
                   const int32_t* job_table, int64_t n_jobs, float* slabs, hipStream_t stream) {
    int rc = train16_fwd_impl(who, d, packed16, rs, sr, loss, R, S, ztab, randomized, t_rand, seed, offset, white, comp_rgb, stash16, stream);
    if (rc) return rc;
    if ((rc = train16_dgrad_impl(who, d, packed16, rs, sr, R, S, ztab, randomized, t_rand, seed, offset, white, loss.ray_ws, 4, stash16, stream))) return rc;
    Net16 n; if ((rc = tn_build_net16(d, &n))) return rc;
    return tn16_launch_wgrad(n, static_cast<const unsigned char*>(stash16), job_table, n_jobs, slabs, sr.step, stream);
}

static int train16_step_impl(const char* who, const tnerf_mlp_desc* d, const void* packed16, const RaySource& rs, const float* target,
                             const int64_t* target_index, int64_t R, int32_t S, const float* ztab, int32_t randomized,
                             const float* t_rand, uint64_t seed, uint64_t offset, int32_t white, double loss_denominator,
                             float* comp_rgb, float* g_comp_ws, int64_t ws_floats, float* loss_out, void* stash16, const int32_t* job_table,
                             int64_t n_jobs, float* slabs, const int32_t* reduce_table, float* grads, hipStream_t stream) {
    if (!target || !comp_rgb || !g_comp_ws || !loss_out || !(loss_denominator > 0.0) || R < 1 || !job_table || n_jobs < 1 || !slabs ||
        !reduce_table || !grads) {
        tn_set_error("%s: target=%p comp=%p g_ws=%p loss=%p denom=%g R=%lld jobs=%p n_jobs=%lld slabs=%p reduce=%p grads=%p", who,
                     (const void*)target, (void*)comp_rgb, (void*)g_comp_ws, (void*)loss_out, loss_denominator, (long long)R,
                     (const void*)job_table, (long long)n_jobs, (void*)slabs, (const void*)reduce_table, (void*)grads);
        return TNERF_EINVAL;
    }
    int rc = tn_check_ray_ws(who, R, ws_floats); if (rc) return rc;
    const LossArgs loss{target, target_index, (float)(1.0 / loss_denominator), g_comp_ws, nullptr};
    rc = tn_step16_core(who, d, packed16, rs, TnStepRef{}, loss, R, S, ztab, randomized, t_rand, seed, offset, white, comp_rgb, stash16,
                            job_table, n_jobs, slabs, stream);
    if (rc) return rc;
    FinishArgs f{};
    f.slabs = slabs; f.reduce_table = reduce_table; f.n_params = tnerf_param_count(d); f.grads = grads;
    f.ray_ws = g_comp_ws; f.R = R; f.inv_denom = loss.inv_denom; f.loss_out = loss_out;
    return tn_launch_finish(f, stream);
}

extern "C" int tnerf_train_step_fused_bf16(const tnerf_mlp_desc* d, const void* packed16, const float* rays_o, const float* rays_d,
                                           const float* target, int64_t R, int32_t S, const float* ztab, int32_t randomized,
                                           const float* t_rand, uint64_t seed, uint64_t offset, int32_t white, double loss_denominator,
                                           float* comp_rgb, float* g_comp_ws, int64_t g_comp_ws_floats, float* loss_out, void* stash16, const int32_t* job_table,
                                           int64_t n_jobs, float* slabs, const int32_t* reduce_table, float* grads, tnerf_stream_t stream) {
    return train16_step_impl("tnerf_train_step_fused_bf16", d, packed16, tn_table_source(rays_o, rays_d), target, nullptr, R, S, ztab,
                             randomized, t_rand, seed, offset, white, loss_denominator, comp_rgb, g_comp_ws, g_comp_ws_floats, loss_out, stash16, job_table,
                             n_jobs, slabs, reduce_table, grads, (hipStream_t)stream);
}

extern "C" int tnerf_train_step_fused_cam_bf16(const tnerf_mlp_desc* d, const void* packed16, const tnerf_camera* cam, const float* pixels,
                                               int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                                               uint64_t seed, uint64_t offset, int32_t white, double loss_denominator, float* comp_rgb,
                                               float* g_comp_ws, int64_t g_comp_ws_floats, float* loss_out, void* stash16, const int32_t* job_table, int64_t n_jobs,
                                               float* slabs, const int32_t* reduce_table, float* grads, tnerf_stream_t stream) {
    RaySource rs;
    int rc = tn_camera_source("tnerf_train_step_fused_cam_bf16", cam, R, &rs); if (rc) return rc;
    if (!cam->pix_index) { tn_set_error("tnerf_train_step_fused_cam_bf16: pix_index is required (it also selects the target pixels)"); return TNERF_EINVAL; }
    return train16_step_impl("tnerf_train_step_fused_cam_bf16", d, packed16, rs, pixels, cam->pix_index, R, S, ztab, randomized, t_rand, seed,
                             offset, white, loss_denominator, comp_rgb, g_comp_ws, g_comp_ws_floats, loss_out, stash16, job_table, n_jobs, slabs, reduce_table,
                             grads, (hipStream_t)stream);
}
