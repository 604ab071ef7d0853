// "Wave-pair" variants of the fused forward / dgrad kernels: TWO wavefronts per 32-sample tile, two waves per SIMD.
//
// Why (tools/mfma_feed_probe.hip, measured on MI355X): with one wave per SIMD every weight-fragment load costs ~29
// cycles of issue that nothing overlaps (71.2 instead of 64 cycles per v_mfma_f32_32x32x2_f32 with one 16-byte load
// per 4 MFMAs), and so does every other bubble of the wave (tile epilogues, layer boundaries, sin/cos, compositing).
// With two waves per SIMD the same stream runs at the 64-cycle pipe rate.  The single-wave kernels need the whole
// 512-register file (HID/2 activations + HID/2 outputs per lane), so here each layer's OUTPUT features are split
// between the two waves of a pair: both hold the full input activations (HID/2 registers), each computes half of the
// n-tiles (HID/4 output registers) and the halves are swapped through LDS once per layer (16 KB per wave, two
// s_barriers).  <= 256 registers per wave -> two 256-thread workgroups (2 pairs each) per CU.
//
// Register-logical order: hin[0 .. HID/4) = the features this wave produced ("own" half: physical n-tiles
// role*NT/2 ..), hin[HID/4 .. HID/2) = the partner's.  Only weight ADDRESSES depend on the role, so both roles run
// the same instruction stream.
#include "mlp_core.hpp"
#include "mlp_args.hpp"
#include <cstdio>
#include <cstdlib>

template <int HID> __global__ void k_train_bwd_pair(BwdArgs a);

#define TNP_PF 4

// wave-uniform float -> SGPR (the builtin is typed int: go through the bits)
__device__ __forceinline__ float tnp_uniform(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }

__device__ __forceinline__ void tnp_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One layer, this wave's half of the n-tiles.  fin(t', acc): t' = 0..NT/2-1 (physical tile role*NT/2 + t').
template <int HID, int NE, bool HAS_HID, bool HAS_ENC, typename Fin>
__device__ __forceinline__ void tnp_layer(const float* __restrict__ packed, int64_t off_bias, int64_t off_enc, int64_t off_hid,
                                          const float (&hin)[HID / 2], const float (&enc)[NE], int lane, int role, Fin&& fin) {
    constexpr int NT = HID / 32, NTH = NT / 2;
    constexpr int GH = HAS_HID ? NT * 4 : 0;
    constexpr int GE = HAS_ENC ? NE / 4 : 0;
    constexpr int GT = GH + GE;
    constexpr int TOTAL = NTH * GT;
    constexpr int PF = TNP_PF;
    // fragment (tile tp, k-block kbp, q) sits at ((tp*NT + kbp)*4 + q) KB; tp = role*NTH + t', kbp = own/partner half
    const f32x4* __restrict__ Wh = reinterpret_cast<const f32x4*>(packed + (HAS_HID ? off_hid : 0)) + lane + (int64_t)role * NTH * NT * 4 * 64;
    const f32x4* __restrict__ WhOwn = Wh + role * NTH * 4 * 64;
    const f32x4* __restrict__ WhPar = Wh + (1 - role) * NTH * 4 * 64;
    const f32x4* __restrict__ We = reinterpret_cast<const f32x4*>(packed + (HAS_ENC ? off_enc : 0)) + lane + (int64_t)role * NTH * GE * 64;
    const f32x4* __restrict__ Bf = reinterpret_cast<const f32x4*>(packed + off_bias) + (lane >> 5) * 4 + role * NTH * 8;

    auto frag = [&](auto ic) TN_INLINE_LAMBDA -> f32x4 {
        constexpr int i = decltype(ic)::value;
        constexpr int t = i / GT, g = i % GT;
        if constexpr (g < GH) {
            constexpr int kb = g / 4, q = g % 4;
            if constexpr (kb < NTH) return WhOwn[((t * NT + kb) * 4 + q) * 64];
            else                    return WhPar[((t * NT + (kb - NTH)) * 4 + q) * 64];
        } else {
            return We[(t * GE + (g - GH)) * 64];
        }
    };
    f32x4 ring[PF];
    tn_static_for<PF>([&](auto ic) TN_INLINE_LAMBDA { if constexpr (decltype(ic)::value < TOTAL) ring[decltype(ic)::value] = frag(ic); });
    tn_static_for<NTH>([&](auto tc) TN_INLINE_LAMBDA {
        constexpr int t = decltype(tc)::value;
        f32x16 acc;
        {
            const f32x4 b0 = Bf[t * 8 + 0], b1 = Bf[t * 8 + 1], b2 = Bf[t * 8 + 2], b3 = Bf[t * 8 + 3];
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[r] = b0[r]; acc[4 + r] = b1[r]; acc[8 + r] = b2[r]; acc[12 + r] = b3[r]; }
        }
        tn_static_for<GT>([&](auto gc) TN_INLINE_LAMBDA {
            constexpr int g = decltype(gc)::value;
            constexpr int i = t * GT + g;
            const f32x4 a4 = ring[i % PF];
            if constexpr (i + PF < TOTAL) ring[i % PF] = frag(std::integral_constant<int, i + PF>{});
            if constexpr (g < GH) {
                acc = TN_MFMA(a4[0], hin[g * 4 + 0], acc); acc = TN_MFMA(a4[1], hin[g * 4 + 1], acc);
                acc = TN_MFMA(a4[2], hin[g * 4 + 2], acc); acc = TN_MFMA(a4[3], hin[g * 4 + 3], acc);
            } else {
                constexpr int e = (g - GH) * 4;
                acc = TN_MFMA(a4[0], enc[e + 0], acc); acc = TN_MFMA(a4[1], enc[e + 1], acc);
                acc = TN_MFMA(a4[2], enc[e + 2], acc); acc = TN_MFMA(a4[3], enc[e + 3], acc);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        fin(tc, acc);
    });
}

// Exchange of the two half-outputs through LDS.  Each wave writes the 16 values of a finished n-tile straight into
// its half of the pair's buffer (no register copy of its own outputs), then after the layer:
//   barrier A (both halves complete) -> every wave reads BOTH halves into hin (own half first) -> barrier B (nobody
//   overwrites the buffer while the partner still reads).
// Element idx (= t*16 + r) of a wave's half lives at float offset ((idx/4)*64 + lane)*4 + idx%4.
template <int HID>
__device__ __forceinline__ void tnp_publish_tile(float* __restrict__ mine, int lane, int t, const float (&v)[16]) {
    f32x4* dst = reinterpret_cast<f32x4*>(mine) + (t * 4) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q * 64] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}
template <int HID>
__device__ __forceinline__ void tnp_gather(const float* __restrict__ xb, int role, int lane, float (&hin)[HID / 2]) {
    constexpr int Q = HID / 16;                                  // float4 pieces per lane and half
    const f32x4* mine = reinterpret_cast<const f32x4*>(xb + role * (HID / 4) * 64) + lane;
    const f32x4* theirs = reinterpret_cast<const f32x4*>(xb + (1 - role) * (HID / 4) * 64) + lane;
    tnp_lds_barrier();
    tn_static_for<Q>([&](auto ic) TN_INLINE_LAMBDA {
        constexpr int i = decltype(ic)::value;
        const f32x4 u = mine[i * 64], v = theirs[i * 64];
        hin[4 * i] = u[0]; hin[4 * i + 1] = u[1]; hin[4 * i + 2] = u[2]; hin[4 * i + 3] = u[3];
        hin[HID / 4 + 4 * i] = v[0]; hin[HID / 4 + 4 * i + 1] = v[1]; hin[HID / 4 + 4 * i + 2] = v[2]; hin[HID / 4 + 4 * i + 3] = v[3];
    });
    tnp_lds_barrier();
}

// ------------------------------------------------------------------------------------------ forward tile
template <int HID, int NE, bool TRAIN>
__device__ __forceinline__ void tnp_mlp_tile(const FwdArgs& a, float px, float py, float pz, int Lf, int64_t m, bool valid, int lane, int role,
                                             float* __restrict__ xb, float (&out4)[4]) {
    constexpr int NT = HID / 32, NTH = NT / 2;
    const MlpLayout& L = a.L;
    const int h = lane >> 5;
    const float* __restrict__ packed = a.packed;
    float* __restrict__ stash = a.stash;
    const int64_t Mp = a.Mp;
    const int64_t ms = valid ? m : Mp + (lane & 31);
    // per-lane stash pointer at (row 4h + 32*first own tile, sample ms); mask words of this wave's tiles
    float* __restrict__ pl = TRAIN ? tn_stash_at(stash, L.stash_rows, ms) + (4 * h + 32 * NTH * role) * 32 : nullptr;
    uint32_t* __restrict__ mrow = TRAIN ? reinterpret_cast<uint32_t*>(stash + TN_STASH_BODY_FLOATS(L, Mp)) + (2 * ms + h) * (NT / 2) + role * (NTH / 2) : nullptr;
    float hin[HID / 2];
    float* __restrict__ mine = xb + role * (HID / 4) * 64;
    uint32_t mb[NTH / 2 > 0 ? NTH / 2 : 1];
    for (int l = 0; l < L.depth; ++l) {
        float* __restrict__ srow = TRAIN ? pl + L.h_row0[l] * 32 : nullptr;
        auto fin = [&](auto tc, const f32x16& acc) TN_INLINE_LAMBDA {
            constexpr int t = decltype(tc)::value;
            if (TRAIN && (t & 1) == 0) mb[t / 2] = 0u;
            float o[16];
            tn_static_for<16>([&](auto rc) TN_INLINE_LAMBDA {
                constexpr int r = decltype(rc)::value;
                const float v = fmaxf(acc[r], 0.0f);
                o[r] = v;
                if (TRAIN) mb[t / 2] |= (v > 0.0f) ? (1u << ((t & 1) * 16 + r)) : 0u;
                if (TRAIN) srow[(32 * t + (r & 3) + 8 * (r >> 2)) * 32] = v;
            });
            tnp_publish_tile<HID>(mine, lane, t, o);
        };
        // The network input is needed by layer 0 and by the skip layer only: it is re-derived from the point there
        // (18-30 sin/cos, <1 % of a layer) instead of occupying NE registers through every layer in between.
        if (l == 0) {
            float enc[NE];
            tn_encode_point<NE>(px, py, pz, Lf, h, enc);
            if (TRAIN && role == 0) {
                float* __restrict__ pe = tn_stash_at(stash, L.stash_rows, ms);
                tn_static_for<NE>([&](auto sc_) TN_INLINE_LAMBDA { constexpr int st = decltype(sc_)::value; pe[(L.enc_row0 + 2 * st + h) * 32] = enc[st]; });
            }
            tnp_layer<HID, NE, false, true>(packed, L.fw_bias[0], L.fw_enc[0], 0, hin, enc, lane, role, fin);
        } else if (l == L.skip_at) {
            float enc[NE];
            tn_encode_point<NE>(px, py, pz, Lf, h, enc);
            tnp_layer<HID, NE, true, true>(packed, L.fw_bias[l], L.fw_enc[l], L.fw_hid[l], hin, enc, lane, role, fin);
        } else {
            float none[NE];
            tnp_layer<HID, NE, true, false>(packed, L.fw_bias[l], 0, L.fw_hid[l], hin, none, lane, role, fin);
        }
        if (TRAIN) {
#pragma unroll
            for (int w = 0; w < NTH / 2; ++w) mrow[(int64_t)l * (Mp + 32) * NT + w] = mb[w];
        }
        tnp_gather<HID>(xb, role, lane, hin);
    }
    // ---- heads (role 0 only; after the last swap both waves hold all HID activations)
    if (role == 0) {
        constexpr int GT = NT * 4;
        const f32x4* __restrict__ Wh = reinterpret_cast<const f32x4*>(packed + L.fw_head) + lane;
        const f32x4* __restrict__ Bf = reinterpret_cast<const f32x4*>(packed + L.fw_head_bias) + h * 4;
        f32x16 acc;
        const f32x4 b0 = Bf[0];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = r < 4 ? b0[r] : 0.0f;
        // logical k-block kb' of hin <-> physical block: own half first (role 0: identity)
        tn_static_for<GT>([&](auto gc) TN_INLINE_LAMBDA {
            constexpr int g = decltype(gc)::value;
            const f32x4 a4 = Wh[g * 64];
            acc = TN_MFMA(a4[0], hin[g * 4 + 0], acc); acc = TN_MFMA(a4[1], hin[g * 4 + 1], acc);
            acc = TN_MFMA(a4[2], hin[g * 4 + 2], acc); acc = TN_MFMA(a4[3], hin[g * 4 + 3], acc);
        });
#pragma unroll
        for (int i = 0; i < 3; ++i) out4[i] = 1.0f / (1.0f + expf(-acc[i]));
        out4[3] = fmaxf(acc[3], 0.0f);
    }
}

template <int HID, int NE, bool TRAIN>
__global__ __launch_bounds__(256, 2) void k_render_fused_pair(FwdArgs a) {
    __shared__ __attribute__((aligned(16))) float xlds[2 * 2 * (HID / 4) * 64 + 2 * 4 * 64];   // 2 pairs x 2 roles x HID/4 x 64 lanes, + head outputs
    float* vlds = xlds + 2 * 2 * (HID / 4) * 64;
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pair = wave >> 1, role = wave & 1;
    float* __restrict__ xb = xlds + pair * (2 * (HID / 4) * 64);
    const int64_t ray0 = (int64_t)blockIdx.x * 2 + pair;
    const bool live = ray0 < a.R;                              // a dead pair still walks every barrier
    const int64_t ray = live ? ray0 : a.R - 1;
    const int j = lane & 31, h = lane >> 5;
    const int S = a.sa.S;
    const int Lf = (a.L.in_dim - 3) / 6;
    float ro_[3], rd_[3];
    tn_fetch_ray(a.rs, ray, ro_, rd_);
    const float ox = ro_[0], oy = ro_[1], oz = ro_[2], dx = rd_[0], dy = rd_[1], dz = rd_[2];
    const float dn = tn_norm3(dx, dy, dz);
    // State that survives a tile lives outside the VGPRs (the MLP needs all 256): running sums and transmittance are
    // wave-uniform scalars (SGPRs), the lower half's head outputs wait in LDS for the upper half.
    float T_in = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, cd = 0.f, ca = 0.f;      // uniform
    float* __restrict__ vbuf = vlds + pair * 4 * 64 + lane;
    for (int sb = 0; sb < S; sb += 32) {
        const bool upper = (sb & 32) != 0;
        {
            const int s = sb + j;
            const bool valid = live && s < S;
            const int sc = s < S ? s : S - 1;
            const float z = tn_depth(a.sa, ray, sc);
            const float px = tn_point(ox, dx, z), py = tn_point(oy, dy, z), pz = tn_point(oz, dz, z);
            float res[4] = {0.f, 0.f, 0.f, 0.f};
            tnp_mlp_tile<HID, NE, TRAIN>(a, px, py, pz, Lf, ray * S + sc, valid, lane, role, xb, res);
            if (role == 0 && !upper && sb + 32 < S) {
#pragma unroll
                for (int i = 0; i < 4; ++i) vbuf[i * 64] = res[i];               // lanes 0..31 hold the lower 32 samples
            }
            if ((sb & 32) == 0 && sb + 32 < S) continue;
            if (role != 0) continue;                                 // compositing: the pair's role-0 wave
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float mv = __shfl(res[i], lane & 31, 64);             // upper half: lanes 32..63 <- lanes 0..31
                v[i] = upper ? (h ? mv : vbuf[i * 64]) : res[i];
            }
            const int s0 = sb & ~63;
            const int s2 = s0 + lane;
            const bool ok = s2 < S && (s2 - s0) < (upper ? 64 : 32);
            const int sc2 = s2 < S ? s2 : S - 1;
            const float z2 = tn_depth(a.sa, ray, sc2);
            const float zn = (s2 + 1 < S) ? tn_depth(a.sa, ray, s2 + 1) : z2;
            const CompTerms t = tn_comp_terms(ok ? v[3] : 0.0f, z2, zn, s2 == S - 1, dn);
            const float om = ok ? t.om : 1.0f;
            const float incl = tn_wave_scan_mul(om, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.0f;
            const float T = T_in * excl;
            const float w = ok ? t.alpha * T : 0.0f;
            cr += tnp_uniform(tn_wave_sum(w * v[0])); cg += tnp_uniform(tn_wave_sum(w * v[1]));
            cb += tnp_uniform(tn_wave_sum(w * v[2])); cd += tnp_uniform(tn_wave_sum(w * z2));
            ca += tnp_uniform(tn_wave_sum(w));
            T_in *= tnp_uniform(__shfl(incl, 63, 64));
            if (TRAIN && ok && live) {
#pragma unroll
                for (int i = 0; i < 4; ++i) tn_stash_at(a.stash, a.L.stash_rows, ray * S + s2)[(a.L.out_row0 + i) * 32] = v[i];
            }
        }
    }
    if (role != 0 || !live) return;
    if (lane == 0) {
        const float bg = a.white ? (1.0f - ca) : 0.0f;
        a.comp[3 * ray] = cr + bg; a.comp[3 * ray + 1] = cg + bg; a.comp[3 * ray + 2] = cb + bg;
        if (a.depth) a.depth[ray] = cd;
        if (a.acc) a.acc[ray] = ca;
    }
}

int tn_launch_fwd_pair(const FwdArgs& a, bool train, hipStream_t stream, const char* who) {
    const dim3 grid((unsigned)((a.R + 1) / 2)), block(256);
    if (getenv("TNERF_DEBUG_OCC")) {
        int nb = -1, nb2 = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_render_fused_pair<256, 20, true>, 256, 0);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, (const void*)k_train_bwd_pair<256>, 256, 0);
        fprintf(stderr, "[tnerf] occupancy: k_render_fused_pair<256,20,true> %d blocks/CU, k_train_bwd_pair<256> %d blocks/CU\n", nb, nb2);
    }
    const int hid = a.L.hidden, ne = a.L.NE;
#define TNP_CASE(H_, N_)                                                                                   \
    if (hid == H_ && ne == N_) {                                                                           \
        if (train) hipLaunchKernelGGL((k_render_fused_pair<H_, N_, true>), grid, block, 0, stream, a);     \
        else       hipLaunchKernelGGL((k_render_fused_pair<H_, N_, false>), grid, block, 0, stream, a);    \
        TN_HIP_CHECK_LAUNCH(who);                                                                          \
        return TNERF_OK;                                                                                   \
    }
    TNP_CASE(256, 20) TNP_CASE(256, 32) TNP_CASE(128, 20) TNP_CASE(128, 32)
#undef TNP_CASE
    tn_set_error("%s: no kernel for hidden=%d, input steps=%d", who, hid, ne);
    return TNERF_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------ backward tile
template <int HID, typename Fin>
__device__ __forceinline__ void tnp_layer_bwd(const float* __restrict__ packed, int64_t off_bw, const float (&dz)[HID / 2], int lane,
                                              int role, Fin&& fin) {
    constexpr int NT = HID / 32, NTH = NT / 2;
    constexpr int GT = NT * 4;
    constexpr int TOTAL = NTH * GT;
    constexpr int PF = TNP_PF;
    const f32x4* __restrict__ Wt = reinterpret_cast<const f32x4*>(packed + off_bw) + lane + (int64_t)role * NTH * NT * 4 * 64;
    const f32x4* __restrict__ WOwn = Wt + role * NTH * 4 * 64;
    const f32x4* __restrict__ WPar = Wt + (1 - role) * NTH * 4 * 64;
    auto frag = [&](auto ic) TN_INLINE_LAMBDA -> f32x4 {
        constexpr int i = decltype(ic)::value;
        constexpr int t = i / GT, g = i % GT, nb = g / 4, q = g % 4;
        if constexpr (nb < NTH) return WOwn[((t * NT + nb) * 4 + q) * 64];
        else                    return WPar[((t * NT + (nb - NTH)) * 4 + q) * 64];
    };
    f32x4 ring[PF];
    tn_static_for<PF>([&](auto ic) TN_INLINE_LAMBDA { ring[decltype(ic)::value] = frag(ic); });
    tn_static_for<NTH>([&](auto tc) TN_INLINE_LAMBDA {
        constexpr int t = decltype(tc)::value;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        tn_static_for<GT>([&](auto gc) TN_INLINE_LAMBDA {
            constexpr int g = decltype(gc)::value;
            constexpr int i = t * GT + g;
            const f32x4 a4 = ring[i % PF];
            if constexpr (i + PF < TOTAL) ring[i % PF] = frag(std::integral_constant<int, i + PF>{});
            acc = TN_MFMA(a4[0], dz[g * 4 + 0], acc); acc = TN_MFMA(a4[1], dz[g * 4 + 1], acc);
            acc = TN_MFMA(a4[2], dz[g * 4 + 2], acc); acc = TN_MFMA(a4[3], dz[g * 4 + 3], acc);
            __builtin_amdgcn_sched_barrier(0);
        });
        fin(tc, acc);
    });
}

template <int HID>
__device__ __forceinline__ void tnp_bwd_tile(const BwdArgs& a, const float (&dzh)[4], int64_t m, bool valid, int lane, int role,
                                             float* __restrict__ xb) {
    constexpr int NT = HID / 32, NTH = NT / 2;
    const MlpLayout& L = a.L;
    const int h = lane >> 5;
    const float* __restrict__ packed = a.packed;
    float* __restrict__ stash = a.stash;
    const int64_t Mp = a.Mp;
    const int64_t ms = valid ? m : Mp + (lane & 31);
    float* __restrict__ pl = tn_stash_at(stash, L.stash_rows, ms) + (4 * h + 32 * NTH * role) * 32;
    if (h == 0 && role == 0) {
        float* __restrict__ p0 = tn_stash_at(stash, L.stash_rows, ms);
#pragma unroll
        for (int i = 0; i < 4; ++i) p0[(L.dzh_row0 + i) * 32] = dzh[i];
    }
    const uint32_t* __restrict__ mrow = reinterpret_cast<const uint32_t*>(stash + TN_STASH_BODY_FLOATS(L, Mp)) + (2 * ms + h) * (NT / 2) + role * (NTH / 2);
    uint32_t mb[NTH / 2 > 0 ? NTH / 2 : 1];
    float dz[HID / 2];
    float* __restrict__ mine = xb + role * (HID / 4) * 64;
    // ---- heads -> own half of dZ_{depth-1}
    {
        const int l = L.depth - 1;
#pragma unroll
        for (int w = 0; w < NTH / 2; ++w) mb[w] = mrow[(int64_t)l * (Mp + 32) * NT + w];
        float* __restrict__ zrow = pl + L.dz_row0[l] * 32;
        const f32x4* __restrict__ Wt = reinterpret_cast<const f32x4*>(packed + L.bw_head) + lane + role * NTH * 64;
        const float b0 = h ? 0.0f : dzh[0], b1 = h ? 0.0f : dzh[1], b2 = h ? 0.0f : dzh[2], b3 = h ? 0.0f : dzh[3];
        tn_static_for<NTH>([&](auto tc) TN_INLINE_LAMBDA {
            constexpr int t = decltype(tc)::value;
            const f32x4 a4 = Wt[t * 64];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            acc = TN_MFMA(a4[0], b0, acc); acc = TN_MFMA(a4[1], b1, acc);
            acc = TN_MFMA(a4[2], b2, acc); acc = TN_MFMA(a4[3], b3, acc);
            float o[16];
            tn_static_for<16>([&](auto rc) TN_INLINE_LAMBDA {
                constexpr int r = decltype(rc)::value;
                const float v = ((mb[t / 2] >> ((t & 1) * 16 + r)) & 1u) ? acc[r] : 0.0f;
                o[r] = v;
                zrow[(32 * t + (r & 3) + 8 * (r >> 2)) * 32] = v;
            });
            tnp_publish_tile<HID>(mine, lane, t, o);
        });
        tnp_gather<HID>(xb, role, lane, dz);
    }
    for (int l = L.depth - 1; l >= 1; --l) {
#pragma unroll
        for (int w = 0; w < NTH / 2; ++w) mb[w] = mrow[(int64_t)(l - 1) * (Mp + 32) * NT + w];
        float* __restrict__ zrow = pl + L.dz_row0[l - 1] * 32;
        tnp_layer_bwd<HID>(packed, L.bw_hid[l], dz, lane, role, [&](auto tc, const f32x16& acc) TN_INLINE_LAMBDA {
            constexpr int t = decltype(tc)::value;
            float o[16];
            tn_static_for<16>([&](auto rc) TN_INLINE_LAMBDA {
                constexpr int r = decltype(rc)::value;
                const float v = ((mb[t / 2] >> ((t & 1) * 16 + r)) & 1u) ? acc[r] : 0.0f;
                o[r] = v;
                zrow[(32 * t + (r & 3) + 8 * (r >> 2)) * 32] = v;
            });
            if (l > 1) tnp_publish_tile<HID>(mine, lane, t, o);       // wave-uniform (dZ_0 is only stored)
        });
        if (l > 1) tnp_gather<HID>(xb, role, lane, dz);
    }
}

template <int HID>
__global__ __launch_bounds__(256, 2) void k_train_bwd_pair(BwdArgs a) {
    __shared__ __attribute__((aligned(16))) float xlds[2 * 2 * (HID / 4) * 64];
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pair = wave >> 1, role = wave & 1;
    float* __restrict__ xb = xlds + pair * (2 * (HID / 4) * 64);
    const int64_t ray0 = (int64_t)blockIdx.x * 2 + pair;
    const bool live = ray0 < a.R;
    const int64_t ray = live ? ray0 : a.R - 1;
    const int S = a.sa.S;
    const int nseg = (S + 63) / 64;
    float ro_[3], rd_[3];
    tn_fetch_ray(a.rs, ray, ro_, rd_);
    const float dn = tn_norm3(rd_[0], rd_[1], rd_[2]);
    const float gr = a.g_comp[3 * ray], gg = a.g_comp[3 * ray + 1], gb = a.g_comp[3 * ray + 2];
    const float gbg = a.white ? (gr + gg + gb) : 0.0f;
    const int64_t mray = ray * S; const int orow = a.L.out_row0 * 32; const int64_t SR = a.L.stash_rows;
    auto outv = [&](int i, int sc) TN_INLINE_LAMBDA { return tn_stash_at(a.stash, SR, mray + sc)[orow + 32 * i]; };

    float segprod = 1.0f;
    if (nseg > 1) {
        for (int g = 0; g < nseg; ++g) {
            const int s = g * 64 + lane; const bool ok = s < S; const int sc = ok ? s : S - 1;
            const float z = tn_depth(a.sa, ray, sc);
            const float zn = (s + 1 < S) ? tn_depth(a.sa, ray, s + 1) : z;
            const CompTerms t = tn_comp_terms(ok ? outv(3, sc) : 0.f, z, zn, s == S - 1, dn);
            const float p = tn_wave_prod(ok ? t.om : 1.0f);
            if (lane == g) segprod = p;
        }
    }
    const float seg_incl = tn_wave_scan_mul(segprod, lane);
    float seg_T = __shfl_up(seg_incl, 1, 64);
    if (lane == 0) seg_T = 1.0f;
    float tail = 0.0f;
    for (int g = nseg - 1; g >= 0; --g) {
        const int s = g * 64 + lane; const bool ok = s < S; const int sc = ok ? s : S - 1;
        const float c0 = outv(0, sc), c1 = outv(1, sc), c2 = outv(2, sc);
        const float sg = ok ? outv(3, sc) : 0.f;
        const float z = tn_depth(a.sa, ray, sc);
        const float zn = (s + 1 < S) ? tn_depth(a.sa, ray, s + 1) : z;
        const CompTerms t = tn_comp_terms(sg, z, zn, s == S - 1, dn);
        const float om = ok ? t.om : 1.0f;
        const float incl = tn_wave_scan_mul(om, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float T = __shfl(seg_T, g, 64) * excl;
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
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int sb = g * 64 + 32 * half;
            if (sb >= S) break;
            float dzh[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) dzh[i] = __shfl(d4[i], 32 * half + (lane & 31), 64);
            const int st = sb + (lane & 31);
            const bool valid = live && st < S;
            tnp_bwd_tile<HID>(a, dzh, ray * S + (st < S ? st : S - 1), valid, lane, role, xb);
        }
    }
}

int tn_launch_train_bwd_pair(const BwdArgs& a, hipStream_t stream) {
    const dim3 grid((unsigned)((a.R + 1) / 2)), block(256);
    if (a.L.hidden == 256) hipLaunchKernelGGL((k_train_bwd_pair<256>), grid, block, 0, stream, a);
    else                   hipLaunchKernelGGL((k_train_bwd_pair<128>), grid, block, 0, stream, a);
    TN_HIP_CHECK_LAUNCH("tnerf_train_bwd_fused/dgrad(pair)");
    return TNERF_OK;
}
