// Backward (dgrad) chain: what autograd derives from volume.py:18-42 and nerf.py:34-40, executed with
// the same register-resident MFMA orientation as the forward kernel.
//
// Per wavefront: one RAY (fused) or one 32-row tile (MLP only).  For each 32-sample tile the wave
//   1. forms dZ_head = dL/d(pre-activation of the rgb/sigma heads) from the composite backward
//      (fused) or from the upstream d_rgb/d_sigma (MLP only),
//   2. walks the layers backwards: dH_{l-1} = W_l^T dZ_l on MFMA, dZ_{l-1} = dH_{l-1} * (H_{l-1} > 0),
//   3. writes every dZ_l (feature-major) next to the forward's H_l stash.
// The weight gradients are then ONE batched GEMM per layer over all samples (wgrad.hip) — they
// cannot live in this kernel: dW is 1.9 MB of accumulators, a CU's register file holds 0.5 MB.
#include "mlp_core.hpp"
#include "mlp_args.hpp"

// dzh[4]: this lane's head gradients (r,g,b,sigma pre-activation) for sample m.
template <int HID>
__device__ __forceinline__ void tn_bwd_tile(const BwdArgs& a, const float (&dzh)[4], int64_t m, bool valid, int lane) {
    constexpr int NT = HID / 32;
    const MlpLayout& L = a.L;
    const int h = lane >> 5;
    const __amdgpu_buffer_rsrc_t wrsrc = tn_packed_rsrc(a.packed, L.packed_floats);
    float* __restrict__ stash = a.stash;
    const int64_t Mp = a.Mp;
    const int64_t ms = valid ? m : Mp + (lane & 31);           // padding lanes use the dump block
    float* __restrict__ pl = tn_stash_at(stash, L.stash_rows, ms) + 4 * h * 32;      // per-lane: (row 4h, sample ms)

    if (h == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pl[(L.dzh_row0 + i) * 32] = dzh[i];            // h == 0 here
    }
    float dz[HID / 2], dznext[HID / 2];
    // ReLU sign bits written by the training forward (layout: tnerf_internal.h)
    const uint32_t* __restrict__ mrow = reinterpret_cast<const uint32_t*>(stash + TN_STASH_BODY_FLOATS(L, Mp)) + (2 * ms + h) * (NT / 2);
    uint32_t mb[NT / 2];
    // ---- heads: dH_last[k] = sum_{n<4} W_head[n][k] dZ_head[n]; lane-half 1 supplies zeros (rows 4..7)
    {
        const int l = L.depth - 1;
#pragma unroll
        for (int w = 0; w < NT / 2; ++w) mb[w] = mrow[(int64_t)l * (Mp + 32) * NT + w];
        float* __restrict__ zrow = pl + L.dz_row0[l] * 32;
        const int sbh = (int)(L.bw_head * 4);
        const float b0 = h ? 0.0f : dzh[0], b1 = h ? 0.0f : dzh[1], b2 = h ? 0.0f : dzh[2], b3 = h ? 0.0f : dzh[3];
        tn_static_for<NT>([&](auto tc) TN_INLINE_LAMBDA {
            constexpr int t = decltype(tc)::value;
            const f32x4 a4 = tn_frag_load(wrsrc, lane * 16, sbh + t * 1024);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            acc = TN_MFMA(a4[0], b0, acc); acc = TN_MFMA(a4[1], b1, acc);
            acc = TN_MFMA(a4[2], b2, acc); acc = TN_MFMA(a4[3], b3, acc);
            tn_static_for<16>([&](auto rc) TN_INLINE_LAMBDA {
                constexpr int r = decltype(rc)::value;
                const float a_ = acc[r];                                                      // ReLU backward: output > 0
                const float v = __int_as_float(__float_as_int(a_) & __builtin_amdgcn_sbfe((int)mb[t / 2], (t & 1) * 16 + r, 1));
                dz[t * 16 + r] = v;
                TN_STASH_STORE(&zrow[(32 * t + (r & 3) + 8 * (r >> 2)) * 32], v);
            });
        });
    }
    // ---- hidden layers, last to first: dZ_l (in dz) -> dZ_{l-1}
    for (int l = L.depth - 1; l >= 1; --l) {
#pragma unroll
        for (int w = 0; w < NT / 2; ++w) mb[w] = mrow[(int64_t)(l - 1) * (Mp + 32) * NT + w];
        float* __restrict__ zrow = pl + L.dz_row0[l - 1] * 32;
        tn_layer_bwd<HID>(wrsrc, L.bw_hid[l], dz, lane, [&](auto tc, const f32x16& acc) TN_INLINE_LAMBDA {
            constexpr int t = decltype(tc)::value;
            tn_static_for<16>([&](auto rc) TN_INLINE_LAMBDA {
                constexpr int r = decltype(rc)::value;
                const float a_ = acc[r];
                const float v = __int_as_float(__float_as_int(a_) & __builtin_amdgcn_sbfe((int)mb[t / 2], (t & 1) * 16 + r, 1));
                dznext[t * 16 + r] = v;
                TN_STASH_STORE(&zrow[(32 * t + (r & 3) + 8 * (r >> 2)) * 32], v);
            });
        });
        tn_static_for<HID / 2>([&](auto ic) TN_INLINE_LAMBDA { dz[decltype(ic)::value] = dznext[decltype(ic)::value]; });
    }
}

// ------------------------------------------------------------------------------ MLP only
template <int HID>
__global__ __launch_bounds__(256, HID == 128 ? 2 : 1) void k_mlp_bwd(BwdArgs a) {
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (!tn_stash_tag_is(a.stash, a.L, a.Mp, TN_TAG_F32)) return;      // not this pipe's forward
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    if (m0 >= a.M) return;
    const int64_t m = m0 + (lane & 31);
    const bool valid = m < a.M;
    const int64_t mc = valid ? m : a.M - 1;
    float dzh[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float c = tn_stash_at(a.stash, a.L.stash_rows, mc)[(a.L.out_row0 + i) * 32];
        dzh[i] = valid ? a.d_rgb[3 * mc + i] * (c * (1.0f - c)) : 0.0f;            // sigmoid backward
    }
    const float sg = tn_stash_at(a.stash, a.L.stash_rows, mc)[(a.L.out_row0 + 3) * 32];
    dzh[3] = (valid && sg > 0.0f) ? a.d_sigma[mc] : 0.0f;                              // ReLU backward
    tn_bwd_tile<HID>(a, dzh, mc, valid, lane);
}

// ------------------------------------------------------------------------------ fused rays
// Composite backward per 64-sample segment (same closed form as k_composite_bwd in stage_kernels.hip,
// with rgb/sigma read from the forward's stash and only dL/dcomp_rgb upstream), then the dgrad chain
// for the segment's two 32-sample tiles.
template <int HID>
__global__ __launch_bounds__(256, HID == 128 ? 2 : 1) void k_train_bwd(BwdArgs a) {
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (!tn_stash_tag_is(a.stash, a.L, a.Mp, TN_TAG_F32)) return;      // not this pipe's forward
    const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
    if (ray >= a.R) return;
    RaySource rs = a.rs; SampleArgs sa = a.sa;
    tn_resolve_step(rs, sa);
    const int S = sa.S;
    const int nseg = (S + 63) / 64;
    float ro_[3], rd_[3];
    tn_fetch_ray(rs, ray, ro_, rd_);
    const float dn = tn_norm3(rd_[0], rd_[1], rd_[2]);
    const float gr = a.g_comp[a.g_stride * ray], gg = a.g_comp[a.g_stride * ray + 1], gb = a.g_comp[a.g_stride * ray + 2];
    const float gbg = a.white ? (gr + gg + gb) : 0.0f;
    const int64_t mray = ray * S; const int orow = a.L.out_row0 * 32; const int64_t SR = a.L.stash_rows;
    auto outv = [&](int i, int sc) TN_INLINE_LAMBDA { return tn_stash_at(a.stash, SR, mray + sc)[orow + 32 * i]; };

    float segprod = 1.0f;
    if (nseg > 1) {
        for (int g = 0; g < nseg; ++g) {
            const int s = g * 64 + lane; const bool ok = s < S; const int sc = ok ? s : S - 1;
            const float z = tn_depth(sa, ray, sc);
            const float zn = (s + 1 < S) ? tn_depth(sa, ray, s + 1) : z;
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
        const float z = tn_depth(sa, ray, sc);
        const float zn = (s + 1 < S) ? tn_depth(sa, ray, s + 1) : z;
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
        d4[0] = ok ? (w * gr) * (c0 * (1.0f - c0)) : 0.f;                          // sigmoid backward of dL/dc = w g
        d4[1] = ok ? (w * gg) * (c1 * (1.0f - c1)) : 0.f;
        d4[2] = ok ? (w * gb) * (c2 * (1.0f - c2)) : 0.f;
        d4[3] = (ok && sg > 0.0f) ? (da * t.e) * t.delta : 0.f;                    // ReLU backward of dL/dsigma
        tail += __shfl(suf, 0, 64);
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int sb = g * 64 + 32 * half;
            if (sb >= S) break;                                                    // wave-uniform
            float dzh[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) dzh[i] = __shfl(d4[i], 32 * half + (lane & 31), 64);
            const int st = sb + (lane & 31);
            const bool valid = st < S;
            tn_bwd_tile<HID>(a, dzh, ray * S + (valid ? st : S - 1), valid, lane);
        }
    }
}

// ----------------------------------------------------------------------------------- dispatch
int tn_launch_mlp_bwd(const BwdArgs& a, hipStream_t stream) {
    const dim3 grid((unsigned)(((a.M + 31) / 32 + 3) / 4)), block(256);
    if (a.L.hidden == 256) hipLaunchKernelGGL((k_mlp_bwd<256>), grid, block, 0, stream, a);
    else                   hipLaunchKernelGGL((k_mlp_bwd<128>), grid, block, 0, stream, a);
    TN_HIP_CHECK_LAUNCH("tnerf_mlp_bwd/dgrad");
    return TNERF_OK;
}

int tn_launch_train_bwd(const BwdArgs& a, hipStream_t stream) {
    const dim3 grid((unsigned)((a.R + 3) / 4)), block(256);
    if (a.L.hidden == 256) hipLaunchKernelGGL((k_train_bwd<256>), grid, block, 0, stream, a);
    else                   hipLaunchKernelGGL((k_train_bwd<128>), grid, block, 0, stream, a);
    TN_HIP_CHECK_LAUNCH("tnerf_train_bwd_fused/dgrad");
    return TNERF_OK;
}
