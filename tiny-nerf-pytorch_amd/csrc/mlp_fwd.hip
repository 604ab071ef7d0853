// Forward kernels: TinyNeRF.forward on its own (x from HBM) and the fused
// ray -> sample -> encode -> MLP -> composite path of render_one / the train step
// (reference src/nerf.py:29-41, src/train.py:50-56 and :114-121).
//
// Launch geometry: 256-thread workgroups = 4 wavefronts = one per SIMD (the kernel uses the whole
// 512-entry VGPR+AGPR file of a SIMD: HID/2 activation registers + HID/2 accumulators + the
// weight-fragment ring).  Each wavefront owns one unit of work: a RAY in the fused kernels (its S
// samples are marched 32 at a time through the register-resident MLP, then alpha-composited in
// registers with a wave-shuffle exclusive scan), or a 32-row tile of x in the MLP-only kernel.
#include "mlp_core.hpp"
#include "mlp_args.hpp"

#ifndef TN_DEBUG_DYN_LDS
#define TN_DEBUG_DYN_LDS 0      // diagnostic builds: dynamic LDS bytes requested only to cap residency at 1 workgroup/CU
#endif
#ifdef TN_STAMPS
#define TN_STAMP(k) do { if (a.stamps && lane == 0) a.stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
// the constant 100 MHz counter next to the shader-clock counter: their ratio is the clock the kernel really ran at
#define TN_STAMP_RT(k) do { if (a.stamps && lane == 0) a.stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TN_STAMP(k) do {} while (0)
#define TN_STAMP_RT(k) do {} while (0)
#endif

// 16 ReLU sign bits (bit r <-> v[r] > 0) of an n-tile's non-negative outputs, merged into the low (ODD = 0) or high half
// of a mask word.  v >= 0, so v's bit pattern + 0x7FFFFFFF carries into bit 31 exactly when v != 0; alignbit shifts
// that bit into the word: 2 VALU per value instead of compare + select + or.
template <int ODD>
__device__ __forceinline__ uint32_t tn_sign_bits16(uint32_t word, const float* v) {
    uint32_t m = 0u;
#pragma unroll
    for (int r = 15; r >= 0; --r) m = __builtin_amdgcn_alignbit(m, __float_as_uint(v[r]) + 0x7FFFFFFFu, 31);
    return ODD ? (word | (m << 16)) : m;
}

// The MLP for one 32-sample tile.  enc: network input registers.  m: this lane's sample index in the
// stash / output (valid if `valid`).  Returns the 4 head outputs (r,g,b after sigmoid; sigma after
// ReLU) in out4 — meaningful on lane-half 0 only.
template <int HID, int NE, bool TRAIN>
__device__ __forceinline__ void tn_mlp_tile(const FwdArgs& a, const float (&enc)[NE], int64_t m, bool valid, int lane,
                                            float (&out4)[4]) {
    constexpr int NT = HID / 32;
    const MlpLayout& L = a.L;
    const int h = lane >> 5;
    const __amdgpu_buffer_rsrc_t wrsrc = tn_packed_rsrc(a.packed, L.packed_floats);
    float* __restrict__ stash = a.stash;
    const int64_t Mp = a.Mp;
    const int64_t ms = valid ? m : Mp + (lane & 31);           // padding lanes write to the dump block: stores need no branch
    float* __restrict__ pl = TRAIN ? tn_stash_at(stash, L.stash_rows, ms) + 4 * h * 32 : nullptr;   // per-lane: (row 4h, sample ms)

    if (TRAIN) {
        tn_static_for<NE>([&](auto sc_) TN_INLINE_LAMBDA {
            constexpr int st = decltype(sc_)::value;
            TN_STASH_STORE(&pl[(L.enc_row0 + 2 * st - 3 * h) * 32], enc[st]);          // row enc_row0 + 2 st + h
        });
    }

    float hcur[HID / 2], hnext[HID / 2];
    // ReLU sign bits of the layer being computed (training only): the backward chain masks with these
    // instead of re-reading 1 KB of activations per sample and layer.
    uint32_t mb[NT / 2];
    uint32_t* __restrict__ mrow = TRAIN ? reinterpret_cast<uint32_t*>(stash + TN_STASH_BODY_FLOATS(L, Mp)) + (2 * ms + h) * (NT / 2) : nullptr;
    TN_STAMP(1);
    // ---- layer 0: input only
    {
        float* __restrict__ srow = TRAIN ? pl + L.h_row0[0] * 32 : nullptr;
        tn_layer<HID, NE, false, true>(wrsrc, L.fw_bias[0], L.fw_enc[0], 0, hnext, enc, lane,
            [&](auto tc, const f32x16& acc) TN_INLINE_LAMBDA {
                constexpr int t = decltype(tc)::value;
                tn_static_for<16>([&](auto rc) TN_INLINE_LAMBDA {
                    constexpr int r = decltype(rc)::value;
                    const float v = fmaxf(acc[r], 0.0f);
                    hcur[t * 16 + r] = v;
                    if (TRAIN) TN_STASH_STORE(&srow[(32 * t + (r & 3) + 8 * (r >> 2)) * 32], v);
                });
                if (TRAIN) mb[t / 2] = tn_sign_bits16<t & 1>(mb[t / 2], &hcur[t * 16]);
            });
        if (TRAIN) {
#pragma unroll
            for (int w = 0; w < NT / 2; ++w) mrow[w] = mb[w];
        }
    }
    TN_STAMP(2);
    // ---- hidden layers
    for (int l = 1; l < L.depth; ++l) {
        float* __restrict__ srow = TRAIN ? pl + L.h_row0[l] * 32 : nullptr;
        auto fin = [&](auto tc, const f32x16& acc) TN_INLINE_LAMBDA {
            constexpr int t = decltype(tc)::value;
            tn_static_for<16>([&](auto rc) TN_INLINE_LAMBDA {
                constexpr int r = decltype(rc)::value;
                const float v = fmaxf(acc[r], 0.0f);
                hnext[t * 16 + r] = v;
                if (TRAIN) TN_STASH_STORE(&srow[(32 * t + (r & 3) + 8 * (r >> 2)) * 32], v);
            });
            if (TRAIN) mb[t / 2] = tn_sign_bits16<t & 1>(mb[t / 2], &hnext[t * 16]);
        };
        if (l == L.skip_at) tn_layer<HID, NE, true, true>(wrsrc, L.fw_bias[l], L.fw_enc[l], L.fw_hid[l], hcur, enc, lane, fin);
        else                tn_layer<HID, NE, true, false>(wrsrc, L.fw_bias[l], 0, L.fw_hid[l], hcur, enc, lane, fin);
        tn_static_for<HID / 2>([&](auto ic) TN_INLINE_LAMBDA { hcur[decltype(ic)::value] = hnext[decltype(ic)::value]; });
        if (TRAIN) {
#pragma unroll
            for (int w = 0; w < NT / 2; ++w) mrow[(int64_t)l * (Mp + 32) * NT + w] = mb[w];   // layer stride = (Mp+32) * 2 halves * NT/2 words
        }
        TN_STAMP(2 + l);
    }
    // ---- heads: one n-tile whose rows 0..2 are rgb.0 and row 3 is sigma.0 (rows 4..31 are zero)
    {
        constexpr int GT = NT * 4;
        const int shd = (int)(L.fw_head * 4);
        f32x16 acc;
        const f32x4 b0 = tn_frag_load(wrsrc, h * 64, (int)(L.fw_head_bias * 4));
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = r < 4 ? b0[r] : 0.0f;
        tn_static_for<GT>([&](auto gc) TN_INLINE_LAMBDA {
            constexpr int g = decltype(gc)::value;
            const f32x4 a4 = tn_frag_load(wrsrc, lane * 16, shd + g * 1024);
            acc = TN_MFMA(a4[0], hcur[g * 4 + 0], acc); acc = TN_MFMA(a4[1], hcur[g * 4 + 1], acc);
            acc = TN_MFMA(a4[2], hcur[g * 4 + 2], acc); acc = TN_MFMA(a4[3], hcur[g * 4 + 3], acc);
        });
#pragma unroll
        for (int i = 0; i < 3; ++i) out4[i] = 1.0f / (1.0f + expf(-acc[i]));   // torch.sigmoid   nerf.py:27,39
        out4[3] = fmaxf(acc[3], 0.0f);                                          // ReLU            nerf.py:26,40
    }
    TN_STAMP(12);
}

// ------------------------------------------------------------------------------ MLP only
template <int HID, int NE, bool TRAIN>
__global__ __launch_bounds__(256, HID == 128 ? 2 : 1) void k_mlp_fwd(FwdArgs a) {
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int64_t m0 = tile * 32;
    if (m0 >= a.M) return;
    const int j = lane & 31, h = lane >> 5;
    const int64_t m = m0 + j;
    const bool valid = m < a.M;
    float enc[NE];
    tn_load_input<NE>(a.x, m, valid, a.L, h, enc);
    float out4[4];
    tn_mlp_tile<HID, NE, TRAIN>(a, enc, m, valid, lane, out4);
    if (valid && h == 0) {
        a.rgb_out[3 * m + 0] = out4[0]; a.rgb_out[3 * m + 1] = out4[1]; a.rgb_out[3 * m + 2] = out4[2];
        a.sigma_out[m] = out4[3];
        if (TRAIN) {
#pragma unroll
            for (int i = 0; i < 4; ++i) tn_stash_at(a.stash, a.L.stash_rows, m)[(a.L.out_row0 + i) * 32] = out4[i];
        }
    }
}

// ------------------------------------------------------------------------------ fused rays
#ifndef TN_NE32_TRAIN_WAVES
#define TN_NE32_TRAIN_WAVES 1      // waves per SIMD of the 128-wide TRAINING forward with 32 input steps (L = 10): at 2 (256 registers) it
#endif                             // kept 25 values in scratch memory; 1 = the 512-register budget, nothing spilled (tests/test_kernel_resources.py)
template <int HID, int NE, bool TRAIN>
__global__ __launch_bounds__(256, HID == 128 ? (NE == 32 && TRAIN ? TN_NE32_TRAIN_WAVES : 2) : 1) void k_render_fused(FwdArgs a) {
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
    if (ray >= a.R) return;
    const int j = lane & 31, h = lane >> 5;
    RaySource rs = a.rs; SampleArgs sa = a.sa;
    if (TRAIN) tn_resolve_step(rs, sa);                    // dataset mode: this step's image and Philox counters
    const int S = sa.S;
    const int Lf = (a.L.in_dim - 3) / 6;
    float ro_[3], rd_[3];
    tn_fetch_ray(rs, ray, ro_, rd_);
    const float ox = ro_[0], oy = ro_[1], oz = ro_[2], dx = rd_[0], dy = rd_[1], dz = rd_[2];
    const float dn = tn_norm3(dx, dy, dz);
    TN_STAMP(0); TN_STAMP_RT(20);
    float T_in = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, cd = 0.f, ca = 0.f;

    // March the ray 32 samples at a time (ONE inlined copy of the MLP body); every second tile (or the
    // last one) the 64 lanes composite a segment: lane l <- sample s0 + l.
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int sb = 0; sb < S; sb += 32) {
        {
            const int s = sb + j;
            const bool valid = s < S;
            const int sc = valid ? s : S - 1;
            const float z = tn_depth(sa, ray, sc);
            const float px = tn_point(ox, dx, z), py = tn_point(oy, dy, z), pz = tn_point(oz, dz, z);
            float enc[NE];
            tn_encode_point<NE>(px, py, pz, Lf, h, enc);
            float res[4];
            tn_mlp_tile<HID, NE, TRAIN>(a, enc, ray * S + sc, valid, lane, res);
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
        const float z = tn_depth(sa, ray, sc);
        const float zn = (s + 1 < S) ? tn_depth(sa, ray, s + 1) : z;
        const CompTerms t = tn_comp_terms(ok ? v[3] : 0.0f, z, zn, s == S - 1, dn);       // volume.py:18-31
        const float om = ok ? t.om : 1.0f;
        const float incl = tn_wave_scan_mul(om, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0f;
        const float T = T_in * excl;
        const float w = ok ? t.alpha * T : 0.0f;                                             // volume.py:34
        cr += w * v[0]; cg += w * v[1]; cb += w * v[2]; cd += w * z; ca += w;                // volume.py:36-38
        T_in *= __shfl(incl, 63, 64);
        if (TRAIN && ok) {
#pragma unroll
            for (int i = 0; i < 4; ++i) tn_stash_at(a.stash, a.L.stash_rows, ray * S + s)[(a.L.out_row0 + i) * 32] = v[i];
        }
    }
    TN_STAMP(13);
    cr = tn_wave_sum(cr); cg = tn_wave_sum(cg); cb = tn_wave_sum(cb); cd = tn_wave_sum(cd); ca = tn_wave_sum(ca);
    if (lane == 0) {
        const float bg = a.white ? (1.0f - ca) : 0.0f;                                       // volume.py:42
        a.comp[3 * ray] = cr + bg; a.comp[3 * ray + 1] = cg + bg; a.comp[3 * ray + 2] = cb + bg;
        if (a.depth) a.depth[ray] = cd;
        if (a.acc) a.acc[ray] = ca;
        if (TRAIN && a.loss.ray_ws) tn_ray_loss(a.loss, rs, ray, cr + bg, cg + bg, cb + bg);      // train.py:122
    }
    TN_STAMP(14); TN_STAMP_RT(21);
}

// ----------------------------------------------------------------------------------- dispatch
__global__ void k_stash_tag(unsigned* __restrict__ w, unsigned tag) { if (threadIdx.x == 0) *w = tag; }

template <bool FUSED, bool TRAIN>
static int launch_fwd(const FwdArgs& a, int64_t units, hipStream_t stream, const char* who) {
    if (TRAIN) {       // this pipe's stash (tnerf_internal.h TNB_TAG): the x3 backward kernels refuse it, tnerf_wgrad picks the fp32 body
        hipLaunchKernelGGL(k_stash_tag, dim3(1), dim3(64), 0, stream, reinterpret_cast<unsigned*>(a.stash + TN_BOUND_OFF(a.L, a.Mp)) + TNB_TAG, TN_TAG_F32);
        TN_HIP_CHECK_LAUNCH(who);
    }
    const dim3 grid((unsigned)((units + 3) / 4)), block(256);
    const int hid = a.L.hidden, ne = a.L.NE;
#define TN_CASE(H_, N_)                                                                                     \
    if (hid == H_ && ne == N_) {                                                                            \
        if (FUSED) hipLaunchKernelGGL((k_render_fused<H_, N_, TRAIN>), grid, block, TN_DEBUG_DYN_LDS, stream, a);  \
        else       hipLaunchKernelGGL((k_mlp_fwd<H_, N_, TRAIN>), grid, block, TN_DEBUG_DYN_LDS, stream, a);      \
        TN_HIP_CHECK_LAUNCH(who);                                                                           \
        return TNERF_OK;                                                                                    \
    }
    TN_CASE(256, 20) TN_CASE(256, 32) TN_CASE(128, 20) TN_CASE(128, 32)
#undef TN_CASE
    tn_set_error("%s: no kernel for hidden=%d, input steps=%d", who, hid, ne);
    return TNERF_EUNSUPPORTED;
}

int tn_launch_fwd(const FwdArgs& a, bool fused, bool train, int64_t units, hipStream_t stream, const char* who) {
    if (fused) return train ? launch_fwd<true, true>(a, units, stream, who) : launch_fwd<true, false>(a, units, stream, who);
    return train ? launch_fwd<false, true>(a, units, stream, who) : launch_fwd<false, false>(a, units, stream, who);
}

extern "C" int tnerf_mlp_fwd(const tnerf_mlp_desc* d, const float* packed, const float* x, int64_t M, float* rgb, float* sigma,
                             float* stash, int64_t Mp, tnerf_stream_t stream) {
    FwdArgs a{};
    int rc = tn_build_layout(d, &a.L); if (rc) return rc;
    if (M == 0) return TNERF_OK;                                   // empty batch: nothing to read or write
    if (M < 0 || !packed || !x || !rgb || !sigma || (stash && Mp < M)) {
        tn_set_error("tnerf_mlp_fwd: M=%lld packed=%p x=%p rgb=%p sigma=%p Mp=%lld", (long long)M, (const void*)packed,
                     (const void*)x, (void*)rgb, (void*)sigma, (long long)Mp);
        return TNERF_EINVAL;
    }
    if (M == 0) return TNERF_OK;
    a.packed = packed; a.x = x; a.M = M; a.rgb_out = rgb; a.sigma_out = sigma; a.stash = stash; a.Mp = Mp;
    const int64_t tiles = (M + 31) / 32;
    return stash ? launch_fwd<false, true>(a, tiles, (hipStream_t)stream, "tnerf_mlp_fwd")
                 : launch_fwd<false, false>(a, tiles, (hipStream_t)stream, "tnerf_mlp_fwd");
}

int tn_camera_source(const char* who, const tnerf_camera* cam, int64_t R, RaySource* rs) {
    if (!cam || (R > 0 && !cam->c2w) || cam->H < 1 || cam->W < 1 || !(cam->focal != 0.0f) ||
        (!cam->pix_index && (cam->pix_first < 0 || cam->pix_first + R > (int64_t)cam->H * cam->W))) {
        tn_set_error("%s: bad tnerf_camera (c2w=%p H=%d W=%d focal=%g pix_index=%p pix_first=%lld, %lld rays)", who,
                     cam ? (const void*)cam->c2w : nullptr, cam ? cam->H : 0, cam ? cam->W : 0, cam ? cam->focal : 0.f,
                     cam ? (const void*)cam->pix_index : nullptr, cam ? (long long)cam->pix_first : 0LL, (long long)R);
        return TNERF_EINVAL;
    }
    *rs = RaySource{nullptr, nullptr, cam->pix_index, cam->c2w, cam->pix_first, cam->H, cam->W, cam->focal};
    return TNERF_OK;
}

int tn_fused_args(const char* who, FwdArgs& a, const tnerf_mlp_desc* d, const float* packed, const RaySource& rs,
                  int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                  uint64_t seed, uint64_t offset, int32_t white) {
    int rc = tn_build_layout(d, &a.L); if (rc) return rc;
    if (a.L.in_dim < 9 || (a.L.in_dim - 3) % 6 != 0) {
        tn_set_error("%s: the fused path needs in_dim = 6L+3 (PositionalEncoding with include_input); got %d", who, a.L.in_dim);
        return TNERF_EUNSUPPORTED;
    }
    if (R < 0 || S < 1 || S > 4096 || (R > 0 && (!packed || !ztab || (!rs.c2w && (!rs.rays_o || !rs.rays_d))))) {
        tn_set_error("%s: R=%lld S=%d (1..4096) packed=%p rays_o=%p rays_d=%p c2w=%p ztab=%p", who, (long long)R, S, (const void*)packed,
                     (const void*)rs.rays_o, (const void*)rs.rays_d, (const void*)rs.c2w, (const void*)ztab);
        return TNERF_EINVAL;
    }
    a.packed = packed; a.rs = rs; a.R = R;
    a.sa = SampleArgs{ztab, t_rand, seed, offset, S, randomized ? 1 : 0};
    a.white = white;
    return TNERF_OK;
}

static int render_impl(const char* who, const tnerf_mlp_desc* d, const float* packed, const RaySource& rs, int64_t R, int32_t S,
                       const float* ztab, int32_t randomized, const float* t_rand, uint64_t seed, uint64_t offset, int32_t white,
                       float* comp, float* depth, float* acc, tnerf_stream_t stream);

extern "C" int tnerf_render_fused(const tnerf_mlp_desc* d, const float* packed, const float* rays_o, const float* rays_d,
                                  int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                                  uint64_t seed, uint64_t offset, int32_t white, float* comp, float* depth, float* acc,
                                  tnerf_stream_t stream) {
    return render_impl("tnerf_render_fused", d, packed, tn_table_source(rays_o, rays_d), R, S, ztab, randomized, t_rand, seed, offset,
                       white, comp, depth, acc, stream);
}

extern "C" int tnerf_render_fused_cam(const tnerf_mlp_desc* d, const float* packed, const tnerf_camera* cam, int64_t R, int32_t S,
                                      const float* ztab, int32_t randomized, const float* t_rand, uint64_t seed, uint64_t offset,
                                      int32_t white, float* comp, float* depth, float* acc, tnerf_stream_t stream) {
    RaySource rs;
    int rc = tn_camera_source("tnerf_render_fused_cam", cam, R, &rs); if (rc) return rc;
    return render_impl("tnerf_render_fused_cam", d, packed, rs, R, S, ztab, randomized, t_rand, seed, offset, white, comp, depth, acc, stream);
}

static int render_impl(const char* who, const tnerf_mlp_desc* d, const float* packed, const RaySource& rs, int64_t R, int32_t S,
                       const float* ztab, int32_t randomized, const float* t_rand, uint64_t seed, uint64_t offset, int32_t white,
                       float* comp, float* depth, float* acc, tnerf_stream_t stream) {
    FwdArgs a{};
    int rc = tn_fused_args(who, a, d, packed, rs, R, S, ztab, randomized, t_rand, seed, offset, white);
    if (rc) return rc;
    if (R == 0) return TNERF_OK;
    if (!comp) { tn_set_error("tnerf_render_fused: comp_rgb is NULL"); return TNERF_EINVAL; }
    a.comp = comp; a.depth = depth; a.acc = acc;
    return launch_fwd<true, false>(a, R, (hipStream_t)stream, who);
}

extern "C" int tnerf_train_fwd_fused(const tnerf_mlp_desc* d, const float* packed, const float* rays_o, const float* rays_d,
                                     int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                                     uint64_t seed, uint64_t offset, int32_t white, float* comp, float* stash, int64_t Mp,
                                     tnerf_stream_t stream) {
    FwdArgs a{};
    int rc = tn_fused_args("tnerf_train_fwd_fused", a, d, packed, tn_table_source(rays_o, rays_d), R, S, ztab, randomized, t_rand, seed, offset, white);
    if (rc) return rc;
    if (R == 0) return TNERF_OK;
    if (!comp || !stash || Mp < R * S) {
        tn_set_error("tnerf_train_fwd_fused: comp=%p stash=%p Mp=%lld < R*S=%lld", (void*)comp, (void*)stash, (long long)Mp, (long long)(R * S));
        return TNERF_EINVAL;
    }
    if (R == 0) return TNERF_OK;
    a.comp = comp; a.stash = stash; a.Mp = Mp;
    return launch_fwd<true, true>(a, R, (hipStream_t)stream, "tnerf_train_fwd_fused");
}

#ifdef TN_STAMPS
// Diagnostic build only (tools/stamp_probe.py): the fused forward kernel with s_memtime stamps per phase.
extern "C" int tnerf_debug_render_stamps(const tnerf_mlp_desc* d, const float* packed, const float* rays_o, const float* rays_d,
                                         int64_t R, int32_t S, const float* ztab, float* comp, float* stash, int64_t Mp,
                                         unsigned long long* stamps, tnerf_stream_t stream) {
    FwdArgs a{};
    int rc = tn_fused_args("tnerf_debug_render_stamps", a, d, packed, tn_table_source(rays_o, rays_d), R, S, ztab, 0, nullptr, 0, 0, 1);
    if (rc) return rc;
    a.comp = comp; a.stamps = stamps; a.stash = stash; a.Mp = Mp;
    return stash ? launch_fwd<true, true>(a, R, (hipStream_t)stream, "stamps") : launch_fwd<true, false>(a, R, (hipStream_t)stream, "stamps");
}
#endif
