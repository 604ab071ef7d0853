// bf16 mode: fused ray -> sample -> encode -> MLP (bf16 MFMA) -> composite (fp32), the body of render_one
// (reference src/train.py:46-56) for BASELINE cfg 4.  One persistent 512-thread workgroup per CU; a wavefront owns one
// ray at a time and marches it 32 samples per pass over the network (mlp16_core.hpp), the eight wavefronts of the
// workgroup sharing the LDS-resident weight stream.
#include "mlp16_core.hpp"
#include "mlp16_args.hpp"

// The network for one 32-sample tile: enc -> res[4] (r,g,b after sigmoid; sigma after ReLU; lane-half 0 only).
template <int HID, bool TRAIN>
__device__ __forceinline__ void tn16_mlp_tile(Pipe16& p, const unsigned char* lds, const Net16& n, int h,
                                              const bf16x8 (&enc)[TN16_KE], float (&res)[4], const Stash16& st, uint32_t sel_off) {
    constexpr int KH = HID / 16;
    const int depth = n.depth, skip_at = n.skip_at;
    const uint32_t vb0 = TN16_RING + 16u * h;                                   // + layer * HID * 4
    bf16x8 X[KH], Y[KH];
    f32x16 acc;
    if constexpr (TRAIN) {                                                      // the network input, as wgrad's B operand
        tn16_stash_tile(lds, sel_off, p.lane16, st, n.ft_enc, enc[0], enc[1], acc);
        tn16_stash_tile(lds, sel_off, p.lane16, st, n.ft_enc + 1, enc[2], enc[3], acc);
    }
    tn16_layer<HID, 0, TRAIN>(p, lds, vb0, X, enc, X, acc, st, sel_off, n.ft_h[0], 0);
    int l = 1;
    while (l < depth) {
        if (l == skip_at) tn16_layer<HID, 2, TRAIN>(p, lds, vb0 + l * HID * 4, X, enc, Y, acc, st, sel_off, n.ft_h[l], l);
        else              tn16_layer<HID, 1, TRAIN>(p, lds, vb0 + l * HID * 4, X, enc, Y, acc, st, sel_off, n.ft_h[l], l);
        if (++l >= depth) break;
        if (l == skip_at) tn16_layer<HID, 2, TRAIN>(p, lds, vb0 + l * HID * 4, Y, enc, X, acc, st, sel_off, n.ft_h[l], l);
        else              tn16_layer<HID, 1, TRAIN>(p, lds, vb0 + l * HID * 4, Y, enc, X, acc, st, sel_off, n.ft_h[l], l);
        ++l;
    }
    if ((depth - 1) & 1) tn16_layer<HID, 3, TRAIN>(p, lds, 0, Y, enc, Y, acc);
    else                 tn16_layer<HID, 3, TRAIN>(p, lds, 0, X, enc, X, acc);
    // heads: rows 0..2 = rgb.0 (sigmoid), row 3 = sigma.0 (ReLU)                                   nerf.py:39-40
    const f32x4 hb = *reinterpret_cast<const f32x4*>(lds + TN16_RING + depth * HID * 4);
#pragma unroll
    for (int i = 0; i < 3; ++i) res[i] = 1.0f / (1.0f + expf(-(acc[i] + hb[i])));
    res[3] = fmaxf(acc[3] + hb[3], 0.0f);
}

template <int HID, bool TRAIN>
__global__ __launch_bounds__(512, 2) void k_render16(Fwd16Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = tn_lane();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j = lane & 31, h = lane >> 5;
    RaySource rs = a.rs; SampleArgs sa = a.sa;
    if (TRAIN) tn_resolve_step(rs, sa);                    // dataset mode: this step's image and Philox counters
    const int S = sa.S, Lf = a.n.Lf;
    const uint32_t sel_off = TN16_SEL_OFF(a.n.n_bias);
    Pipe16 p;
    tn16_prologue(p, lds, a.packed, a.n, a.packed, a.n.n_stage, lane, wave, TRAIN);

    // Every wave of the workgroup runs the same number of network passes (the stage barriers are workgroup-wide):
    // rays beyond R are computed on a clamped index and stored nowhere (training: into the dump tile).
    const int64_t n_groups = (a.R + 7) / 8;
    const int TPR = (S + 31) / 32;
    for (int64_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const int64_t ray = g * 8 + wave;
        const bool rvalid = ray < a.R;
        const int64_t rayc = rvalid ? ray : a.R - 1;
        float dn;
        { float ro_[3], rd_[3]; tn_fetch_ray(rs, rayc, ro_, rd_); dn = tn_norm3(rd_[0], rd_[1], rd_[2]); }
        float T_in = 1.0f, cr = 0.f, cg = 0.f, cb = 0.f, cd = 0.f, ca = 0.f;
        // March the ray 32 samples per pass; every second pass (or the last one) the 64 lanes composite a segment:
        // lane l <- sample s0 + l.
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int sb = 0; sb < S; sb += 32) {
            tn_opaque_sources(rs, sa);
            {
                const int s = sb + j;
                const int sc = s < S ? s : S - 1;
                const float z = tn_depth(sa, rayc, sc);
                bf16x8 enc[TN16_KE];
                // (the ray is fetched again for every tile, behind a barrier the loop-invariant-code motion cannot cross: origin and direction
                //  are needed here only — six registers less held through the layer walk; as k_renderx3)
                int64_t rayt = rayc; asm volatile("" : "+s"(rayt));
                float ro_[3], rd_[3];
                tn_fetch_ray(rs, rayt, ro_, rd_);
                tn16_encode(tn_point(ro_[0], rd_[0], z), tn_point(ro_[1], rd_[1], z), tn_point(ro_[2], rd_[2], z), Lf, h, enc);
                Stash16 st{};
                int64_t tile = 0;
                if constexpr (TRAIN) {
                    tile = rvalid ? rayc * TPR + (sb >> 5) : a.n_tiles;
                    st.frag = a.stash + (tile * a.n.n_ft) * TN16_FT_BYTES;
                    st.mask = a.stash + TN16_STASH_FRAG_BYTES(a.n, a.n_tiles) + tile * (64 * (HID / 64) * 4);
                    st.mask_lstride = (a.n_tiles + 1) * (64 * (HID / 64) * 4);
                }
                float res[4];
                tn16_mlp_tile<HID, TRAIN>(p, lds, a.n, h, enc, res, st, sel_off);
                if constexpr (TRAIN) {
                    f32x4* o4 = reinterpret_cast<f32x4*>(a.stash + TN16_STASH_FRAG_BYTES(a.n, a.n_tiles) + TN16_STASH_MASK_BYTES(a.n, a.n_tiles)) + (tile * 32 + j);
                    const bool live = s < S;                                   // slots past S: zero outputs
                    const f32x4 r4 = {live ? res[0] : 0.f, live ? res[1] : 0.f, live ? res[2] : 0.f, live ? res[3] : 0.f};
                    if (h == 0) *o4 = r4;
                }
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
        }
        cr = tn_wave_sum(cr); cg = tn_wave_sum(cg); cb = tn_wave_sum(cb); cd = tn_wave_sum(cd); ca = tn_wave_sum(ca);
        if (lane == 0 && rvalid) {
            const float bg = a.white ? (1.0f - ca) : 0.0f;                                    // volume.py:42
            a.comp[3 * ray] = cr + bg; a.comp[3 * ray + 1] = cg + bg; a.comp[3 * ray + 2] = cb + bg;
            if (a.depth) a.depth[ray] = cd;
            if (a.acc) a.acc[ray] = ca;
            if (TRAIN && a.loss.ray_ws) tn_ray_loss(a.loss, rs, ray, cr + bg, cg + bg, cb + bg);      // train.py:122
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no DMA may still be writing this workgroup's LDS at exit
}

int tn16_launch_fwd(const Fwd16Args& a, bool train, hipStream_t stream, const char* who) {
    const int dev = tn_stream_device(stream), n_cu = tn_device_cus(dev);
    const int64_t groups = (a.R + 7) / 8;
    const dim3 grid((unsigned)(groups < n_cu ? groups : n_cu)), block(512);
    const size_t lds_bytes = TN16_SEL_OFF(a.n.n_bias) + 2048;
#define TN16_CASE(H_, T_)                                                                                                        \
    if (a.n.hidden == H_ && train == T_) {                                                                                        \
        static std::atomic<uint32_t> seen_[TN_MAX_DEVICES];                                                                       \
        if (int rc_ = tn_grant_dyn_lds(reinterpret_cast<const void*>(&k_render16<H_, T_>), lds_bytes, dev, seen_, who)) return rc_; \
        hipLaunchKernelGGL((k_render16<H_, T_>), grid, block, lds_bytes, stream, a);                                              \
        TN_HIP_CHECK_LAUNCH(who);                                                                                                 \
        return TNERF_OK;                                                                                                          \
    }
    TN16_CASE(256, false) TN16_CASE(256, true) TN16_CASE(128, false) TN16_CASE(128, true)
#undef TN16_CASE
    tn_set_error("%s: no bf16 kernel for hidden=%d", who, a.n.hidden);
    return TNERF_EUNSUPPORTED;
}

// ----------------------------------------------------------------------------------- entry points
int tn16_fused_args(const char* who, Fwd16Args& a, const tnerf_mlp_desc* d, const void* packed16, const RaySource& rs, int64_t R, int32_t S,
                    const float* ztab, int32_t randomized, const float* t_rand, uint64_t seed, uint64_t offset, int32_t white) {
    int rc = tn_build_net16(d, &a.n); if (rc) return rc;
    if (R < 0 || S < 1 || S > 4096 || (R > 0 && (!packed16 || !ztab || (!rs.c2w && (!rs.rays_o || !rs.rays_d))))) {
        tn_set_error("%s: R=%lld S=%d (1..4096) packed16=%p rays_o=%p rays_d=%p c2w=%p ztab=%p", who, (long long)R, S, packed16,
                     (const void*)rs.rays_o, (const void*)rs.rays_d, (const void*)rs.c2w, (const void*)ztab);
        return TNERF_EINVAL;
    }
    a.packed = static_cast<const unsigned char*>(packed16); a.rs = rs; a.R = R;
    a.sa = SampleArgs{ztab, t_rand, seed, offset, S, randomized ? 1 : 0};
    a.white = white;
    return TNERF_OK;
}

static int render16_impl(const char* who, const tnerf_mlp_desc* d, const void* packed16, const RaySource& rs, int64_t R, int32_t S,
                         const float* ztab, int32_t randomized, const float* t_rand, uint64_t seed, uint64_t offset, int32_t white,
                         float* comp, float* depth, float* acc, tnerf_stream_t stream) {
    Fwd16Args a{};
    int rc = tn16_fused_args(who, a, d, packed16, rs, R, S, ztab, randomized, t_rand, seed, offset, white); if (rc) return rc;
    if (R == 0) return TNERF_OK;
    if (!comp) { tn_set_error("%s: comp_rgb is NULL", who); return TNERF_EINVAL; }
    a.comp = comp; a.depth = depth; a.acc = acc;
    return tn16_launch_fwd(a, false, (hipStream_t)stream, who);
}

extern "C" int tnerf_render_fused_bf16(const tnerf_mlp_desc* d, const void* packed16, const float* rays_o, const float* rays_d,
                                       int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                                       uint64_t seed, uint64_t offset, int32_t white, float* comp, float* depth, float* acc,
                                       tnerf_stream_t stream) {
    return render16_impl("tnerf_render_fused_bf16", d, packed16, tn_table_source(rays_o, rays_d), R, S, ztab, randomized, t_rand, seed,
                         offset, white, comp, depth, acc, stream);
}

extern "C" int tnerf_render_fused_cam_bf16(const tnerf_mlp_desc* d, const void* packed16, const tnerf_camera* cam, int64_t R, int32_t S,
                                           const float* ztab, int32_t randomized, const float* t_rand, uint64_t seed, uint64_t offset,
                                           int32_t white, float* comp, float* depth, float* acc, tnerf_stream_t stream) {
    RaySource rs;
    int rc = tn_camera_source("tnerf_render_fused_cam_bf16", cam, R, &rs); if (rc) return rc;
    return render16_impl("tnerf_render_fused_cam_bf16", d, packed16, rs, R, S, ztab, randomized, t_rand, seed, offset, white, comp, depth,
                         acc, stream);
}

// ----------------------------------------------------------------------------------- packing
__global__ __launch_bounds__(256) void k_pack16(const float* __restrict__ params, const int32_t* __restrict__ table, int64_t n_w,
                                                int64_t n_all, unsigned short* __restrict__ out16, float* __restrict__ out32) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_all) return;
    const int32_t s = table[i];
    const float v = s >= 0 ? params[s] : 0.0f;
    if (i < n_w) out16[i] = __builtin_bit_cast(unsigned short, (__bf16)v);      // round to nearest even
    else out32[i - n_w] = v;
}

extern "C" int tnerf_mlp_pack_bf16(const tnerf_mlp_desc* d, const float* params, const int32_t* table, void* packed16,
                                   tnerf_stream_t stream) {
    Net16 n; int rc = tn_build_net16(d, &n); if (rc) return rc;
    if (!params || !table || !packed16) {
        tn_set_error("tnerf_mlp_pack_bf16: params=%p table=%p packed16=%p", (const void*)params, (const void*)table, packed16);
        return TNERF_EINVAL;
    }
    const int64_t n_w = (int64_t)(n.n_frag + n.n_bw_frag) * 512;
    hipLaunchKernelGGL(k_pack16, dim3((unsigned)((n.pack_entries + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, table, n_w,
                       n.pack_entries, static_cast<unsigned short*>(packed16),
                       reinterpret_cast<float*>(static_cast<unsigned char*>(packed16) + n.bias_off));
    TN_HIP_CHECK_LAUNCH("tnerf_mlp_pack_bf16");
    return TNERF_OK;
}
