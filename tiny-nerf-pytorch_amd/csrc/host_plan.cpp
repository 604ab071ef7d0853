// Host-side planning for libtnerf_hip.so: no GPU calls in this file.
//   * bit-exact depth tables of stratified_samples            [reference src/sampling.py:16-23]
//   * flat parameter layout of TinyNeRF                         [reference src/nerf.py:18-27]
//   * MFMA-fragment packing table, wgrad job table, slab->gradient gather table
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include "tnerf_internal.h"

static thread_local char g_err[512] = "ok";

extern "C" void tn_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int tnerf_version(void) { return TNERF_ABI_VERSION; }
extern "C" const char* tnerf_last_error_string(void) { return g_err; }
// Per-ray workspace of the fused train steps: dL/dcomp_rgb (3) and the squared error (1) of every ray, one 16-byte store per ray.
extern "C" int64_t tnerf_train_ws_floats(int64_t n_rays) {
    if (n_rays < 1) { tn_set_error("tnerf_train_ws_floats: n_rays=%lld", (long long)n_rays); return TNERF_EINVAL; }
    return TN_RAY_WS_FLOATS * n_rays;
}
// Capacity checks of caller-allocated workspaces whose required size depends on the library version (the layout of the per-ray
// workspace and of the stash are private to it): an undersized buffer is an error before anything is launched, not a write
// past its end.
extern "C" int tn_check_ray_ws(const char* who, int64_t n_rays, int64_t ws_floats) {
    if (ws_floats >= TN_RAY_WS_FLOATS * n_rays) return TNERF_OK;
    tn_set_error("%s: per-ray workspace of %lld floats, %lld rays need tnerf_train_ws_floats() = %lld", who, (long long)ws_floats,
                 (long long)n_rays, (long long)(TN_RAY_WS_FLOATS * n_rays));
    return TNERF_ESMALL;
}
extern "C" int tn_check_stash32(const char* who, const tnerf_mlp_desc* d, int64_t M, int64_t stride, int64_t capacity) {
    MlpLayout L; int rc = tn_build_layout(d, &L); if (rc) return rc;
    if (stride < M || (stride & 63)) { tn_set_error("%s: stash_row_stride=%lld for %lld samples (tnerf_plan_sizes.stash_row_stride: a multiple of 64 >= the samples)", who,
                                                     (long long)stride, (long long)M); return TNERF_EINVAL; }
    const int64_t need = TN_BOUND_OFF(L, stride) + TN_BOUND_FLOATS;                  // = tnerf_plan_sizes.stash_floats of `stride` samples
    if (capacity >= need) return TNERF_OK;
    tn_set_error("%s: stash of %lld floats, row stride %lld needs tnerf_plan_sizes.stash_floats = %lld", who, (long long)capacity, (long long)stride, (long long)need);
    return TNERF_ESMALL;
}
extern "C" int tn_check_stash16(const char* who, const tnerf_mlp_desc* d, int64_t n_rays, int32_t n_samples, int64_t capacity) {
    Net16 n; int rc = tn_build_net16(d, &n); if (rc) return rc;
    const int64_t tiles = n_rays * ((n_samples + 31) / 32);
    const int64_t need = TN16_STASH_FRAG_BYTES(n, tiles) + TN16_STASH_MASK_BYTES(n, tiles) + TN16_STASH_OUT_BYTES(n, tiles);
    if (capacity >= need) return TNERF_OK;
    tn_set_error("%s: bf16 stash of %lld bytes, %lld rays x %d samples need tnerf_bf16_train_plan.stash_bytes = %lld", who, (long long)capacity,
                 (long long)n_rays, n_samples, (long long)need);
    return TNERF_ESMALL;
}

// ---------------------------------------------------------------------------- depth tables
// torch.linspace(0,1,S) on CPU fp32 (ATen RangeFactories): step = (end-start)/(S-1) in fp32,
// element i = start + step*i for i < S/2 and end - step*(S-1-i) otherwise, each as ONE fused
// multiply-add.  z = near*(1-t) + far*t with every product/sum rounded separately.
extern "C" int tnerf_sample_tables(float near_, float far_, int32_t S, float* ztab, float* t_out) {
    if (S < 1 || !ztab) { tn_set_error("tnerf_sample_tables: n_samples=%d, ztab=%p", S, (void*)ztab); return TNERF_EINVAL; }
    volatile float nearf = near_, farf = far_;
    float* z = ztab; float* lo = ztab + S; float* hi = ztab + 2 * S;
    const float step = (S > 1) ? (1.0f - 0.0f) / (float)(S - 1) : 0.0f;
    const int half = S / 2;
    for (int i = 0; i < S; ++i) {
        float t;
        if (S == 1) t = 0.0f;
        else if (i < half) t = fmaf(step, (float)i, 0.0f);
        else t = fmaf(-step, (float)(S - 1 - i), 1.0f);
        if (t_out) t_out[i] = t;
        volatile float om = 1.0f - t;
        volatile float a = nearf * om;
        volatile float b = farf * t;
        volatile float zz = a + b;
        z[i] = zz;
    }
    for (int i = 0; i < S; ++i) {
        if (i == 0) lo[i] = z[0];
        else { volatile float s = z[i - 1] + z[i]; volatile float m = 0.5f * s; lo[i] = m; }
        if (i == S - 1) hi[i] = z[S - 1];
        else { volatile float s = z[i] + z[i + 1]; volatile float m = 0.5f * s; hi[i] = m; }
    }
    return TNERF_OK;
}

// ------------------------------------------------------------------------------ model layout
static int check_desc(const tnerf_mlp_desc* d) {
    if (!d) { tn_set_error("NULL tnerf_mlp_desc"); return TNERF_EINVAL; }
    if (d->hidden < 1 || d->hidden > 256) {
        // any width up to 256 runs on the 128- or 256-wide kernels with zero-padded weights (the padding units stay exactly 0
        // through bias 0 + ReLU and receive no gradient that is ever read); wider layers need kernels that are not built
        tn_set_error("hidden=%d: widths 1..256 are built", d->hidden); return TNERF_EUNSUPPORTED; }
    if (d->depth < 1 || d->depth > TN_MAXD) { tn_set_error("depth=%d out of [1,%d]", d->depth, TN_MAXD); return TNERF_EUNSUPPORTED; }
    if (d->in_dim < 1 || d->in_dim > 64) { tn_set_error("in_dim=%d out of [1,64]", d->in_dim); return TNERF_EUNSUPPORTED; }
    if (d->skip_at < 0 || d->skip_at >= d->depth) {
        // skip_at == depth would feed hidden+in_dim features to Linear(hidden, .) heads: the reference raises there too
        tn_set_error("skip_at=%d must be 0 (none) or in [1, depth-1]", d->skip_at); return TNERF_EINVAL; }
    return TNERF_OK;
}

extern "C" int tnerf_input_pairing(int32_t in_dim, int32_t* emap, int32_t* n_steps) {
    if (in_dim < 1 || in_dim > 64 || !emap || !n_steps) { tn_set_error("tnerf_input_pairing: bad args"); return TNERF_EINVAL; }
    for (int i = 0; i < 64; ++i) emap[i] = -1;
    int used;
    if (in_dim >= 9 && (in_dim - 3) % 6 == 0) {          // [x, sin(2^k x), cos(2^k x)]: pair sin with cos
        const int L = (in_dim - 3) / 6;
        for (int k = 0; k < L; ++k)
            for (int c = 0; c < 3; ++c) {
                emap[2 * (3 * k + c) + 0] = 3 + 6 * k + c;
                emap[2 * (3 * k + c) + 1] = 3 + 6 * k + 3 + c;
            }
        emap[2 * (3 * L) + 0] = 0; emap[2 * (3 * L) + 1] = 1;
        emap[2 * (3 * L + 1) + 0] = 2;
        used = 3 * L + 2;
    } else {
        for (int c = 0; c < in_dim; ++c) emap[c] = c;
        used = (in_dim + 1) / 2;
    }
    *n_steps = used <= 20 ? 20 : 32;
    return TNERF_OK;
}

static int64_t true_param_count(const tnerf_mlp_desc* d) {
    int64_t n = 0; int fan = d->in_dim;
    for (int l = 0; l < d->depth; ++l) {
        n += (int64_t)d->hidden * fan + d->hidden;
        fan = (d->skip_at > 0 && l == d->skip_at - 1) ? d->hidden + d->in_dim : d->hidden;
    }
    return n + 4 * (int64_t)d->hidden + 4;
}

extern "C" int tn_build_layout(const tnerf_mlp_desc* d_true, MlpLayout* L) {
    int rc = check_desc(d_true); if (rc) return rc;
    tnerf_mlp_desc padded = *d_true;
    padded.hidden = d_true->hidden <= 128 ? 128 : 256;                 // the kernel width
    const tnerf_mlp_desc* d = &padded;
    memset(L, 0, sizeof(*L));
    L->in_dim = d->in_dim; L->hidden = d->hidden; L->depth = d->depth; L->skip_at = d->skip_at; L->flags = d->flags;
    L->hidden_true = d_true->hidden;
    L->NT = d->hidden / 32;
    int32_t em[64], ne;
    rc = tnerf_input_pairing(d->in_dim, em, &ne); if (rc) return rc;
    L->NE = ne;
    for (int s = 0; s < TN_MAX_STEPS; ++s) { L->emap[s][0] = (int16_t)em[2 * s]; L->emap[s][1] = (int16_t)em[2 * s + 1]; }
    const int NT = L->NT, NE = L->NE, H = d->hidden;
    // flat parameters, state_dict order
    int64_t off = 0; int fan = d->in_dim;
    for (int l = 0; l < d->depth; ++l) {
        L->fan_in[l] = fan; L->p_w[l] = off; off += (int64_t)H * fan; L->p_b[l] = off; off += H;
        fan = (d->skip_at > 0 && l == d->skip_at - 1) ? H + d->in_dim : H;
    }
    L->p_ws = off; off += H; L->p_bs = off; off += 1; L->p_wc = off; off += 3 * (int64_t)H; L->p_bc = off; off += 3;
    L->n_params_padded = off;
    L->n_params = true_param_count(d_true);
    // packed segments
    int64_t po = 0;
    for (int l = 0; l < d->depth; ++l) {
        L->fw_bias[l] = po; po += NT * 32;
        const bool enc = (l == 0) || (d->skip_at > 0 && l == d->skip_at);
        L->fw_enc[l] = enc ? po : -1; if (enc) po += (int64_t)NT * NE * 64;
        L->fw_hid[l] = (l > 0) ? po : -1; if (l > 0) po += (int64_t)NT * NT * 1024;
    }
    L->fw_head_bias = po; po += 32;
    L->fw_head = po; po += (int64_t)NT * 1024;
    for (int l = 0; l < d->depth; ++l) { L->bw_hid[l] = (l > 0) ? po : -1; if (l > 0) po += (int64_t)NT * NT * 1024; }
    L->bw_head = po; po += (int64_t)NT * 256;
    L->packed_floats = po;
    // stash rows
    int row = 0;
    L->enc_row0 = row; row += 2 * NE;
    for (int l = 0; l < d->depth; ++l) { L->h_row0[l] = row; row += H; }
    L->out_row0 = row; row += 4;
    for (int l = 0; l < d->depth; ++l) { L->dz_row0[l] = row; row += H; }
    L->dzh_row0 = row; row += 4;
    L->stash_rows = row;
    return TNERF_OK;
}

// Padded parameter space (what the table builders below index) <-> the caller's flat parameters (state_dict order, true
// width): to_true[padded index] = true index or -1 (padding).  Identity when hidden is 128 or 256.
static void param_maps(const MlpLayout& L, std::vector<int32_t>& to_true, std::vector<int32_t>* to_padded) {
    const int Hk = L.hidden, Ht = L.hidden_true;
    to_true.assign((size_t)L.n_params_padded, -1);
    if (to_padded) to_padded->assign((size_t)L.n_params, -1);
    int64_t t = 0;
    auto link = [&](int64_t pp) { to_true[(size_t)pp] = (int32_t)t; if (to_padded) (*to_padded)[(size_t)t] = (int32_t)pp; ++t; };
    for (int l = 0; l < L.depth; ++l) {
        const int fan_k = L.fan_in[l];                                   // padded fan-in: in_dim | Hk | Hk + in_dim
        for (int n = 0; n < Ht; ++n)
            for (int k = 0; k < fan_k; ++k) {
                if (l > 0 && k >= Ht && k < Hk) continue;                // padding columns of the hidden part
                link(L.p_w[l] + (int64_t)n * fan_k + k);
            }
        for (int n = 0; n < Ht; ++n) link(L.p_b[l] + n);
    }
    for (int k = 0; k < Ht; ++k) link(L.p_ws + k);
    link(L.p_bs);
    for (int n = 0; n < 3; ++n) for (int k = 0; k < Ht; ++k) link(L.p_wc + (int64_t)n * Hk + k);
    for (int n = 0; n < 3; ++n) link(L.p_bc + n);
}
// A reduce table built in the padded space -> the caller's table (header + 2 ints per TRUE parameter)
static void reduce_to_true(const MlpLayout& L, const int32_t* padded, int32_t* out) {
    std::vector<int32_t> to_true, to_padded; param_maps(L, to_true, &to_padded);
    memcpy(out, padded, sizeof(int32_t) * TN_RED_HDR);
    for (int64_t t = 0; t < L.n_params; ++t) {
        out[TN_RED_HDR + 2 * t] = padded[TN_RED_HDR + 2 * (int64_t)to_padded[(size_t)t]];
        out[TN_RED_HDR + 2 * t + 1] = padded[TN_RED_HDR + 2 * (int64_t)to_padded[(size_t)t] + 1];
    }
}
static void remap_sources(const MlpLayout& L, int32_t* table, int64_t n) {      // pack-table entries: padded -> true index
    if (L.hidden_true == L.hidden) return;
    std::vector<int32_t> to_true; param_maps(L, to_true, nullptr);
    for (int64_t i = 0; i < n; ++i) if (table[i] >= 0) table[i] = to_true[(size_t)table[i]];
}

extern "C" int64_t tnerf_param_count(const tnerf_mlp_desc* d) {
    MlpLayout L; if (tn_build_layout(d, &L)) return -1;
    return L.n_params;
}

extern "C" int tnerf_param_layout(const tnerf_mlp_desc* d, int64_t* offsets, int64_t* rows, int64_t* cols) {
    MlpLayout L; int rc = tn_build_layout(d, &L); if (rc) return rc;
    if (!offsets || !rows || !cols) { tn_set_error("tnerf_param_layout: NULL output"); return TNERF_EINVAL; }
    int k = 0; int64_t off = 0; int fan = d->in_dim;                    // the caller's (true-width) flat layout, state_dict order
    auto put = [&](int64_t r, int64_t c) { offsets[k] = off; rows[k] = r; cols[k] = c; off += r * c; ++k; };
    for (int l = 0; l < d->depth; ++l) {
        put(d->hidden, fan); put(d->hidden, 1);
        fan = (d->skip_at > 0 && l == d->skip_at - 1) ? d->hidden + d->in_dim : d->hidden;
    }
    put(1, d->hidden); put(1, 1); put(3, d->hidden); put(3, 1);
    return TNERF_OK;
}

// ------------------------------------------------------------------------------- wgrad plan
// Cycles of a workgroup of the x3 weight-gradient kernel per 32-sample block = TN_X3_TILE x (tiles per wave) + TN_X3_FIXED, fitted to the
// per-workgroup stamps of the WHOLE kernel running (tools/wgrad_x3_probe.py on a -DTN_STAMPS build, 4096 x 64 samples, 8x256): 4750 /
// 2740 / 2394 cycles per block for 8 / 2 / 1 tiles per wave.  (Round 3's first fit, 650 / 2500, was taken class by class and gave the
// bandwidth-hungry input / skip / head classes too few workgroups: they ran 15 % longer than the 256 x 256 classes.)
#ifndef TN_X3_TILE
#define TN_X3_TILE 335.0
#define TN_X3_FIXED 2070.0
#endif
namespace {
struct JobClass {
    int a_row0, a_rows, b_row0, b_rows, n_at, n_bt, wa, has_bias;
    int a_bound, b_bound;  // TNB_* index of the rows' magnitude bound
    int cost;              // max 32x32 tiles per wave
    int chunks;            // workgroups
    int64_t slab0;         // float offset of chunk 0
    int64_t slab_stride;   // floats per chunk
};

// The 8 waves of a wgrad workgroup form a WA x (8/WA) grid over the (n_at x n_bt) tile grid; every wave gets
// ta x tb tiles (ta = ceil(n_at/WA), ...) or, if its origin is outside the grid, nothing.  Only the (ta,tb) shapes
// instantiated in wgrad.hip are eligible, and partially filled waves are not allowed.
static bool split_ok(int n_at, int n_bt, int wa, int* cost, int* ld) {
    const int wb = 8 / wa;
    const int ta = (n_at + wa - 1) / wa, tb = (n_bt + wb - 1) / wb;
    const bool shape = (ta == 2 && tb == 4) || (ta == 1 && tb == 2) || (ta == 2 && tb == 1) || (ta == 1 && tb == 1);
    if (!shape || n_at % ta != 0 || n_bt % tb != 0) return false;
    *cost = ta * tb; *ld = ta + tb;
    return true;
}
static void pick_split(int n_at, int n_bt, int* wa_out, int* cost_out) {
    int best_wa = 0, best_cost = 1 << 30, best_ld = 1 << 30;
    for (int wa = 1; wa <= 8; wa *= 2) {
        int cost, ld;
        if (!split_ok(n_at, n_bt, wa, &cost, &ld)) continue;
        if (cost < best_cost || (cost == best_cost && ld < best_ld)) { best_cost = cost; best_ld = ld; best_wa = wa; }
    }
    *wa_out = best_wa; *cost_out = best_cost;
}

static int build_classes(const MlpLayout& L, int64_t M, int n_cu, std::vector<JobClass>& cls, int64_t* slab_total) {
    const int H = L.hidden, NT = L.NT, encT = (2 * L.NE + 31) / 32;
    bool bad_split = false;
    auto add = [&](int a0, int ar, int b0, int br, int nat, int nbt, int bias, int ab, int bb) {
        JobClass c{}; c.a_row0 = a0; c.a_rows = ar; c.b_row0 = b0; c.b_rows = br; c.n_at = nat; c.n_bt = nbt; c.has_bias = bias;
        c.a_bound = ab; c.b_bound = bb;
        pick_split(nat, nbt, &c.wa, &c.cost); cls.push_back(c);
        if (c.wa == 0) bad_split = true;
    };
    // class index order is what reduce_table refers to:
    //   [l]            l>=1 : (dZ_l, H_{l-1})     owns b_l
    //   [0]                 : (dZ_0, ENC)         owns b_0
    //   [depth]             : (dZ_skip, ENC)      (only if skip)
    //   [last]              : (dZ_head, H_{depth-1}) owns head biases
    add(L.dz_row0[0], H, L.enc_row0, 2 * L.NE, NT, encT, 1, TNB_DZ(0), TNB_ENC);
    for (int l = 1; l < L.depth; ++l) add(L.dz_row0[l], H, L.h_row0[l - 1], H, NT, NT, 1, TNB_DZ(l), TNB_H(l - 1));
    if (L.skip_at > 0) add(L.dz_row0[L.skip_at], H, L.enc_row0, 2 * L.NE, NT, encT, 0, TNB_DZ(L.skip_at), TNB_ENC);
    add(L.dzh_row0, 4, L.h_row0[L.depth - 1], H, 1, NT, 1, TNB_DZH, TNB_H(L.depth - 1));
    if (bad_split) { tn_set_error("wgrad: no kernel for this layer-shape / wave split"); return TNERF_EUNSUPPORTED; }
    if ((int)cls.size() > TN_RED_MAXCLS) { tn_set_error("too many wgrad job classes"); return TNERF_EUNSUPPORTED; }
    const int64_t MB = (M + 31) / 32;
    // One workgroup per CU and never more (a 257th workgroup would run alone in a second round and double
    // the kernel time): start with one chunk per class, then keep splitting the class whose workgroups are
    // the longest (cost per sample x samples per chunk) until every CU has one.
    for (auto& c : cls) c.chunks = 1;
    int total = (int)cls.size();
    while (total < n_cu) {
        int best = -1; double best_t = 0.0;
        for (size_t i = 0; i < cls.size(); ++i) {
            if (cls[i].chunks >= MB) continue;
            // cycles per 32-sample block, measured (tools/wgrad_probe.py, LDS-DMA staging): 17.86k / 5.2k / 3.2k for
            // 8 / 2 / 1 tiles per wave, i.e. ~2107 per tile (16 MFMA steps of 64 cycles, two waves per SIMD) + ~1000 fixed
            // The split-bf16 kernel (default pipe): ~TN_X3_TILE per tile + TN_X3_FIXED (tools/wgrad_x3_probe.py, -DTN_STAMPS build)
            // Jobs of <= 256 combined rows run the kernel's two-item converter (wgrad.hip NI = 2) and cost by the bytes they stream:
            // 2345 / 1754 / 1582 cycles per block for 256 / 192 / 160 rows (4x128 network, the same stamps) = ~9.2 per row.
            const int rows = (cls[i].n_at + cls[i].n_bt) * 32;
            const double per_block = (L.flags & TNERF_FLAG_FP32_MFMA) ? (double)cls[i].cost * 2107.0 + 1000.0
                                   : rows <= 256                     ? 9.2 * (double)rows
                                                                      : (double)cls[i].cost * TN_X3_TILE + TN_X3_FIXED;
            const double t = per_block * (double)((MB + cls[i].chunks - 1) / cls[i].chunks);
            if (t > best_t) { best_t = t; best = (int)i; }
        }
        if (best < 0) break;
        ++cls[best].chunks; ++total;
    }
    int64_t off = 0;
    for (auto& c : cls) {
        // make every chunk non-empty
        const int64_t per = (MB + c.chunks - 1) / c.chunks;
        c.chunks = (int)((MB + per - 1) / per);
        c.slab_stride = (int64_t)c.n_at * 32 * (c.n_bt * 32) + (int64_t)c.n_at * 32;
        c.slab0 = off; off += c.slab_stride * c.chunks;
    }
    if (off >= (int64_t)1 << 31) { tn_set_error("slab workspace exceeds int32 offsets"); return TNERF_EUNSUPPORTED; }
    *slab_total = off;
    return TNERF_OK;
}
}  // namespace

extern "C" int tnerf_plan_sizes_query(const tnerf_mlp_desc* d, int64_t M, int32_t n_cu, tnerf_plan_sizes* out) {
    MlpLayout L; int rc = tn_build_layout(d, &L); if (rc) return rc;
    if (!out || M < 1 || n_cu < 1) { tn_set_error("tnerf_plan_sizes_query: M=%lld n_cu=%d", (long long)M, n_cu); return TNERF_EINVAL; }
    std::vector<JobClass> cls; int64_t slab = 0;
    rc = build_classes(L, M, n_cu, cls, &slab); if (rc) return rc;
    int64_t jobs = 0; for (auto& c : cls) jobs += c.chunks;
    const int64_t Mp = (M + 63) / 64 * 64;
    out->n_params = L.n_params;
    out->packed_floats = L.packed_floats;
    out->stash_floats = TN_BOUND_OFF(L, Mp) + TN_BOUND_FLOATS;
    out->slab_floats = slab;
    out->job_ints = jobs * TN_JOB_INTS;
    out->reduce_ints = TN_RED_HDR + 2 * L.n_params;
    out->n_jobs = jobs;
    out->stash_row_stride = Mp;
    return TNERF_OK;
}

static void fill_pack_table(const MlpLayout& L, int32_t* T) {
    const int NT = L.NT, NE = L.NE, H = L.hidden;
    for (int64_t i = 0; i < L.packed_floats; ++i) T[i] = -1;
    auto head_w = [&](int n, int k) -> int32_t {         // head row order: r,g,b,sigma
        if (n < 3) return (int32_t)(L.p_wc + (int64_t)n * H + k);
        if (n == 3) return (int32_t)(L.p_ws + k);
        return -1;
    };
    auto head_b = [&](int n) -> int32_t { return n < 3 ? (int32_t)(L.p_bc + n) : (n == 3 ? (int32_t)L.p_bs : -1); };
    for (int l = 0; l < L.depth; ++l) {
        const int fan = L.fan_in[l];
        for (int t = 0; t < NT; ++t)
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < 16; ++r)
                    T[L.fw_bias[l] + (t * 2 + h) * 16 + r] = (int32_t)(L.p_b[l] + 32 * t + TN_ACC_ROW(r, h));
        if (L.fw_enc[l] >= 0) {
            const int coloff = (l == 0) ? 0 : H;
            for (int t = 0; t < NT; ++t)
                for (int g = 0; g < NE / 4; ++g)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int p = 0; p < 4; ++p) {
                            const int c = L.emap[4 * g + p][lane >> 5];
                            if (c < 0) continue;
                            T[L.fw_enc[l] + (((int64_t)t * (NE / 4) + g) * 64 + lane) * 4 + p] =
                                (int32_t)(L.p_w[l] + (int64_t)(32 * t + (lane & 31)) * fan + coloff + c);
                        }
        }
        if (L.fw_hid[l] >= 0) {
            for (int t = 0; t < NT; ++t) for (int kb = 0; kb < NT; ++kb) for (int q = 0; q < 4; ++q)
                for (int lane = 0; lane < 64; ++lane) for (int p = 0; p < 4; ++p) {
                    const int n = 32 * t + (lane & 31), k = 32 * kb + 8 * q + 4 * (lane >> 5) + p;
                    T[L.fw_hid[l] + ((((int64_t)t * NT + kb) * 4 + q) * 64 + lane) * 4 + p] = (int32_t)(L.p_w[l] + (int64_t)n * fan + k);
                    // transposed copy for dgrad: out row = k-tile index (t here), reduction over n-block (kb here)
                    const int kk = 32 * t + (lane & 31), nn = 32 * kb + 8 * q + 4 * (lane >> 5) + p;
                    T[L.bw_hid[l] + ((((int64_t)t * NT + kb) * 4 + q) * 64 + lane) * 4 + p] = (int32_t)(L.p_w[l] + (int64_t)nn * fan + kk);
                }
        }
    }
    for (int h = 0; h < 2; ++h) for (int r = 0; r < 16; ++r) T[L.fw_head_bias + h * 16 + r] = head_b(TN_ACC_ROW(r, h));
    for (int kb = 0; kb < NT; ++kb) for (int q = 0; q < 4; ++q) for (int lane = 0; lane < 64; ++lane) for (int p = 0; p < 4; ++p)
        T[L.fw_head + (((int64_t)kb * 4 + q) * 64 + lane) * 4 + p] = head_w(lane & 31, 32 * kb + 8 * q + 4 * (lane >> 5) + p);
    for (int kt = 0; kt < NT; ++kt) for (int lane = 0; lane < 64; ++lane) for (int p = 0; p < 4; ++p)
        T[L.bw_head + ((int64_t)kt * 64 + lane) * 4 + p] = head_w(4 * (lane >> 5) + p, 32 * kt + (lane & 31));
}

extern "C" int tnerf_plan_fill(const tnerf_mlp_desc* d, int64_t M, int32_t n_cu,
                               int32_t* pack_table, int32_t* job_table, int32_t* reduce_table) {
    MlpLayout L; int rc = tn_build_layout(d, &L); if (rc) return rc;
    if (M < 1 || n_cu < 1) { tn_set_error("tnerf_plan_fill: M=%lld n_cu=%d", (long long)M, n_cu); return TNERF_EINVAL; }
    std::vector<JobClass> cls; int64_t slab = 0;
    rc = build_classes(L, M, n_cu, cls, &slab); if (rc) return rc;
    if (pack_table) { fill_pack_table(L, pack_table); remap_sources(L, pack_table, L.packed_floats); }
    const int64_t MB = (M + 31) / 32;
    std::vector<int32_t> red_padded;                                     // the table below is built in the padded parameter space
    int32_t* const reduce_out = reduce_table;
    if (reduce_table && L.hidden_true != L.hidden) { red_padded.assign((size_t)(TN_RED_HDR + 2 * L.n_params_padded), 0); reduce_table = red_padded.data(); }
    if (job_table) {
        // interleave classes so that heavy and light workgroups are spread over the dispatch order
        int64_t j = 0;
        int maxch = 0; for (auto& c : cls) maxch = std::max(maxch, c.chunks);
        for (int ch = 0; ch < maxch; ++ch)
            for (size_t ci = 0; ci < cls.size(); ++ci) {
                const JobClass& c = cls[ci];
                if (ch >= c.chunks) continue;
                const int64_t per = (MB + c.chunks - 1) / c.chunks;
                int32_t* r = job_table + j * TN_JOB_INTS;
                memset(r, 0, sizeof(int32_t) * TN_JOB_INTS);
                r[JOB_A_ROW0] = c.a_row0; r[JOB_A_ROWS] = c.a_rows; r[JOB_B_ROW0] = c.b_row0; r[JOB_B_ROWS] = c.b_rows;
                r[JOB_N_AT] = c.n_at; r[JOB_N_BT] = c.n_bt; r[JOB_WA] = c.wa;
                r[JOB_MBLK0] = (int32_t)(ch * per);
                r[JOB_MBLKN] = (int32_t)std::max<int64_t>(0, std::min<int64_t>(per, MB - ch * per));
                r[JOB_SLAB_OFF] = (int32_t)(c.slab0 + c.slab_stride * ch);
                r[JOB_CLASS] = (int32_t)ci; r[JOB_HAS_BIAS] = c.has_bias;
                r[JOB_A_BOUND] = c.a_bound; r[JOB_B_BOUND] = c.b_bound;
                ++j;
            }
    }
    if (reduce_table) {
        int32_t* hdr = reduce_table;
        memset(hdr, 0, sizeof(int32_t) * TN_RED_HDR);
        hdr[0] = (int32_t)cls.size();
        for (size_t ci = 0; ci < cls.size(); ++ci) {
            hdr[1 + 4 * ci + 0] = (int32_t)cls[ci].slab0;
            hdr[1 + 4 * ci + 1] = (int32_t)cls[ci].slab_stride;
            hdr[1 + 4 * ci + 2] = cls[ci].chunks;
        }
        int32_t* E = reduce_table + TN_RED_HDR;
        auto put = [&](int64_t param, int cls_id, int64_t elem) { E[2 * param] = (int32_t)elem; E[2 * param + 1] = cls_id; };
        const int H = L.hidden;
        const int cls_skip_enc = L.skip_at > 0 ? L.depth : -1;
        const int cls_head = (int)cls.size() - 1;
        // inverse of the input pairing: input column -> stash ENC row
        int inv[64]; for (int c = 0; c < 64; ++c) inv[c] = -1;
        for (int s = 0; s < L.NE; ++s) for (int h = 0; h < 2; ++h) if (L.emap[s][h] >= 0) inv[L.emap[s][h]] = 2 * s + h;
        for (int l = 0; l < L.depth; ++l) {
            const int fan = L.fan_in[l];
            const JobClass& cm = cls[l];               // class l: (dZ_l, input of layer l) main part
            const int ldm = cm.n_bt * 32;
            for (int n = 0; n < H; ++n) {
                for (int k = 0; k < fan; ++k) {
                    const int64_t pi = L.p_w[l] + (int64_t)n * fan + k;
                    if (l == 0) put(pi, 0, (int64_t)n * ldm + inv[k]);
                    else if (k < H) put(pi, l, (int64_t)n * ldm + k);
                    else put(pi, cls_skip_enc, (int64_t)n * (cls[cls_skip_enc].n_bt * 32) + inv[k - H]);
                }
                put(L.p_b[l] + n, l, (int64_t)cm.n_at * 32 * ldm + n);
            }
        }
        const JobClass& chd = cls[cls_head];
        const int ldh = chd.n_bt * 32;
        for (int k = 0; k < H; ++k) {
            put(L.p_ws + k, cls_head, (int64_t)3 * ldh + k);
            for (int n = 0; n < 3; ++n) put(L.p_wc + (int64_t)n * H + k, cls_head, (int64_t)n * ldh + k);
        }
        put(L.p_bs, cls_head, (int64_t)32 * ldh + 3);
        for (int n = 0; n < 3; ++n) put(L.p_bc + n, cls_head, (int64_t)32 * ldh + n);
        if (reduce_table != reduce_out) reduce_to_true(L, reduce_table, reduce_out);
    }
    return TNERF_OK;
}

// ------------------------------------------------------------------------------------ bf16 mode
extern "C" int tn_build_net16(const tnerf_mlp_desc* d, Net16* n) {
    int rc = check_desc(d); if (rc) return rc;
    if (d->in_dim < 9 || (d->in_dim - 3) % 6 != 0) {
        tn_set_error("bf16 mode: only the fused paths are built and they need in_dim = 6L+3; got %d", d->in_dim);
        return TNERF_EUNSUPPORTED;
    }
    memset(n, 0, sizeof(*n));
    tnerf_mlp_desc padded = *d;
    padded.hidden = d->hidden <= 128 ? 128 : 256;                       // the kernel width (zero-padded weights, see check_desc)
    d = &padded;
    n->in_dim = d->in_dim; n->hidden = d->hidden; n->depth = d->depth; n->skip_at = d->skip_at;
    n->Lf = (d->in_dim - 3) / 6;
    const int NT = d->hidden / 32, KH = d->hidden / 16;
    int frags = NT * TN16_KE;
    for (int l = 1; l < d->depth; ++l) frags += NT * (KH + ((d->skip_at > 0 && l == d->skip_at) ? TN16_KE : 0));
    frags += TN16_STAGE;                                     // heads: KH (<= 16) fragments padded to one stage
    const int bw = TN16_STAGE + (d->depth - 1) * NT * KH;    // heads^T: NT (<= 16) fragments padded to one stage
    if (frags % TN16_STAGE != 0 || bw % TN16_STAGE != 0) { tn_set_error("bf16 mode: fragment stream is not a whole number of stages"); return TNERF_EUNSUPPORTED; }
    n->n_frag = frags; n->n_stage = frags / TN16_STAGE;
    n->n_bw_frag = bw; n->n_bw_stage = bw / TN16_STAGE;
    n->bias_off = (frags + bw) * 1024;
    n->n_bias = d->depth * d->hidden + 4;
    n->packed_bytes = (int64_t)n->bias_off + (int64_t)n->n_bias * 4;
    n->pack_entries = (int64_t)(frags + bw) * 512 + n->n_bias;
    int ft = 0;
    n->ft_enc = ft; ft += 2;
    for (int l = 0; l < d->depth; ++l) { n->ft_h[l] = ft; ft += NT; }
    for (int l = 0; l < d->depth; ++l) { n->ft_dz[l] = ft; ft += NT; }
    n->ft_dzh = ft; ft += 1;
    n->n_ft = ft;
    return TNERF_OK;
}

extern "C" int tnerf_bf16_plan_sizes(const tnerf_mlp_desc* d, tnerf_bf16_sizes* out) {
    Net16 n; int rc = tn_build_net16(d, &n); if (rc) return rc;
    if (!out) { tn_set_error("tnerf_bf16_plan_sizes: NULL output"); return TNERF_EINVAL; }
    out->packed_bytes = n.packed_bytes; out->pack_entries = n.pack_entries;
    out->n_fragments = n.n_frag + n.n_bw_frag; out->bias_offset_bytes = n.bias_off;
    out->n_fwd_fragments = n.n_frag;
    return TNERF_OK;
}

namespace {
inline int hid_feature16(int s, int h, int e) { return 32 * (s >> 1) + TN_ACC_ROW(8 * (s & 1) + e, h); }
inline int enc_column16(int Lf, int u, int h, int e) {
    const int a = 8 * u + e;
    if (a < 3 * Lf) return 3 + 6 * (a / 3) + (a % 3) + 3 * h;        // encoding.py:27-33 column order
    if (a == 3 * Lf) return h;                                        // x | y
    if (a == 3 * Lf + 1) return h ? -1 : 2;                           // z | 0
    return -1;
}
}  // namespace

extern "C" int tnerf_bf16_pack_table(const tnerf_mlp_desc* d, int32_t* T) {
    Net16 n; int rc = tn_build_net16(d, &n); if (rc) return rc;
    MlpLayout L; rc = tn_build_layout(d, &L); if (rc) return rc;
    if (!T) { tn_set_error("tnerf_bf16_pack_table: NULL table"); return TNERF_EINVAL; }
    for (int64_t i = 0; i < n.pack_entries; ++i) T[i] = -1;
    const int H = L.hidden, NT = H / 32, KH = H / 16, Lf = n.Lf;
    int64_t f = 0;                                                        // running fragment index
    auto put = [&](int lane, int e, int64_t src) { T[(f * 64 + lane) * 8 + e] = (int32_t)src; };
    auto head_w = [&](int row, int k) -> int64_t { return row < 3 ? L.p_wc + (int64_t)row * H + k : (row == 3 ? L.p_ws + k : -1); };
    // ---- forward stream
    for (int l = 0; l < L.depth; ++l) {
        const int fan = L.fan_in[l];
        const bool skip = L.skip_at > 0 && l == L.skip_at;
        for (int t = 0; t < NT; ++t) {
            if (l > 0)
                for (int s = 0; s < KH; ++s, ++f)
                    for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e)
                        put(lane, e, L.p_w[l] + (int64_t)(32 * t + (lane & 31)) * fan + hid_feature16(s, lane >> 5, e));
            if (l == 0 || skip)
                for (int u = 0; u < TN16_KE; ++u, ++f)
                    for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e) {
                        const int c = enc_column16(Lf, u, lane >> 5, e);
                        if (c >= 0) put(lane, e, L.p_w[l] + (int64_t)(32 * t + (lane & 31)) * fan + (l == 0 ? 0 : H) + c);
                    }
        }
    }
    for (int s = 0; s < KH; ++s, ++f)                                     // head tile: rows r,g,b (rgb.0) and sigma.0
        for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e) {
            const int64_t src = head_w(lane & 31, hid_feature16(s, lane >> 5, e));
            if (src >= 0) put(lane, e, src);
        }
    f = n.n_frag;
    // ---- backward stream
    for (int t = 0; t < NT; ++t, ++f)                                     // heads^T: dH[k] = sum_{row<4} W_head[row][k] dZh[row]
        for (int lane = 0; lane < 32; ++lane) for (int e = 0; e < 4; ++e) put(lane, e, head_w(e, 32 * t + lane));
    f = n.n_frag + TN16_STAGE;
    for (int l = L.depth - 1; l >= 1; --l) {
        const int fan = L.fan_in[l];
        for (int t = 0; t < NT; ++t)
            for (int s = 0; s < KH; ++s, ++f)
                for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e)
                    put(lane, e, L.p_w[l] + (int64_t)hid_feature16(s, lane >> 5, e) * fan + (32 * t + (lane & 31)));
    }
    int32_t* B = T + (int64_t)(n.n_frag + n.n_bw_frag) * 512;
    for (int l = 0; l < L.depth; ++l) for (int j = 0; j < H; ++j) B[l * H + j] = (int32_t)(L.p_b[l] + j);
    for (int j = 0; j < 3; ++j) B[L.depth * H + j] = (int32_t)(L.p_bc + j);
    B[L.depth * H + 3] = (int32_t)L.p_bs;
    remap_sources(L, T, n.pack_entries);
    return TNERF_OK;
}

// ---- bf16 training plan: weight-gradient jobs over the tile-organised stash (tnerf_internal.h)
namespace {
struct Job16Class { int a_ft0, n_at, b_ft0, n_bt, wa, has_bias, chunks; int64_t slab0, slab_stride; };

int build_classes16(const Net16& n, int64_t tiles, int n_cu, std::vector<Job16Class>& cls, int64_t* slab_total) {
    const int NT = n.hidden / 32;
    bool bad = false;
    auto add = [&](int a0, int nat, int b0, int nbt, int bias) {
        Job16Class c{}; c.a_ft0 = a0; c.n_at = nat; c.b_ft0 = b0; c.n_bt = nbt; c.has_bias = bias;
        int cost; pick_split(nat, nbt, &c.wa, &cost); if (c.wa == 0) bad = true;
        cls.push_back(c);
    };
    // class order (what the reduce table refers to): [0] (dZ_0, ENC) owns b_0; [l] (dZ_l, H_{l-1}) owns b_l;
    // [depth] (dZ_skip, ENC) if skip; [last] (dZ_head, H_{depth-1}) owns the head biases
    add(n.ft_dz[0], NT, n.ft_enc, 2, 1);
    for (int l = 1; l < n.depth; ++l) add(n.ft_dz[l], NT, n.ft_h[l - 1], NT, 1);
    if (n.skip_at > 0) add(n.ft_dz[n.skip_at], NT, n.ft_enc, 2, 0);
    add(n.ft_dzh, 1, n.ft_h[n.depth - 1], NT, 1);
    if (bad) { tn_set_error("bf16 wgrad: no kernel for this layer-shape / wave split"); return TNERF_EUNSUPPORTED; }
    if ((int)cls.size() > TN_RED_MAXCLS) { tn_set_error("too many wgrad job classes"); return TNERF_EUNSUPPORTED; }
    // The kernel is HBM-bound: a workgroup's time ~ bytes it streams = (n_at + n_bt) * 2 KB per tile (+ a fixed cost per
    // tile).  One workgroup per CU: keep splitting the class whose workgroups stream the most.
    for (auto& c : cls) c.chunks = 1;
    int total = (int)cls.size();
    while (total < n_cu) {
        int best = -1; double best_t = 0.0;
        for (size_t i = 0; i < cls.size(); ++i) {
            if (cls[i].chunks >= tiles) continue;
            // fitted to per-workgroup stamps of the whole kernel (tools/bf16_time_probe.py on a -DTN_STAMPS build, 8x256, 4096 x 64): 2700 /
            // 1664 / 1637 cycles per sample tile for the 8x8 / 8x2 / 1x8 classes = the bytes, plus a little for the one-tile-row head class
            const double t = ((double)(cls[i].n_at + cls[i].n_bt) * 2048.0 + (cls[i].n_at < 2 ? 1400.0 : 0.0)) * (double)((tiles + cls[i].chunks - 1) / cls[i].chunks);
            if (t > best_t) { best_t = t; best = (int)i; }
        }
        if (best < 0) break;
        ++cls[best].chunks; ++total;
    }
    int64_t off = 0;
    for (auto& c : cls) {
        const int64_t per = (tiles + c.chunks - 1) / c.chunks;
        c.chunks = (int)((tiles + per - 1) / per);
        c.slab_stride = (int64_t)c.n_at * 32 * (c.n_bt * 32) + (int64_t)c.n_at * 32;
        c.slab0 = off; off += c.slab_stride * c.chunks;
    }
    if (off >= (int64_t)1 << 31) { tn_set_error("slab workspace exceeds int32 offsets"); return TNERF_EUNSUPPORTED; }
    *slab_total = off;
    return TNERF_OK;
}
}  // namespace

extern "C" int tnerf_bf16_train_sizes(const tnerf_mlp_desc* d, int64_t n_rays, int32_t n_samples, int32_t n_cu,
                                      tnerf_bf16_train_plan* out) {
    Net16 n; int rc = tn_build_net16(d, &n); if (rc) return rc;
    if (!out || n_rays < 1 || n_samples < 1 || n_cu < 1) {
        tn_set_error("tnerf_bf16_train_sizes: rays=%lld samples=%d n_cu=%d", (long long)n_rays, n_samples, n_cu); return TNERF_EINVAL; }
    const int64_t tiles = n_rays * ((n_samples + 31) / 32);
    std::vector<Job16Class> cls; int64_t slab = 0;
    rc = build_classes16(n, tiles, n_cu, cls, &slab); if (rc) return rc;
    int64_t jobs = 0; for (auto& c : cls) jobs += c.chunks;
    MlpLayout L; rc = tn_build_layout(d, &L); if (rc) return rc;
    out->n_tiles = tiles;
    out->stash_bytes = TN16_STASH_FRAG_BYTES(n, tiles) + TN16_STASH_MASK_BYTES(n, tiles) + TN16_STASH_OUT_BYTES(n, tiles);
    out->slab_floats = slab;
    out->job_ints = jobs * TN_JOB_INTS;
    out->reduce_ints = TN_RED_HDR + 2 * L.n_params;
    out->n_jobs = jobs;
    return TNERF_OK;
}

extern "C" int tnerf_bf16_train_fill(const tnerf_mlp_desc* d, int64_t n_rays, int32_t n_samples, int32_t n_cu,
                                     int32_t* job_table, int32_t* reduce_table) {
    Net16 n; int rc = tn_build_net16(d, &n); if (rc) return rc;
    MlpLayout L; rc = tn_build_layout(d, &L); if (rc) return rc;
    if (n_rays < 1 || n_samples < 1 || n_cu < 1) { tn_set_error("tnerf_bf16_train_fill: bad sizes"); return TNERF_EINVAL; }
    const int64_t tiles = n_rays * ((n_samples + 31) / 32);
    std::vector<Job16Class> cls; int64_t slab = 0;
    rc = build_classes16(n, tiles, n_cu, cls, &slab); if (rc) return rc;
    if (job_table) {
        int64_t j = 0;
        int maxch = 0; for (auto& c : cls) maxch = std::max(maxch, c.chunks);
        for (int ch = 0; ch < maxch; ++ch)
            for (size_t ci = 0; ci < cls.size(); ++ci) {
                const Job16Class& c = cls[ci];
                if (ch >= c.chunks) continue;
                const int64_t per = (tiles + c.chunks - 1) / c.chunks;
                int32_t* r = job_table + j * TN_JOB_INTS;
                memset(r, 0, sizeof(int32_t) * TN_JOB_INTS);
                r[JOB_A_ROW0] = c.a_ft0; r[JOB_A_ROWS] = c.n_at * 32; r[JOB_B_ROW0] = c.b_ft0; r[JOB_B_ROWS] = c.n_bt * 32;
                r[JOB_N_AT] = c.n_at; r[JOB_N_BT] = c.n_bt; r[JOB_WA] = c.wa;
                r[JOB_MBLK0] = (int32_t)(ch * per);
                r[JOB_MBLKN] = (int32_t)std::max<int64_t>(0, std::min<int64_t>(per, tiles - ch * per));
                r[JOB_SLAB_OFF] = (int32_t)(c.slab0 + c.slab_stride * ch);
                r[JOB_CLASS] = (int32_t)ci; r[JOB_HAS_BIAS] = c.has_bias;
                ++j;
            }
    }
    std::vector<int32_t> red_padded;                                     // built in the padded parameter space
    int32_t* const reduce_out = reduce_table;
    if (reduce_table && L.hidden_true != L.hidden) { red_padded.assign((size_t)(TN_RED_HDR + 2 * L.n_params_padded), 0); reduce_table = red_padded.data(); }
    if (reduce_table) {
        int32_t* hdr = reduce_table;
        memset(hdr, 0, sizeof(int32_t) * TN_RED_HDR);
        hdr[0] = (int32_t)cls.size();
        for (size_t ci = 0; ci < cls.size(); ++ci) {
            hdr[1 + 4 * ci + 0] = (int32_t)cls[ci].slab0;
            hdr[1 + 4 * ci + 1] = (int32_t)cls[ci].slab_stride;
            hdr[1 + 4 * ci + 2] = cls[ci].chunks;
        }
        int32_t* E = reduce_table + TN_RED_HDR;
        auto put = [&](int64_t param, int cls_id, int64_t elem) { E[2 * param] = (int32_t)elem; E[2 * param + 1] = cls_id; };
        const int H = L.hidden;
        const int cls_skip_enc = L.skip_at > 0 ? L.depth : -1;
        const int cls_head = (int)cls.size() - 1;
        // input column -> column (32 T + c) of the two ENC feature tiles
        int inv[64]; for (int c = 0; c < 64; ++c) inv[c] = -1;
        for (int u = 0; u < TN16_KE; ++u) for (int h = 0; h < 2; ++h) for (int e = 0; e < 8; ++e) {
            const int col = enc_column16(n.Lf, u, h, e);
            if (col >= 0) inv[col] = 32 * (u >> 1) + TN_ACC_ROW(8 * (u & 1) + e, h);
        }
        for (int l = 0; l < L.depth; ++l) {
            const int fan = L.fan_in[l];
            const int ldm = cls[l].n_bt * 32;
            for (int row = 0; row < H; ++row) {
                for (int k = 0; k < fan; ++k) {
                    const int64_t pi = L.p_w[l] + (int64_t)row * fan + k;
                    if (l == 0) put(pi, 0, (int64_t)row * ldm + inv[k]);
                    else if (k < H) put(pi, l, (int64_t)row * ldm + k);
                    else put(pi, cls_skip_enc, (int64_t)row * 64 + inv[k - H]);
                }
                put(L.p_b[l] + row, l, (int64_t)cls[l].n_at * 32 * ldm + row);
            }
        }
        const int ldh = cls[cls_head].n_bt * 32;
        for (int k = 0; k < H; ++k) {
            put(L.p_ws + k, cls_head, (int64_t)3 * ldh + k);
            for (int row = 0; row < 3; ++row) put(L.p_wc + (int64_t)row * H + k, cls_head, (int64_t)row * ldh + k);
        }
        put(L.p_bs, cls_head, (int64_t)32 * ldh + 3);
        for (int row = 0; row < 3; ++row) put(L.p_bc + row, cls_head, (int64_t)32 * ldh + row);
        if (reduce_table != reduce_out) reduce_to_true(L, reduce_table, reduce_out);
    }
    return TNERF_OK;
}

// ------------------------------------------------------------------------------------ x3 chain (tnerf_internal.h, NetX3)
extern "C" int tn_build_netx3(const tnerf_mlp_desc* d, NetX3* n) {
    int rc = check_desc(d); if (rc) return rc;
    if (d->in_dim < 9 || (d->in_dim - 3) % 6 != 0) {
        tn_set_error("x3 chain: only the fused paths are built and they need in_dim = 6L+3; got %d", d->in_dim);
        return TNERF_EUNSUPPORTED;
    }
    memset(n, 0, sizeof(*n));
    const int H = d->hidden <= 128 ? 128 : 256;                            // the kernel width (zero-padded weights, see check_desc)
    n->in_dim = d->in_dim; n->hidden = H; n->depth = d->depth; n->skip_at = d->skip_at; n->Lf = (d->in_dim - 3) / 6;
    n->NT = H / 32; n->KH = H / 16; n->rec_frags = n->NT / 2 * TX_NP;       // a record covers HALF of the output tiles
    int rec = 2 * TN16_KE;
    n->fw_rec0[0] = 0;
    for (int l = 1; l < d->depth; ++l) { n->fw_rec0[l] = rec; rec += 2 * (n->KH + ((d->skip_at > 0 && l == d->skip_at) ? TN16_KE : 0)); }
    n->fw_rec0[d->depth] = rec;
    rec += n->KH / (n->NT / 2);                                             // heads: NT/2 k-steps of the one head tile per record
    n->fw_rec0[d->depth + 1] = rec;
    if ((rec * n->rec_frags) % TX_STAGE != 0) { tn_set_error("x3 chain: the record stream is not a whole number of stages"); return TNERF_EUNSUPPORTED; }
    n->n_rec = rec; n->n_stage = rec * n->rec_frags / TX_STAGE;
    const int rps = TX_STAGE / n->rec_frags;                               // records per stage
    n->n_bw_rec = rps + (d->depth - 1) * 2 * n->KH;                         // heads^T (two records, padded to a stage) + transposed hidden layers
    if ((n->n_bw_rec * n->rec_frags) % TX_STAGE != 0) { tn_set_error("x3 chain: the backward record stream is not a whole number of stages"); return TNERF_EUNSUPPORTED; }
    n->n_bw_stage = n->n_bw_rec * n->rec_frags / TX_STAGE;
    n->bias_off = (rec + n->n_bw_rec) * n->rec_frags * 1024;
    n->n_bias = d->depth * H + 4;
    n->meta_off = n->bias_off + n->n_bias * 4;                              // the scale records follow the biases (one LDS copy serves both)
    n->packed_bytes = (int64_t)n->meta_off + (int64_t)(d->depth + 1) * TX_META * 4;
    n->pack_entries = (int64_t)(rec + n->n_bw_rec) * n->rec_frags * 512 + n->n_bias;
    return TNERF_OK;
}

extern "C" int tnerf_x3_plan_sizes(const tnerf_mlp_desc* d, tnerf_bf16_sizes* out) {
    NetX3 n; int rc = tn_build_netx3(d, &n); if (rc) return rc;
    if (!out) { tn_set_error("tnerf_x3_plan_sizes: NULL output"); return TNERF_EINVAL; }
    out->packed_bytes = n.packed_bytes; out->pack_entries = n.pack_entries;
    out->n_fragments = (int64_t)(n.n_rec + n.n_bw_rec) * n.rec_frags; out->bias_offset_bytes = n.bias_off;
    out->n_fwd_fragments = (int64_t)n.n_rec * n.rec_frags;
    return TNERF_OK;
}

extern "C" int tnerf_x3_pack_table(const tnerf_mlp_desc* d, int32_t* T) {
    NetX3 n; int rc = tn_build_netx3(d, &n); if (rc) return rc;
    MlpLayout L; rc = tn_build_layout(d, &L); if (rc) return rc;
    if (!T) { tn_set_error("tnerf_x3_pack_table: NULL table"); return TNERF_EINVAL; }
    for (int64_t i = 0; i < n.pack_entries; ++i) T[i] = -1;
    const int H = L.hidden, NH = n.NT / 2, KH = n.KH, Lf = n.Lf;
    int64_t rec = 0;                                                        // running record index
    // both piece fragments of (record, tile slot tl) refer to the same parameters: the pack kernel derives the piece from
    // the fragment's position (fragment index mod TX_NP).  Tile slot tl of a half-h record is output tile h*NH + tl.
    auto put = [&](int tl, int lane, int e, int64_t src) {
        for (int piece = 0; piece < TX_NP; ++piece) T[(((rec * NH + tl) * TX_NP + piece) * 64 + lane) * 8 + e] = (int32_t)src;
    };
    auto head_w = [&](int row, int k) -> int64_t { return row < 3 ? L.p_wc + (int64_t)row * H + k : (row == 3 ? L.p_ws + k : -1); };
    for (int l = 0; l < L.depth; ++l) {
        const int fan = L.fan_in[l];
        const bool skip = L.skip_at > 0 && l == L.skip_at;
        for (int half = 0; half < 2; ++half) {                              // half-pass A (tiles 0..NH-1), then B
            if (l > 0)
                for (int s = 0; s < KH; ++s, ++rec)
                    for (int tl = 0; tl < NH; ++tl) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e)
                        put(tl, lane, e, L.p_w[l] + (int64_t)(32 * (half * NH + tl) + (lane & 31)) * fan + hid_feature16(s, lane >> 5, e));
            if (l == 0 || skip)
                for (int u = 0; u < TN16_KE; ++u, ++rec)
                    for (int tl = 0; tl < NH; ++tl) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e) {
                        const int c = enc_column16(Lf, u, lane >> 5, e);
                        if (c >= 0) put(tl, lane, e, L.p_w[l] + (int64_t)(32 * (half * NH + tl) + (lane & 31)) * fan + (l == 0 ? 0 : H) + c);
                    }
        }
    }
    for (int s = 0; s < KH; ++s) {                                          // heads: rows r,g,b (rgb.0), sigma.0; slot s % NH of record s / NH
        for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e) {
            const int64_t src = head_w(lane & 31, hid_feature16(s, lane >> 5, e));
            if (src >= 0) put(s % NH, lane, e, src);
        }
        if (s % NH == NH - 1) ++rec;
    }
    // ---- backward stream
    rec = n.n_rec;
    for (int half = 0; half < 2; ++half, ++rec)                             // heads^T: dH[k] = sum_{row<4} W_head[row][k] dZh[row]
        for (int tl = 0; tl < NH; ++tl)
            for (int lane = 0; lane < 32; ++lane) for (int e = 0; e < 4; ++e) put(tl, lane, e, head_w(e, 32 * (half * NH + tl) + lane));
    rec = n.n_rec + TX_STAGE / n.rec_frags;
    for (int l = L.depth - 1; l >= 1; --l) {
        const int fan = L.fan_in[l];
        for (int half = 0; half < 2; ++half)
            for (int s = 0; s < KH; ++s, ++rec)
                for (int tl = 0; tl < NH; ++tl) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e)
                    put(tl, lane, e, L.p_w[l] + (int64_t)hid_feature16(s, lane >> 5, e) * fan + (32 * (half * NH + tl) + (lane & 31)));
    }
    int32_t* B = T + (int64_t)(n.n_rec + n.n_bw_rec) * n.rec_frags * 512;
    for (int l = 0; l < L.depth; ++l) for (int j = 0; j < H; ++j) B[l * H + j] = (int32_t)(L.p_b[l] + j);
    for (int j = 0; j < 3; ++j) B[L.depth * H + j] = (int32_t)(L.p_bc + j);
    B[L.depth * H + 3] = (int32_t)L.p_bs;
    remap_sources(L, T, n.pack_entries);
    return TNERF_OK;
}
