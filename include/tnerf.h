/*
 * tnerf.h — C ABI of libtnerf_hip.so: the MI355X (gfx950) TinyNeRF render/train hot path.
 *
 * The reference (avihaig/tiny-nerf-pytorch) has no FFI/plugin interface of its own: its hot
 * path is the Python call surface src/{rays,sampling,encoding,nerf,volume}.py + the step body
 * and render_one of src/train.py.  Each entry point below names the reference function
 * (file:line under /root/reference) whose arithmetic it replaces; the Python mirror of that call
 * surface (tiny-nerf-pytorch_amd/src/{rays,sampling,...}.py) binds these symbols with ctypes.
 *
 * Conventions
 *   - plain C: raw device pointers (tensor.data_ptr()), explicit sizes, scalars; no torch types.
 *   - return int: 0 = ok, <0 = TNERF_E* (bad argument), >0 = hipError_t of a failed HIP call.
 *     tnerf_last_error_string() gives a thread-local description.  Nothing throws across the ABI.
 *   - the CALLER owns all memory (outputs, stashes, workspaces, tables); the library never
 *     allocates or frees device memory.  Process-wide state is limited to values that are written once and never
 *     change: per-device kernel facts (CU count, granted dynamic-LDS size; atomics indexed by device) and the dlopen'ed
 *     RCCL / hipBLAS function tables (std::call_once each; handles and communicators are the caller's).
 *   - every device entry point is asynchronous on the given stream (hipStream_t passed as void*)
 *     and performs no host synchronisation, so it can be captured into a hipGraph.
 *   - all tensors are contiguous fp32 unless stated; index tensors are int64.
 *   - functions marked HOST touch no GPU and may be called without one (unit-tested on CPU).
 */
#ifndef TNERF_H
#define TNERF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TNERF_ABI_VERSION 3

#define TNERF_OK            0
#define TNERF_EINVAL       (-1)  /* bad size / NULL pointer / inconsistent arguments        */
#define TNERF_EUNSUPPORTED (-2)  /* valid in the reference, not built here (see message)    */
#define TNERF_ESMALL       (-3)  /* caller-provided table/workspace too small               */

typedef void* tnerf_stream_t;    /* hipStream_t */

/* TinyNeRF(in_dim, hidden, depth, skip_at)                       [src/nerf.py:10-27]
 * skip_at: the input is concatenated after layer index skip_at-1 (cat([h, x]), nerf.py:37-38);
 * 1 <= skip_at <= depth-1, or 0 for "no skip".  1 <= hidden <= 256 (widths other than 128 / 256 run zero-padded on the
 * 128- / 256-wide kernels: the host tables embed the model, the padding units stay exactly 0); in_dim <= 64.  Wider models and the
 * gradient w.r.t. the input: the tnerf_mlp_*_generic entry points near the end of this header. */
typedef struct tnerf_mlp_desc {
    int32_t in_dim;
    int32_t hidden;
    int32_t depth;
    int32_t skip_at;
    int32_t flags;           /* TNERF_FLAG_*; 0 = defaults                                        */
} tnerf_mlp_desc;
/* Which matrix pipe the fp32 kernels of the fused paths use: the weight-gradient kernel (tnerf_wgrad with a stash written by
 * an x3 chain, inside tnerf_train_*) and, wherever a tnerf_mlp_pack_x3 stream is passed (the *_x3 entry points, the packed_x3
 * arguments), the forward and dgrad chain kernels.
 * Default (0), "x3": the fp16 matrix pipe with THREE partial products.  Every fp32 operand is scaled by a power of two (per
 * layer for weights, per sample for activations / per row group in the weight-gradient GEMMs) and carried as two fp16
 * pieces x = x1 + x2 (round-to-nearest, |x - x1 - x2| <= 2^-22 |x|); a*b is formed as a1*b1 in one fp32 accumulator and
 * a1*b2 + a2*b1 in a second one (three v_mfma_f32_32x32x16_f16 per 16 k), a2*b2 <= 2^-22 |ab| is dropped.  This is an
 * APPROXIMATION of fp32 arithmetic with fp32-grade error, not fp32 arithmetic: measured against fp64 the activations are
 * closer than an fp32 fma chain's and every gradient tensor is within 2x of the reference's own CPU fp32 error (tests/
 * test_gpu_parity.py states the bounds; DESIGN.md 3 has the pipe's accumulation model) — at 3/16 of the fp32-MFMA time.
 * TNERF_FLAG_FP32_MFMA selects v_mfma_f32_32x32x2_f32 (plain fp32 fma chains) for all of them instead; the entry points
 * that take only the fp32 pack (tnerf_render_fused, tnerf_train_fwd_fused, tnerf_mlp_fwd, ...) always run those. */
#define TNERF_FLAG_FP32_MFMA 1

/* Sizes (in elements) of everything the caller must allocate for a model + sample count. */
typedef struct tnerf_plan_sizes {
    int64_t n_params;        /* flat fp32 parameter / gradient / Adam-moment buffers          */
    int64_t packed_floats;   /* MFMA-fragment-ordered copy of the weights (fwd + transposed)  */
    int64_t stash_floats;    /* activations saved by a training forward (block-major) + sign bits */
    int64_t slab_floats;     /* weight-gradient partial slabs (one per wgrad workgroup)       */
    int64_t job_ints;        /* wgrad job table (int32)                                       */
    int64_t reduce_ints;     /* slab -> flat-gradient gather table (int32, 2 per parameter)   */
    int64_t n_jobs;          /* wgrad workgroups                                              */
    int64_t stash_row_stride;/* padded sample count Mp (row stride of the stash)              */
} tnerf_plan_sizes;

/* ---------------------------------------------------------------------------------- misc */
int         tnerf_version(void);                 /* HOST: TNERF_ABI_VERSION — a binding MUST compare it with the header it
                                                    was written against before the first call (ABI 2 -> 3 changed signatures) */
const char* tnerf_last_error_string(void);       /* HOST: thread-local, never NULL             */

/* ------------------------------------------------------------------------ host-side tables */
/* HOST. Depth tables of stratified_samples: t = torch.linspace(0,1,S) as ATen computes it on CPU,
 * z = near*(1-t) + far*t, and the jitter interval [lo, hi] around each z (mids of neighbours).
 * Bit-exact with the reference's CPU fp32 arithmetic.            [src/sampling.py:16-23]
 * out: ztab[3*S] = { z[S] | lo[S] | hi[S] };  t_out[S] (may be NULL). */
int tnerf_sample_tables(float near, float far, int32_t n_samples, float* ztab, float* t_out);

/* HOST. Offsets (in floats) of the 2*depth+4 parameter tensors inside the flat buffer, in
 * state_dict order layers.i.weight, layers.i.bias, sigma.0.weight, sigma.0.bias, rgb.0.weight,
 * rgb.0.bias, plus their (rows, cols).                         [src/nerf.py:18-27]           */
int64_t tnerf_param_count(const tnerf_mlp_desc* d);
int     tnerf_param_layout(const tnerf_mlp_desc* d, int64_t* offsets, int64_t* rows, int64_t* cols);

/* HOST. Column pairing of the network input used by the MFMA kernels: step s feeds input column
 * emap[2*s+0] to lane-half 0 and emap[2*s+1] to lane-half 1 (-1 = zero padding).  For
 * in_dim = 6L+3 (PositionalEncoding(L, include_input=True), encoding.py:27-33) the pairing is
 * (sin(2^k x_c), cos(2^k x_c)); otherwise consecutive columns.  n_steps is 20 or 32. */
int tnerf_input_pairing(int32_t in_dim, int32_t* emap /* [64] */, int32_t* n_steps);

/* HOST. Sizes of the buffers/tables for `n_samples_total` = rays*samples (or rows of x), with the
 * wgrad work split over `n_cu` compute units (256 on MI355X). */
int tnerf_plan_sizes_query(const tnerf_mlp_desc* d, int64_t n_samples_total, int32_t n_cu,
                           tnerf_plan_sizes* out);

/* HOST. Fill the three int32 tables (caller uploads them to the GPU):
 *   pack_table  [packed_floats]: packed[i] = params[pack_table[i]] (or 0 if < 0)
 *   job_table   [job_ints]     : one record per wgrad workgroup
 *   reduce_table[reduce_ints]  : for parameter i: {slab offset of chunk 0, job-class id}       */
int tnerf_plan_fill(const tnerf_mlp_desc* d, int64_t n_samples_total, int32_t n_cu,
                    int32_t* pack_table, int32_t* job_table, int32_t* reduce_table);

/* --------------------------------------------------------------------------- stage kernels */
/* get_rays(H, W, focal, c2w)                                     [src/rays.py:3-33]
 * c2w: device, 16 floats row-major.  rays_d: [H*W,3] unit directions, flat index row*W+col.
 * rays_o: [H*W,3] or NULL (the reference returns a stride-0 expand of the translation). */
int tnerf_get_rays(int32_t H, int32_t W, float focal, const float* c2w,
                   float* rays_o, float* rays_d, tnerf_stream_t stream);

/* stratified_samples (+ optional PositionalEncoding of the points)
 *                                      [src/sampling.py:3-28, src/encoding.py:21-33]
 * ztab: device copy of tnerf_sample_tables' table (3*S).
 * randomized=0: z = ztab.z;  randomized=1: z = lo + (hi-lo)*u with u = t_rand[r,s] if t_rand != NULL
 * (parity mode: the caller drew torch.rand) else Philox4x32-10(seed, offset + r*S+s).
 * Outputs (each may be NULL): z_vals [R,S], pts [R,S,3], enc [R*S, 6L(+3)]. */
int tnerf_sample_encode_fwd(const float* rays_o, const float* rays_d, int64_t n_rays, int32_t n_samples,
                            const float* ztab, int32_t randomized, const float* t_rand,
                            uint64_t seed, uint64_t offset,
                            float* z_vals, float* pts,
                            float* enc, int32_t num_freqs, int32_t include_input,
                            tnerf_stream_t stream);

/* stratified_samples with PER-RAY near / far ("tensors broadcastable to (N_rays, 1)", src/sampling.py:8,17).
 * t_tab: device copy of tnerf_sample_tables' t_out (the fp32 torch.linspace(0,1,S)); near, far: [R] device floats.
 * Bins of ray r: near_r*(1-t_i) + far_r*t_i, rounded op by op like the reference's broadcast; jitter as above.
 * Outputs (one may be NULL): z_vals [R,S], pts [R,S,3]. */
int tnerf_sample_per_ray_fwd(const float* rays_o, const float* rays_d, int64_t n_rays, int32_t n_samples,
                             const float* t_tab, const float* near, const float* far,
                             int32_t randomized, const float* t_rand, uint64_t philox_seed, uint64_t philox_offset,
                             float* z_vals, float* pts, tnerf_stream_t stream);

/* PositionalEncoding.forward on arbitrary points x [n,3] -> out [n, 6L(+3)]   [src/encoding.py:21-33] */
int tnerf_posenc_fwd(const float* x, int64_t n, int32_t num_freqs, int32_t include_input,
                     float* out, tnerf_stream_t stream);

/* volume_render forward                                          [src/volume.py:3-44]
 * rgb [R,S,3], sigma [R,S], z_vals [R,S], rays_d [R,3] -> comp_rgb [R,3], depth [R], acc [R],
 * weights [R,S] (depth/acc/weights may be NULL).  One ray per wavefront. */
int tnerf_composite_fwd(const float* rgb, const float* sigma, const float* z_vals, const float* rays_d,
                        int64_t n_rays, int32_t n_samples, int32_t white_bkgd,
                        float* comp_rgb, float* depth, float* acc, float* weights,
                        tnerf_stream_t stream);

/* volume_render backward w.r.t. rgb and sigma (what autograd derives from volume.py:18-42).
 * Upstream gradients g_comp [R,3], g_depth [R], g_acc [R], g_weights [R,S]; any may be NULL (= 0). */
int tnerf_composite_bwd(const float* rgb, const float* sigma, const float* z_vals, const float* rays_d,
                        int64_t n_rays, int32_t n_samples, int32_t white_bkgd,
                        const float* g_comp, const float* g_depth, const float* g_acc, const float* g_weights,
                        float* d_rgb, float* d_sigma, tnerf_stream_t stream);

/* The geometry side of the same chain — what the reference's autograd would also give (its training never asks: the points carry
 * no grad, train.py:114-117), for callers that learn poses or resample:
 *   tnerf_composite_bwd_geom: tnerf_composite_bwd + d_z [R,S] (dL/dz_vals) and d_rays_d [R,3] (either may be NULL)   [volume.py:18-44]
 *   tnerf_sample_bwd        : pts = o + d z  ->  dL/drays_o [R,3], dL/drays_d [R,3], dL/dz [R,S] from g_pts [R,S,3]; z_row_stride 0 =
 *                             one shared row of depths (the non-randomized table)                                   [sampling.py:27]
 *   tnerf_posenc_bwd        : dL/dx [n,3] from g_out [n, 6L(+3)]                                                    [encoding.py:27-33]
 *   tnerf_get_rays_bwd      : dL/dc2w (16 floats; rotation block from g_rays_d, deterministic two-level sum; the translation
 *                             column's gradient is the sum of dL/drays_o, which the stride-0 expand gives the caller for free);
 *                             scratch of tnerf_get_rays_bwd_scratch_floats(H, W) floats (HOST query), else TNERF_ESMALL       [rays.py:21-31] */
int64_t tnerf_get_rays_bwd_scratch_floats(int32_t H, int32_t W);
int tnerf_get_rays_bwd(int32_t H, int32_t W, float focal, const float* c2w, const float* g_rays_d, float* scratch, int64_t scratch_floats,
                       float* d_c2w, tnerf_stream_t stream);
int tnerf_composite_bwd_geom(const float* rgb, const float* sigma, const float* z_vals, const float* rays_d,
                             int64_t n_rays, int32_t n_samples, int32_t white_bkgd,
                             const float* g_comp, const float* g_depth, const float* g_acc, const float* g_weights,
                             float* d_rgb, float* d_sigma, float* d_z, float* d_rays_d, tnerf_stream_t stream);
int tnerf_sample_bwd(const float* rays_d, const float* z_vals, int64_t z_row_stride, const float* g_pts, int64_t n_rays, int32_t n_samples,
                     float* d_rays_o, float* d_rays_d, float* d_z, tnerf_stream_t stream);
int tnerf_posenc_bwd(const float* x, int64_t n, int32_t num_freqs, int32_t include_input, const float* g_out, float* d_x,
                     tnerf_stream_t stream);

/* ------------------------------------------------------------------------------- MLP (MFMA) */
/* Re-order the flat parameters into MFMA fragment order: packed[i] = params[pack_table[i]]. */
int tnerf_mlp_pack(const float* params, const int32_t* pack_table, int64_t packed_floats,
                   float* packed, tnerf_stream_t stream);

/* TinyNeRF.forward                                               [src/nerf.py:29-41]
 * x [M,in_dim] -> rgb [M,3] (sigmoid), sigma [M,1] (ReLU).  stash != NULL saves the activations
 * needed by tnerf_mlp_bwd (row stride stash_row_stride from the plan of M). */
int tnerf_mlp_fwd(const tnerf_mlp_desc* d, const float* packed, const float* x, int64_t n_rows,
                  float* rgb, float* sigma, float* stash, int64_t stash_row_stride,
                  tnerf_stream_t stream);

/* Backward of TinyNeRF.forward w.r.t. all parameters: grads [n_params] is OVERWRITTEN with
 * dL/dparams for upstream d_rgb [M,3], d_sigma [M,1].  (No gradient w.r.t. x: the reference never
 * asks for one — pts carry no grad, train.py:114-117.)  Deterministic (slab reduction, no atomics). */
int tnerf_mlp_bwd(const tnerf_mlp_desc* d, const float* packed, int64_t n_rows,
                  const float* d_rgb, const float* d_sigma,
                  float* stash, int64_t stash_row_stride,
                  const int32_t* job_table, int64_t n_jobs, float* slabs,
                  const int32_t* reduce_table, float* grads, tnerf_stream_t stream);

/* ----------------------------------------------------------------------------- fused paths */
/* Body of render_one for a batch of rays: sample -> encode -> MLP -> composite with nothing but
 * rays in and colours out of HBM.                                [src/train.py:50-56]
 * randomized / t_rand / seed / offset as in tnerf_sample_encode_fwd.
 * comp_rgb [R,3]; depth [R], acc [R] may be NULL.  Requires in_dim == 6L+3. */
int tnerf_render_fused(const tnerf_mlp_desc* d, const float* packed,
                       const float* rays_o, const float* rays_d, int64_t n_rays, int32_t n_samples,
                       const float* ztab, int32_t randomized, const float* t_rand,
                       uint64_t seed, uint64_t offset, int32_t white_bkgd,
                       float* comp_rgb, float* depth, float* acc, tnerf_stream_t stream);

/* Training forward of the same body (train.py:114-121): also fills `stash`. */
int tnerf_train_fwd_fused(const tnerf_mlp_desc* d, const float* packed,
                          const float* rays_o, const float* rays_d, int64_t n_rays, int32_t n_samples,
                          const float* ztab, int32_t randomized, const float* t_rand,
                          uint64_t seed, uint64_t offset, int32_t white_bkgd,
                          float* comp_rgb, float* stash, int64_t stash_row_stride,
                          tnerf_stream_t stream);

/* Training backward: given g_comp = dL/dcomp_rgb [R,3], overwrite grads [n_params]
 * (what loss.backward() accumulates, train.py:126).  Same sampling arguments as the forward.
 * packed_x3: NULL, or the tnerf_mlp_pack_x3 stream of the same parameters — the dgrad chain then runs on the fp16 matrix
 * pipe (x3) unless desc.flags has TNERF_FLAG_FP32_MFMA. */
int tnerf_train_bwd_fused(const tnerf_mlp_desc* d, const float* packed,
                          const float* rays_o, const float* rays_d, int64_t n_rays, int32_t n_samples,
                          const float* ztab, int32_t randomized, const float* t_rand,
                          uint64_t seed, uint64_t offset, int32_t white_bkgd,
                          const float* g_comp, float* stash, int64_t stash_row_stride,
                          const int32_t* job_table, int64_t n_jobs, float* slabs,
                          const int32_t* reduce_table, float* grads, const void* packed_x3, tnerf_stream_t stream);

/* The three stages of tnerf_train_bwd_fused as separate calls (same arguments; used to time each kernel):
 *   dgrad : composite backward + register-resident dgrad chain, fills the dZ rows of the stash
 *   wgrad : per-layer weight-gradient GEMMs over the stash, one partial slab per workgroup
 *   reduce: fixed-order sum of the slabs into grads [n_params] (overwritten) */
int tnerf_train_dgrad_fused(const tnerf_mlp_desc* d, const float* packed,
                            const float* rays_o, const float* rays_d, int64_t n_rays, int32_t n_samples,
                            const float* ztab, int32_t randomized, const float* t_rand,
                            uint64_t seed, uint64_t offset, int32_t white_bkgd,
                            const float* g_comp, float* stash, int64_t stash_row_stride, tnerf_stream_t stream);
int tnerf_wgrad(const tnerf_mlp_desc* d, const float* stash, int64_t stash_row_stride, int64_t n_samples_total,
                const int32_t* job_table, int64_t n_jobs, float* slabs, tnerf_stream_t stream);
int tnerf_wgrad_reduce(const float* slabs, const int32_t* reduce_table, int64_t n_params, float* grads,
                       tnerf_stream_t stream);

/* One whole minibatch step up to (not including) the optimizer (train.py:114-126):
 * forward, loss = sum((comp-target)^2)/loss_denominator, backward -> grads (overwritten),
 * loss_out[0] = this batch's loss contribution (device scalar), comp_rgb [R,3].
 * g_comp_ws: per-ray workspace (dL/dcomp_rgb and the squared error of every ray, written by the forward kernel) of
 * g_comp_ws_floats >= tnerf_train_ws_floats(n_rays) floats; a smaller capacity is refused with TNERF_ESMALL before anything
 * is launched.  (ABI 1 needed 3*R floats, ABI 2 needs 4*R: the capacity argument exists so that a caller sized for an older
 * library gets an error instead of an out-of-bounds write.)
 * loss_denominator = 3*R reproduces torch.mean (train.py:122); a ray shard passes the GLOBAL 3*R. */
int64_t tnerf_train_ws_floats(int64_t n_rays);   /* HOST: floats of g_comp_ws / tnerf_step_args.ray_ws for n_rays rays; < 0: bad n_rays */
int tnerf_train_step_fused(const tnerf_mlp_desc* d, const float* packed,
                           const float* rays_o, const float* rays_d, const float* target,
                           int64_t n_rays, int32_t n_samples,
                           const float* ztab, int32_t randomized, const float* t_rand,
                           uint64_t seed, uint64_t offset, int32_t white_bkgd, double loss_denominator,
                           float* comp_rgb, float* g_comp_ws, int64_t g_comp_ws_floats, float* loss_out,
                           float* stash, int64_t stash_row_stride,
                           const int32_t* job_table, int64_t n_jobs, float* slabs,
                           const int32_t* reduce_table, float* grads,
                           const void* packed_x3 /* NULL, or the record stream of tnerf_mlp_pack_x3: the forward and dgrad chains
                                                    then run on the x3 kernels (unless TNERF_FLAG_FP32_MFMA) */,
                           tnerf_stream_t stream);

/* Camera-sourced rays: instead of gathering from the (N,HW,3) tables that the reference
 * precomputes with get_rays (src/train.py:94-101,110-112), ray r is generated inside the fused kernels from the
 * pose and the flat pixel index p = pix_index ? pix_index[r] : pix_first + r, with the arithmetic of tnerf_get_rays. */
typedef struct tnerf_camera {
    const float*   c2w;        /* device, 16 floats row-major                               */
    int32_t        H, W;
    float          focal;
    const int64_t* pix_index;  /* device [n_rays] flat pixel indices (randint, train.py:109) or NULL */
    int64_t        pix_first;  /* used when pix_index == NULL: pixels pix_first .. pix_first+n_rays-1 */
} tnerf_camera;

/* tnerf_render_fused with camera rays (one chunk of render_one, src/train.py:46-56). */
int tnerf_render_fused_cam(const tnerf_mlp_desc* d, const float* packed, const tnerf_camera* cam,
                           int64_t n_rays, int32_t n_samples,
                           const float* ztab, int32_t randomized, const float* t_rand,
                           uint64_t seed, uint64_t offset, int32_t white_bkgd,
                           float* comp_rgb, float* depth, float* acc, tnerf_stream_t stream);

/* tnerf_train_step_fused with camera rays; the target colour of ray r is pixels[pix_index[r]] (pixels: [H*W,3] of the
 * image being trained on, src/train.py:101,112). */
int tnerf_train_step_fused_cam(const tnerf_mlp_desc* d, const float* packed, const tnerf_camera* cam,
                               const float* pixels, int64_t n_rays, int32_t n_samples,
                               const float* ztab, int32_t randomized, const float* t_rand,
                               uint64_t seed, uint64_t offset, int32_t white_bkgd, double loss_denominator,
                               float* comp_rgb, float* g_comp_ws, int64_t g_comp_ws_floats, float* loss_out,
                               float* stash, int64_t stash_row_stride,
                               const int32_t* job_table, int64_t n_jobs, float* slabs,
                               const int32_t* reduce_table, float* grads, const void* packed_x3, tnerf_stream_t stream);

/* -------------------------------------------------------------- the whole step on device-resident state */
/* The body of the reference training loop (src/train.py:106-128) with NO per-step host input: which image (step % N,
 * train.py:108), which pixels (torch.randint there, :109) and which jitter (rand_like, sampling.py:24) come from a
 * device-side step counter and Philox4x32-10 inside the kernels; the loss gradient 2 (C - target) / denom (train.py:122)
 * is formed by the forward kernel; the counter advances inside the weight-gradient kernel; slab reduction, Adam
 * (train.py:80,125-128) and the re-packing of the updated weights are ONE finishing kernel.  One GPU: 4 launches per
 * step whose arguments never change, so the step can be captured into a hipGraph (below) and replayed.
 * Draws of step s (0-based) for global ray row g = ray_first + r, sample k:
 *     pixel  = Philox(seed ^ 0x9E3779B97F4A7C15, s * n_rays_global + g)  mod  H*W           (32 random bits)
 *     jitter = Philox(seed, (s * n_rays_global + g) * n_samples + k)  ->  24-bit uniform in [0,1)
 * so ranks that shard the rows of one global batch draw exactly what one GPU would. */
#define TNERF_PHASE_GRADIENT 1   /* forward + loss gradient + dgrad + wgrad -> slabs; *step += 1                       */
#define TNERF_PHASE_REDUCE   2   /* slabs -> grads (fixed order), loss_out = sum of squared errors / denom             */
#define TNERF_PHASE_UPDATE   4   /* Adam on (params, exp_avg, exp_avg_sq) with t = *step, scatter into `packed`        */
typedef struct tnerf_step_args {
    tnerf_mlp_desc desc;
    int32_t precision;            /* 0: fp32 kernels, 1: bf16 mode                                                   */
    int32_t phases;               /* TNERF_PHASE_* bits; REDUCE|UPDATE in one call = one kernel; multi-GPU: GRADIENT|REDUCE,
                                     all-reduce of grads, then UPDATE                                                 */
    /* dataset, resident in HBM */
    const float* poses;           /* [n_images,16]                                                                   */
    const float* pixels;          /* [n_images, H*W, 3]                                                              */
    int32_t n_images, H, W;
    float focal;
    /* batch */
    int64_t n_rays;               /* rays of THIS rank                                                               */
    int64_t ray_first;            /* first global row of this rank (dist.shard_bounds)                               */
    int64_t n_rays_global;        /* rows of the global batch                                                        */
    int32_t n_samples, white_bkgd;
    const float* ztab;            /* tnerf_sample_tables                                                             */
    uint64_t seed;
    double loss_denominator;      /* 3 * n_rays_global reproduces torch.mean                                         */
    int64_t* step;                /* DEVICE: completed steps                                                         */
    /* weights: fp32 fragment-packed floats (tnerf_mlp_pack) or the bf16 stream (tnerf_mlp_pack_bf16)                */
    const void* packed;
    /* workspaces */
    float* comp_rgb;              /* [n_rays,3]                                                                      */
    float* ray_ws;                /* per-ray workspace: dL/dcomp_rgb and the squared error of every ray              */
    int64_t ray_ws_floats;        /* its capacity, >= tnerf_train_ws_floats(n_rays), else TNERF_ESMALL                */
    int32_t* pix_out;             /* [n_rays] or NULL: the pixel every ray trained on                                */
    float* loss_out;              /* device scalar or NULL                                                           */
    void* stash; int64_t stash_row_stride;      /* fp32: plan stash + its row stride; bf16: the tile stash            */
    int64_t stash_capacity;       /* capacity of `stash`: fp32 floats >= tnerf_plan_sizes.stash_floats of n_rays*n_samples,
                                     bf16 bytes >= tnerf_bf16_train_plan.stash_bytes; smaller: TNERF_ESMALL              */
    const int32_t* job_table; int64_t n_jobs; float* slabs;
    const int32_t* reduce_table; float* grads;
    /* optimizer */
    float* params; float* exp_avg; float* exp_avg_sq;
    float lr, beta1, beta2, eps;
    const int32_t* scatter_table; /* [n_params, scatter_width]: positions of parameter i in `packed` (-1 terminated),
                                     the inverse of the pack table; NULL = do not re-pack                            */
    int32_t scatter_width;
    /* fp32 only: the x3 record stream (tnerf_mlp_pack_x3) — the forward and dgrad chains then run on the x3 kernels — and the inverse of
     * its pack table so that the finishing kernel keeps it current.  NULL = fp32-MFMA forward.                       */
    const void* packed_x3;
    const int32_t* scatter_x3;
    int32_t scatter_x3_width;
    const int32_t* pack_x3;       /* the x3 pack table itself (tnerf_x3_pack_table, on the device): after the update the layers' weight
                                     maxima and scales are refreshed through it (required with scatter_x3)                          */
} tnerf_step_args;
int tnerf_train_step_dataset(const tnerf_step_args* args, tnerf_stream_t stream);

/* hipGraph capture of whatever the caller launches on `stream` (not the NULL stream) between begin and end — e.g. one
 * tnerf_train_step_dataset call — and replay of the instantiated graph.  graph_exec is owned by the caller. */
int tnerf_graph_begin(tnerf_stream_t stream);
int tnerf_graph_end(tnerf_stream_t stream, void** graph_exec_out);
int tnerf_graph_launch(void* graph_exec, tnerf_stream_t stream);
int tnerf_graph_destroy(void* graph_exec);

/* ------------------------------------------------------------------- bf16 mode (BASELINE cfg 4) */
/* The same fused paths with bf16 weights and activations on v_mfma_f32_32x32x16_bf16: fp32 accumulation, fp32 biases,
 * fp32 ray / sample / compositing arithmetic (sample bins stay bit-exact); the encoder output and every hidden
 * activation are rounded to bf16 (round-to-nearest-even) where the fp32 path keeps fp32.  Weights are streamed once
 * per workgroup through an LDS ring and shared by its four wavefronts (64 samples each).  Tolerances against the fp32
 * path: SURVEY.md 8(d) cfg 4.  Requires in_dim = 6L+3. */
typedef struct tnerf_bf16_sizes {
    int64_t packed_bytes;       /* fragment-stream of bf16 weights followed by the fp32 biases          */
    int64_t pack_entries;       /* int32 entries of the pack table                                       */
    int64_t n_fragments;        /* 1 KB MFMA A-fragments: forward stream + backward (transposed) stream  */
    int64_t bias_offset_bytes;  /* where the fp32 biases start inside the packed buffer                  */
    int64_t n_fwd_fragments;    /* fragments of the forward stream (one pass over the network)           */
} tnerf_bf16_sizes;

/* HOST. Buffer / table sizes for a model. */
int tnerf_bf16_plan_sizes(const tnerf_mlp_desc* d, tnerf_bf16_sizes* out);
/* HOST. table[pack_entries]: first n_fragments*512 entries -> bf16 element i of the stream = bf16(params[table[i]])
 * (0 if < 0); the remaining entries -> fp32 bias j = params[table[n_fragments*512 + j]]. */
int tnerf_bf16_pack_table(const tnerf_mlp_desc* d, int32_t* table);
/* Gather + round the flat fp32 parameters into the packed buffer (after every optimizer step). [src/nerf.py:18-27] */
int tnerf_mlp_pack_bf16(const tnerf_mlp_desc* d, const float* params, const int32_t* table, void* packed16,
                        tnerf_stream_t stream);
/* tnerf_render_fused / tnerf_render_fused_cam in bf16 mode (same arguments; packed16 from tnerf_mlp_pack_bf16).
 *                                                                         [src/train.py:46-56] */
int tnerf_render_fused_bf16(const tnerf_mlp_desc* d, const void* packed16,
                            const float* rays_o, const float* rays_d, int64_t n_rays, int32_t n_samples,
                            const float* ztab, int32_t randomized, const float* t_rand,
                            uint64_t seed, uint64_t offset, int32_t white_bkgd,
                            float* comp_rgb, float* depth, float* acc, tnerf_stream_t stream);
int tnerf_render_fused_cam_bf16(const tnerf_mlp_desc* d, const void* packed16, const tnerf_camera* cam,
                                int64_t n_rays, int32_t n_samples,
                                const float* ztab, int32_t randomized, const float* t_rand,
                                uint64_t seed, uint64_t offset, int32_t white_bkgd,
                                float* comp_rgb, float* depth, float* acc, tnerf_stream_t stream);

/* bf16 training (forward + backward of the step body, src/train.py:114-126): activations and activation gradients are
 * rounded to bf16 between layers, every product is accumulated in fp32, the weight gradients leave the kernels in fp32
 * (slabs -> fixed-order reduction -> grads[n_params], exactly like the fp32 path) and the optimizer keeps fp32 master
 * weights.  The stash is organised by tiles of 32 sample slots (n_tiles = n_rays * ceil(n_samples / 32)). */
typedef struct tnerf_bf16_train_plan {
    int64_t n_tiles;
    int64_t stash_bytes;        /* bf16 activations / activation gradients (in the weight-gradient kernel's operand order), ReLU bits, head outputs */
    int64_t slab_floats;        /* fp32 weight-gradient partial slabs (one per wgrad workgroup)                           */
    int64_t job_ints;           /* wgrad job table (int32)                                                                */
    int64_t reduce_ints;        /* slab -> flat-gradient gather table (int32), same format as tnerf_plan_fill's           */
    int64_t n_jobs;
} tnerf_bf16_train_plan;
/* HOST. */
int tnerf_bf16_train_sizes(const tnerf_mlp_desc* d, int64_t n_rays, int32_t n_samples, int32_t n_cu, tnerf_bf16_train_plan* out);
/* HOST. Fill job_table[job_ints] and reduce_table[reduce_ints] (either may be NULL). */
int tnerf_bf16_train_fill(const tnerf_mlp_desc* d, int64_t n_rays, int32_t n_samples, int32_t n_cu,
                          int32_t* job_table, int32_t* reduce_table);
/* The stages of a bf16 training step (arguments as in the fp32 entry points of the same names):
 *   fwd  : comp_rgb [R,3] + stash;  dgrad: g_comp -> dZ records of the stash;  wgrad: stash -> slabs
 * followed by tnerf_wgrad_reduce(slabs, reduce_table, n_params, grads). */
int tnerf_train_fwd_fused_bf16(const tnerf_mlp_desc* d, const void* packed16,
                               const float* rays_o, const float* rays_d, int64_t n_rays, int32_t n_samples,
                               const float* ztab, int32_t randomized, const float* t_rand,
                               uint64_t seed, uint64_t offset, int32_t white_bkgd,
                               float* comp_rgb, void* stash16, tnerf_stream_t stream);
int tnerf_train_dgrad_fused_bf16(const tnerf_mlp_desc* d, const void* packed16,
                                 const float* rays_o, const float* rays_d, int64_t n_rays, int32_t n_samples,
                                 const float* ztab, int32_t randomized, const float* t_rand,
                                 uint64_t seed, uint64_t offset, int32_t white_bkgd,
                                 const float* g_comp, void* stash16, tnerf_stream_t stream);
int tnerf_wgrad_bf16(const tnerf_mlp_desc* d, const void* stash16, int64_t n_tiles,
                     const int32_t* job_table, int64_t n_jobs, float* slabs, tnerf_stream_t stream);
/* tnerf_train_step_fused / tnerf_train_step_fused_cam in bf16 mode: forward, loss, backward -> grads (overwritten). */
int tnerf_train_step_fused_bf16(const tnerf_mlp_desc* d, const void* packed16,
                                const float* rays_o, const float* rays_d, const float* target,
                                int64_t n_rays, int32_t n_samples,
                                const float* ztab, int32_t randomized, const float* t_rand,
                                uint64_t seed, uint64_t offset, int32_t white_bkgd, double loss_denominator,
                                float* comp_rgb, float* g_comp_ws, int64_t g_comp_ws_floats, float* loss_out, void* stash16,
                                const int32_t* job_table, int64_t n_jobs, float* slabs,
                                const int32_t* reduce_table, float* grads, tnerf_stream_t stream);
int tnerf_train_step_fused_cam_bf16(const tnerf_mlp_desc* d, const void* packed16, const tnerf_camera* cam,
                                    const float* pixels, int64_t n_rays, int32_t n_samples,
                                    const float* ztab, int32_t randomized, const float* t_rand,
                                    uint64_t seed, uint64_t offset, int32_t white_bkgd, double loss_denominator,
                                    float* comp_rgb, float* g_comp_ws, int64_t g_comp_ws_floats, float* loss_out, void* stash16,
                                    const int32_t* job_table, int64_t n_jobs, float* slabs,
                                    const int32_t* reduce_table, float* grads, tnerf_stream_t stream);

/* ------------------------------------------------ fp32 chain on the fp16 matrix pipe ("x3": three partial products) */
/* The fused fp32 paths with the MLP's products formed on the fp16 matrix pipe as described at TNERF_FLAG_FP32_MFMA above:
 * weights and activations are carried as two scaled fp16 pieces each, three v_mfma_f32_32x32x16_f16 per k-step into split
 * accumulators (leading products / corrections).  Inputs, outputs, sample bins, encoder, compositing, and the training
 * stash are those of tnerf_render_fused / tnerf_train_fwd_fused; results carry fp32-grade error (an approximation, not
 * bit-equal to an fp32 fma chain).  Requires in_dim = 6L+3.  packed3: the record stream of tnerf_mlp_pack_x3 (sizes / table:
 * tnerf_x3_plan_sizes, tnerf_x3_pack_table; tnerf_bf16_sizes is reused: n_fragments = n_fwd_fragments = 1 KB fragments of
 * the stream, which ends with the per-layer scale records the kernels read). */
int tnerf_x3_plan_sizes(const tnerf_mlp_desc* d, tnerf_bf16_sizes* out);                     /* HOST */
int tnerf_x3_pack_table(const tnerf_mlp_desc* d, int32_t* table);                           /* HOST: table[pack_entries] */
int tnerf_mlp_pack_x3(const tnerf_mlp_desc* d, const float* params, const int32_t* table, void* packed3,
                      tnerf_stream_t stream);
/* The same with the layers' scales chosen for max(max|W_l|, scale_floor).  A caller that hands the stream to tnerf_train_step_dataset
 * passes 16 * lr: that entry point re-scatters the updated weights with the scale chosen before the update (and keeps the same headroom
 * itself from then on), so a layer whose weights are smaller than one optimizer step — a near-zero initialised layer — cannot outgrow
 * the fp16 range of its stream in its first step.  0 = tnerf_mlp_pack_x3. */
int tnerf_mlp_pack_x3_floor(const tnerf_mlp_desc* d, const float* params, const int32_t* table, void* packed3,
                            float scale_floor, tnerf_stream_t stream);
/* The x3 pipe's DOMAIN.  Its scales are one power of two per layer (weights) and per sample (activations); an element keeps its 22
 * bits down to 2^-15 of its block's maximum and loses them below.  counts [4 * (depth + 1)] (device, uint32) receives per layer
 * (index depth = the heads): nonzero weights, weights below 2^-13 max|W_l|, nonzero biases, biases below 2^-14 max|b_l| — against the
 * maxima in packed3's scale records (tnerf_mlp_pack_x3 or the last tnerf_train_step_dataset left them).  A caller that wants
 * fp32-grade results for ANY weights reads the counts back and runs layers where the second count is a sizeable share of the first
 * (the Python binding: more than 1/64) with TNERF_FLAG_FP32_MFMA; well-conditioned networks (every initialisation and every trained
 * TinyNeRF seen here) have shares below 1e-3.  No counterpart in the reference. */
int tnerf_x3_domain_counts(const tnerf_mlp_desc* d, const float* params, const int32_t* table, const void* packed3,
                           uint32_t* counts, tnerf_stream_t stream);
int tnerf_render_fused_x3(const tnerf_mlp_desc* d, const void* packed3,
                          const float* rays_o, const float* rays_d, int64_t n_rays, int32_t n_samples,
                          const float* ztab, int32_t randomized, const float* t_rand,
                          uint64_t seed, uint64_t offset, int32_t white_bkgd,
                          float* comp_rgb, float* depth, float* acc, tnerf_stream_t stream);
int tnerf_render_fused_cam_x3(const tnerf_mlp_desc* d, const void* packed3, const tnerf_camera* cam,
                              int64_t n_rays, int32_t n_samples,
                              const float* ztab, int32_t randomized, const float* t_rand,
                              uint64_t seed, uint64_t offset, int32_t white_bkgd,
                              float* comp_rgb, float* depth, float* acc, tnerf_stream_t stream);
/* Training forward: comp_rgb + the fp32 stash of tnerf_train_fwd_fused (same layout: the dgrad / weight-gradient entry
 * points consume it unchanged). */
int tnerf_train_fwd_fused_x3(const tnerf_mlp_desc* d, const void* packed3,
                             const float* rays_o, const float* rays_d, int64_t n_rays, int32_t n_samples,
                             const float* ztab, int32_t randomized, const float* t_rand,
                             uint64_t seed, uint64_t offset, int32_t white_bkgd,
                             float* comp_rgb, float* stash, int64_t stash_row_stride, tnerf_stream_t stream);

/* tnerf_mlp_fwd / tnerf_mlp_bwd (TinyNeRF.forward and its backward on x[M, in_dim] in memory, src/nerf.py:29-41) on the
 * x3 chain; in_dim = 6L+3 as for every x3 entry point (other input widths: the fp32-MFMA entry points).  Same
 * outputs, same stash, same gradients to fp32 rounding. */
int tnerf_mlp_fwd_x3(const tnerf_mlp_desc* d, const void* packed_x3, const float* x, int64_t n_rows,
                     float* rgb, float* sigma, float* stash, int64_t stash_row_stride, tnerf_stream_t stream);
int tnerf_mlp_bwd_x3(const tnerf_mlp_desc* d, const void* packed_x3, int64_t n_rows,
                     const float* d_rgb, const float* d_sigma, float* stash, int64_t stash_row_stride,
                     const int32_t* job_table, int64_t n_jobs, float* slabs,
                     const int32_t* reduce_table, float* grads, tnerf_stream_t stream);

/* tnerf_train_dgrad_fused on the x3 chain: reads the backward record stream of `packed_x3` (heads^T and the
 * transposed hidden layers, which tnerf_mlp_pack_x3 writes behind the forward stream) and fills the same dZ rows. */
int tnerf_train_dgrad_fused_x3(const tnerf_mlp_desc* d, const void* packed_x3,
                               const float* rays_o, const float* rays_d, int64_t n_rays, int32_t n_samples,
                               const float* ztab, int32_t randomized, const float* t_rand,
                               uint64_t seed, uint64_t offset, int32_t white_bkgd,
                               const float* g_comp, float* stash, int64_t stash_row_stride, tnerf_stream_t stream);

/* torch.optim.Adam(lr, betas, eps, weight_decay=0) on the flat buffers   [src/train.py:80,127]
 * step = 1-based step count t used for the bias corrections; grad_scale multiplies the gradient
 * first (1/world_size after an all-reduce SUM of already globally-normalised shards = 1). */
int tnerf_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                    int64_t n, float lr, float beta1, float beta2, float eps, int64_t step,
                    float grad_scale, tnerf_stream_t stream);

/* ------------------------------------------- any width, and the gradient w.r.t. the input: layer by layer */
/* TinyNeRF.forward and its backward for the shapes the register-resident kernels above do not cover (hidden > 256, in_dim > 64;
 * the reference's constructor takes any, src/nerf.py:10-27) and, for ANY shape, with the gradient w.r.t. the network input
 * (src/nerf.py:29-41 is differentiable in x although the reference's training never asks: its points carry no grad).  Not the hot
 * path: one fp32 library SGEMM per layer (hipBLAS, atomics off: deterministic) between small elementwise kernels.
 *   d            : in_dim, hidden <= 4096, depth <= 64, skip_at as above (flags ignored)
 *   blas_handle  : a hipblasHandle_t of the CALLER (the library creates none); its stream / atomics / pointer mode are set for the
 *                  call and restored.  libhipblas.so is dlopen'ed on first use.
 *   params/grads : HOST arrays of 2*depth+4 device pointers in state_dict order (layers.i.weight [out,in], layers.i.bias, sigma.0.*,
 *                  rgb.0.*), each tensor contiguous — no flat buffer, no packed copy.  grads are OVERWRITTEN.
 *   acts         : [depth][n_rows][hidden] post-ReLU activations, filled by the forward, read by the backward
 *   scratch      : the backward's gradient buffers; dx: NULL or [n_rows, in_dim] (overwritten with dL/dx)
 * Undersized acts / scratch: TNERF_ESMALL. */
int64_t tnerf_mlp_generic_acts_floats(const tnerf_mlp_desc* d, int64_t n_rows);      /* HOST */
int64_t tnerf_mlp_generic_scratch_floats(const tnerf_mlp_desc* d, int64_t n_rows);   /* HOST */
int tnerf_mlp_fwd_generic(const tnerf_mlp_desc* d, void* blas_handle, const float* const* params, const float* x, int64_t n_rows,
                          float* rgb, float* sigma, float* acts, int64_t acts_floats, tnerf_stream_t stream);
int tnerf_mlp_bwd_generic(const tnerf_mlp_desc* d, void* blas_handle, const float* const* params, const float* x, int64_t n_rows,
                          const float* rgb, const float* sigma, const float* d_rgb, const float* d_sigma,
                          const float* acts, int64_t acts_floats, float* scratch, int64_t scratch_floats,
                          float* const* grads, float* dx, tnerf_stream_t stream);

/* --------------------------------------------------------------------------- RCCL (optional) */
/* Sum the flat gradient over ranks with RCCL (ncclAllReduce, ncclSum) on `stream`.
 * `comm` is an ncclComm_t created by tnerf_comm_init_rank.  librccl.so is dlopen'ed on first use. */
int tnerf_comm_unique_id(void* id128 /* 128 bytes out */);
int tnerf_comm_init_rank(const void* id128, int32_t n_ranks, int32_t rank, void** comm_out);
int tnerf_comm_destroy(void* comm);
int tnerf_allreduce_grads(void* comm, float* grads, int64_t n, tnerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TNERF_H */
