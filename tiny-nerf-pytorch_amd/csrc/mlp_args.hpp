// Kernel argument blocks shared between the kernel files and the C-ABI entry points.
#pragma once
#include "dev_common.hpp"

struct FwdArgs {
    MlpLayout L;
    const float* packed;
    // fused mode
    RaySource rs; int64_t R; SampleArgs sa; int32_t white;
    float* comp; float* depth; float* acc;
    // mlp-only mode
    const float* x; int64_t M; float* rgb_out; float* sigma_out;
    // training stash (NULL for inference)
    float* stash; int64_t Mp;
    // training step: loss gradient written by the ray's wave (ray_ws == NULL: no loss)
    LossArgs loss;
    // diagnostic builds only (-DTN_STAMPS): s_memtime stamps of the first ray of each workgroup
    unsigned long long* stamps;
};

struct BwdArgs {
    MlpLayout L;
    const float* packed;
    float* stash; int64_t Mp;
    // fused mode
    RaySource rs; int64_t R; SampleArgs sa; int32_t white;
    const float* g_comp; int32_t g_stride;      // dL/dcomp_rgb of ray r at g_comp[g_stride * r + c] (3: [R,3] tensor, 4: ray_ws)
    // mlp-only mode
    int64_t M; const float* d_rgb; const float* d_sigma;
};

// mlp_fwd.hip
int tn_launch_fwd(const FwdArgs& a, bool fused, bool train, int64_t units, hipStream_t stream, const char* who);
int tn_fused_args(const char* who, FwdArgs& a, const tnerf_mlp_desc* d, const float* packed, const RaySource& rs,
                  int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                  uint64_t seed, uint64_t offset, int32_t white);
int tn_camera_source(const char* who, const tnerf_camera* cam, int64_t R, RaySource* rs);
inline RaySource tn_table_source(const float* rays_o, const float* rays_d) { return RaySource{rays_o, rays_d, nullptr, nullptr, 0, 0, 0, 0.0f}; }
// mlp_bwd.hip
int tn_launch_mlp_bwd(const BwdArgs& a, hipStream_t stream);
int tn_launch_train_bwd(const BwdArgs& a, hipStream_t stream);
// wgrad.hip
// mode 0: fp32-MFMA body, 1: x3 body (three fp16 partial products), 2: whichever the stash's pipe tag names; bounds = the stash's bound words
int tn_launch_wgrad(const float* stash, int64_t stash_rows, int64_t M, const int32_t* jobs, int64_t n_jobs, float* slabs, int64_t* step_inc, hipStream_t stream, int mode,
                    const float* bounds);
int tn_launch_reduce(const float* slabs, const int32_t* reduce_table, int64_t n_params, float* grads, hipStream_t stream);
// Dataset mode: the device step counter and the Philox counters consumed per step (zero = per-call mode)
struct TnStepRef { int64_t* step; uint64_t per_step; };
// packed3 != NULL (and not TNERF_FLAG_FP32_MFMA): the forward runs on the x3 chain kernel (mlpx3.hip), same stash
int tn_step32_core(const char* who, const tnerf_mlp_desc* d, const float* packed, const void* packed3, const RaySource& rs, const TnStepRef& sr,
                   const LossArgs& loss, int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                   uint64_t seed, uint64_t offset, int32_t white, float* comp_rgb, float* stash, int64_t Mp,
                   const int32_t* job_table, int64_t n_jobs, float* slabs, hipStream_t stream);
int tnx3_mlp_dgrad(const char* who, const BwdArgs& b, const tnerf_mlp_desc* d, const void* packed3, hipStream_t stream);
int tnx3_train_dgrad(const char* who, const BwdArgs& b, const tnerf_mlp_desc* d, const void* packed3, hipStream_t stream, bool heads_done = false);
int tnx3_train_fwd(const char* who, const tnerf_mlp_desc* d, const void* packed3, const RaySource& rs, const TnStepRef& sr,
                   const LossArgs& loss, int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                   uint64_t seed, uint64_t offset, int32_t white, float* comp, float* stash, int64_t Mp, hipStream_t stream, bool* heads_done = nullptr);
// The finishing kernel of a step: [slab reduction -> grads] [+ loss = inv_denom * sum ray_ws[.,3]] [+ Adam + re-pack of the updated weights]
struct FinishArgs {
    // reduce (slabs == NULL: grads already hold the gradient, e.g. after the all-reduce)
    const float* slabs; const int32_t* reduce_table;
    int64_t n_params; float* grads;
    // loss (loss_out == NULL: none)
    const float* ray_ws; int64_t R; float inv_denom; float* loss_out;
    // Adam (params == NULL: none).  t = *step (already incremented by the weight-gradient kernel) or step_host
    float* params; float* m; float* v; float lr, b1, b2, eps, gscale; const int64_t* step; int64_t step_host;
    // re-pack: parameter i goes to packed position scatter[i * width + k] (< 0: none); positions < bf16_elems are bf16
    // elements of the fragment stream, the rest fp32 (biases) counted from bias_base
    const int32_t* scatter; int32_t width; void* packed; int64_t bf16_elems; int64_t bias_off_bytes;
    // second re-pack target: the x3 record stream (element i carries piece (i >> 9) mod TX_NP of its parameter times the layer's
    // scale, record [3] of the TX_META floats at n3.meta_off; biases behind x3_elems).  The kernel's last workgroup publishes the
    // scale records (tx_stats_final, post = 1).
    const int32_t* scatter3; int32_t width3; void* packed3; int64_t x3_elems; NetX3 n3;
    float scale_floor;      // scatter3: the floor under max|W| when the next scales are chosen (tx_stats_final)
    int32_t fold_stats;     // scatter3: the finishing kernel's last workgroup publishes the scale records (small networks; otherwise the caller launches tnx3_launch_stats(post = 1) behind it)
};
#define TN_FOLD_STATS_BLOCKS 512
int tnx3_launch_stats(const NetX3& n, const float* params, const int32_t* table, void* packed3, int post, hipStream_t stream, float scale_floor);
int tn_launch_finish(const FinishArgs& f, hipStream_t stream);
