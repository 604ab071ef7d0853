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
    // diagnostic builds only (-DTN_STAMPS): s_memtime stamps of the first ray of each workgroup
    unsigned long long* stamps;
};

struct BwdArgs {
    MlpLayout L;
    const float* packed;
    float* stash; int64_t Mp;
    // fused mode
    RaySource rs; int64_t R; SampleArgs sa; int32_t white;
    const float* g_comp;
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
int tn_launch_wgrad(const float* stash, int64_t stash_rows, int64_t M, const int32_t* jobs, int64_t n_jobs, float* slabs, hipStream_t stream);
int tn_launch_reduce(const float* slabs, const int32_t* reduce_table, int64_t n_params, float* grads, hipStream_t stream);
int tn_launch_loss_grad(const float* comp, const float* target, const int64_t* target_index, int64_t R, double denom, float* g_comp, float* loss_out, hipStream_t stream);
