// Kernel argument blocks and launchers of the bf16 mode.
#pragma once
#include "mlp_args.hpp"

struct Fwd16Args {
    Net16 n;
    const unsigned char* packed;
    RaySource rs; int64_t R; SampleArgs sa; int32_t white;
    float* comp; float* depth; float* acc;
    // training: tile-organised stash (tnerf_internal.h), n_tiles = R * ceil(S/32)
    unsigned char* stash; int64_t n_tiles;
    // training step: loss gradient written by the ray's wave (ray_ws == NULL: none)
    LossArgs loss;
    // dgrad only: dL/dcomp_rgb of ray r at g_comp[g_stride * r + c]
    const float* g_comp; int32_t g_stride;
};

// mlp16_fwd.hip
int tn16_launch_fwd(const Fwd16Args& a, bool train, hipStream_t stream, const char* who);
int tn16_fused_args(const char* who, Fwd16Args& a, const tnerf_mlp_desc* d, const void* packed16, const RaySource& rs, int64_t R, int32_t S,
                    const float* ztab, int32_t randomized, const float* t_rand, uint64_t seed, uint64_t offset, int32_t white);
// mlp16_bwd.hip
int tn16_launch_dgrad(const Fwd16Args& a, hipStream_t stream, const char* who);
int tn16_launch_wgrad(const Net16& n, const unsigned char* stash, const int32_t* jobs, int64_t n_jobs, float* slabs, int64_t* step_inc, hipStream_t stream);
int tn_step16_core(const char* who, const tnerf_mlp_desc* d, const void* packed16, const RaySource& rs, const TnStepRef& sr,
                   const LossArgs& loss, int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                   uint64_t seed, uint64_t offset, int32_t white, float* comp_rgb, void* stash16,
                   const int32_t* job_table, int64_t n_jobs, float* slabs, hipStream_t stream);
