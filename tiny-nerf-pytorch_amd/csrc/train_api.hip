// C-ABI entry points that chain several kernels: backward passes, the whole train step, RCCL.
#include <dlfcn.h>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include "mlp_args.hpp"

int tn_launch_reduce(const float* slabs, const int32_t* reduce_table, int64_t n_params, float* grads, hipStream_t stream) {
    FinishArgs f{};
    f.slabs = slabs; f.reduce_table = reduce_table; f.n_params = n_params; f.grads = grads;
    return tn_launch_finish(f, stream);
}

static int bwd_common_check(const char* who, const float* packed, float* stash, int64_t Mp, int64_t M, const int32_t* job_table,
                            int64_t n_jobs, float* slabs, const int32_t* reduce_table, float* grads) {
    if (!packed || !stash || Mp < M || !job_table || n_jobs < 1 || !slabs || !reduce_table || !grads) {
        tn_set_error("%s: packed=%p stash=%p Mp=%lld M=%lld jobs=%p n_jobs=%lld slabs=%p reduce=%p grads=%p", who, (const void*)packed,
                     (void*)stash, (long long)Mp, (long long)M, (const void*)job_table, (long long)n_jobs, (void*)slabs,
                     (const void*)reduce_table, (void*)grads);
        return TNERF_EINVAL;
    }
    return TNERF_OK;
}

extern "C" int tnerf_mlp_bwd(const tnerf_mlp_desc* d, const float* packed, int64_t M, const float* d_rgb, const float* d_sigma,
                             float* stash, int64_t Mp, const int32_t* job_table, int64_t n_jobs, float* slabs,
                             const int32_t* reduce_table, float* grads, tnerf_stream_t stream) {
    BwdArgs a{};
    int rc = tn_build_layout(d, &a.L); if (rc) return rc;
    if (M < 1 || !d_rgb || !d_sigma) { tn_set_error("tnerf_mlp_bwd: M=%lld d_rgb=%p d_sigma=%p", (long long)M, (const void*)d_rgb, (const void*)d_sigma); return TNERF_EINVAL; }
    rc = bwd_common_check("tnerf_mlp_bwd", packed, stash, Mp, M, job_table, n_jobs, slabs, reduce_table, grads); if (rc) return rc;
    a.packed = packed; a.stash = stash; a.Mp = Mp; a.M = M; a.d_rgb = d_rgb; a.d_sigma = d_sigma;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = tn_launch_mlp_bwd(a, s))) return rc;
    // (the x3 weight-gradient kernel scales its operands by bounds only the x3 chain kernels leave in the stash)
    if ((rc = tn_launch_wgrad(stash, a.L.stash_rows, M, job_table, n_jobs, slabs, nullptr, s, 0, stash + TN_BOUND_OFF(a.L, Mp)))) return rc;
    return tn_launch_reduce(slabs, reduce_table, a.L.n_params, grads, s);
}

extern "C" int tnerf_mlp_bwd_x3(const tnerf_mlp_desc* d, const void* packed_x3, int64_t M, const float* d_rgb, const float* d_sigma,
                                float* stash, int64_t Mp, const int32_t* job_table, int64_t n_jobs, float* slabs,
                                const int32_t* reduce_table, float* grads, tnerf_stream_t stream) {
    const char* who = "tnerf_mlp_bwd_x3";
    BwdArgs a{};
    int rc = tn_build_layout(d, &a.L); if (rc) return rc;
    if (M < 1 || !d_rgb || !d_sigma) { tn_set_error("%s: M=%lld d_rgb=%p d_sigma=%p", who, (long long)M, (const void*)d_rgb, (const void*)d_sigma); return TNERF_EINVAL; }
    rc = bwd_common_check(who, reinterpret_cast<const float*>(packed_x3), stash, Mp, M, job_table, n_jobs, slabs, reduce_table, grads); if (rc) return rc;
    a.packed = nullptr; a.stash = stash; a.Mp = Mp; a.M = M; a.d_rgb = d_rgb; a.d_sigma = d_sigma;
    hipStream_t s = (hipStream_t)stream;
    if ((rc = tnx3_mlp_dgrad(who, a, d, packed_x3, s))) return rc;
    if ((rc = tn_launch_wgrad(stash, a.L.stash_rows, M, job_table, n_jobs, slabs, nullptr, s, 1, stash + TN_BOUND_OFF(a.L, Mp)))) return rc;
    return tn_launch_reduce(slabs, reduce_table, a.L.n_params, grads, s);
}

// dgrad + wgrad (+ slab reduction when reduce_table != NULL).  g_comp: dL/dcomp_rgb with row stride g_stride.
static int train_bwd_impl(const char* who, const tnerf_mlp_desc* d, const float* packed, const void* packed3, const RaySource& rs, const TnStepRef& sr,
                          int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand, uint64_t seed,
                          uint64_t offset, int32_t white, const float* g_comp, int32_t g_stride, float* stash, int64_t Mp,
                          const int32_t* job_table, int64_t n_jobs, float* slabs, const int32_t* reduce_table, float* grads, hipStream_t s,
                          bool heads_done = false) {
    FwdArgs f{};
    int rc = tn_fused_args(who, f, d, packed, rs, R, S, ztab, randomized, t_rand, seed, offset, white); if (rc) return rc;
    if (R < 1 || !g_comp) { tn_set_error("%s: R=%lld g_comp=%p", who, (long long)R, (const void*)g_comp); return TNERF_EINVAL; }
    static const int32_t no_table = 0;
    rc = bwd_common_check(who, packed, stash, Mp, R * S, job_table, n_jobs, slabs, reduce_table ? reduce_table : &no_table, grads ? grads : slabs); if (rc) return rc;
    BwdArgs a{};
    a.L = f.L; a.packed = packed; a.stash = stash; a.Mp = Mp; a.rs = f.rs; a.R = R; a.sa = f.sa;
    a.sa.step = sr.step; a.sa.per_step = sr.per_step;
    a.white = white; a.g_comp = g_comp; a.g_stride = g_stride;
    const bool x3 = packed3 && !(d->flags & TNERF_FLAG_FP32_MFMA);
    if (x3) rc = tnx3_train_dgrad(who, a, d, packed3, s, heads_done);
    else    rc = tn_launch_train_bwd(a, s);
    if (rc) return rc;
    if ((rc = tn_launch_wgrad(stash, a.L.stash_rows, R * S, job_table, n_jobs, slabs, sr.step, s, x3 ? 1 : 0, stash + TN_BOUND_OFF(a.L, Mp)))) return rc;
    if (!reduce_table) return TNERF_OK;
    return tn_launch_reduce(slabs, reduce_table, a.L.n_params, grads, s);
}

extern "C" int tnerf_train_bwd_fused(const tnerf_mlp_desc* d, const float* packed, const float* rays_o, const float* rays_d,
                                     int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                                     uint64_t seed, uint64_t offset, int32_t white, const float* g_comp, float* stash, int64_t Mp,
                                     const int32_t* job_table, int64_t n_jobs, float* slabs, const int32_t* reduce_table,
                                     float* grads, const void* packed_x3, tnerf_stream_t stream) {
    return train_bwd_impl("tnerf_train_bwd_fused", d, packed, packed_x3, tn_table_source(rays_o, rays_d), TnStepRef{}, R, S, ztab, randomized, t_rand, seed, offset, white,
                          g_comp, 3, stash, Mp, job_table, n_jobs, slabs, reduce_table, grads, (hipStream_t)stream);
}

extern "C" int tnerf_train_dgrad_fused(const tnerf_mlp_desc* d, const float* packed, const float* rays_o, const float* rays_d,
                                       int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                                       uint64_t seed, uint64_t offset, int32_t white, const float* g_comp, float* stash, int64_t Mp,
                                       tnerf_stream_t stream) {
    FwdArgs f{};
    int rc = tn_fused_args("tnerf_train_dgrad_fused", f, d, packed, tn_table_source(rays_o, rays_d), R, S, ztab, randomized, t_rand, seed, offset, white);
    if (rc) return rc;
    if (R < 1 || !g_comp || !stash || Mp < R * S) { tn_set_error("tnerf_train_dgrad_fused: R=%lld g_comp=%p stash=%p Mp=%lld", (long long)R, (const void*)g_comp, (void*)stash, (long long)Mp); return TNERF_EINVAL; }
    BwdArgs a{};
    a.L = f.L; a.packed = packed; a.stash = stash; a.Mp = Mp; a.rs = f.rs; a.R = R; a.sa = f.sa;
    a.white = white; a.g_comp = g_comp; a.g_stride = 3;
    return tn_launch_train_bwd(a, (hipStream_t)stream);
}

extern "C" int tnerf_train_dgrad_fused_x3(const tnerf_mlp_desc* d, const void* packed_x3, const float* rays_o, const float* rays_d,
                                          int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                                          uint64_t seed, uint64_t offset, int32_t white, const float* g_comp, float* stash, int64_t Mp,
                                          tnerf_stream_t stream) {
    const char* who = "tnerf_train_dgrad_fused_x3";
    FwdArgs f{};
    int rc = tn_fused_args(who, f, d, reinterpret_cast<const float*>(packed_x3), tn_table_source(rays_o, rays_d), R, S, ztab, randomized, t_rand, seed, offset, white);
    if (rc) return rc;
    if (R < 1 || !g_comp || !stash || Mp < R * S) { tn_set_error("%s: R=%lld g_comp=%p stash=%p Mp=%lld", who, (long long)R, (const void*)g_comp, (void*)stash, (long long)Mp); return TNERF_EINVAL; }
    BwdArgs a{};
    a.L = f.L; a.packed = nullptr; a.stash = stash; a.Mp = Mp; a.rs = f.rs; a.R = R; a.sa = f.sa;
    a.white = white; a.g_comp = g_comp; a.g_stride = 3;
    return tnx3_train_dgrad(who, a, d, packed_x3, (hipStream_t)stream);
}

extern "C" int tnerf_wgrad(const tnerf_mlp_desc* d, const float* stash, int64_t Mp, int64_t M, const int32_t* job_table, int64_t n_jobs,
                           float* slabs, tnerf_stream_t stream) {
    MlpLayout L; int rc = tn_build_layout(d, &L); if (rc) return rc;
    if (!stash || Mp < M || M < 1 || !job_table || n_jobs < 1 || !slabs) { tn_set_error("tnerf_wgrad: bad arguments"); return TNERF_EINVAL; }
    // which forward filled the stash is not this call's to know: the kernel reads the stash's pipe tag (tnerf_internal.h TNB_TAG)
    return tn_launch_wgrad(stash, L.stash_rows, M, job_table, n_jobs, slabs, nullptr, (hipStream_t)stream, 2, stash + TN_BOUND_OFF(L, Mp));
}

extern "C" int tnerf_wgrad_reduce(const float* slabs, const int32_t* reduce_table, int64_t n_params, float* grads, tnerf_stream_t stream) {
    if (!slabs || !reduce_table || n_params < 1 || !grads) { tn_set_error("tnerf_wgrad_reduce: bad arguments"); return TNERF_EINVAL; }
    return tn_launch_reduce(slabs, reduce_table, n_params, grads, (hipStream_t)stream);
}

static int tn_train_fwd_impl(const char* who, const tnerf_mlp_desc* d, const float* packed, const RaySource& rs, const TnStepRef& sr,
                             const LossArgs& loss, int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                             uint64_t seed, uint64_t offset, int32_t white, float* comp, float* stash, int64_t Mp, hipStream_t stream) {
    FwdArgs a{};
    int rc = tn_fused_args(who, a, d, packed, rs, R, S, ztab, randomized, t_rand, seed, offset, white);
    if (rc) return rc;
    if (!comp || !stash || Mp < R * S) { tn_set_error("%s: comp=%p stash=%p Mp=%lld < R*S=%lld", who, (void*)comp, (void*)stash, (long long)Mp, (long long)(R * S)); return TNERF_EINVAL; }
    a.comp = comp; a.stash = stash; a.Mp = Mp; a.loss = loss;
    a.sa.step = sr.step; a.sa.per_step = sr.per_step;
    return tn_launch_fwd(a, true, true, R, stream, who);
}

// forward (+ loss gradient per ray) -> dgrad -> wgrad: the step up to the slabs.  Shared by the per-call entry points below
// and by tnerf_train_step_dataset (step_api.hip).
int tn_step32_core(const char* who, const tnerf_mlp_desc* d, const float* packed, const void* packed3, const RaySource& rs, const TnStepRef& sr,
                   const LossArgs& loss, int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                   uint64_t seed, uint64_t offset, int32_t white, float* comp_rgb, float* stash, int64_t Mp,
                   const int32_t* job_table, int64_t n_jobs, float* slabs, hipStream_t stream) {
    int rc;
    bool heads_done = false;      // x3 tile route: the forward's compositing kernel also ran the compositing backward (tnx3_launch_fwd)
    if (packed3 && !(d->flags & TNERF_FLAG_FP32_MFMA))
        rc = tnx3_train_fwd(who, d, packed3, rs, sr, loss, R, S, ztab, randomized, t_rand, seed, offset, white, comp_rgb, stash, Mp, stream, &heads_done);
    else
        rc = tn_train_fwd_impl(who, d, packed, rs, sr, loss, R, S, ztab, randomized, t_rand, seed, offset, white, comp_rgb, stash, Mp, stream);
    if (rc) return rc;
    return train_bwd_impl(who, d, packed, packed3, rs, sr, R, S, ztab, randomized, t_rand, seed, offset, white, loss.ray_ws, 4, stash, Mp,
                          job_table, n_jobs, slabs, nullptr, nullptr, stream, heads_done);
}

static int train_step_impl(const char* who, const tnerf_mlp_desc* d, const float* packed, const RaySource& rs, const float* target,
                           const int64_t* target_index, int64_t R, int32_t S, const float* ztab, int32_t randomized,
                           const float* t_rand, uint64_t seed, uint64_t offset, int32_t white, double loss_denominator,
                           float* comp_rgb, float* g_comp_ws, int64_t ws_floats, float* loss_out, float* stash, int64_t Mp,
                           const int32_t* job_table, int64_t n_jobs, float* slabs, const int32_t* reduce_table,
                           float* grads, const void* packed_x3, hipStream_t stream) {
    if (!target || !comp_rgb || !g_comp_ws || !loss_out || !(loss_denominator > 0.0) || R < 1) {
        tn_set_error("%s: target=%p comp=%p g_ws=%p loss=%p denom=%g R=%lld", who, (const void*)target, (void*)comp_rgb,
                     (void*)g_comp_ws, (void*)loss_out, loss_denominator, (long long)R);
        return TNERF_EINVAL;
    }
    if (!reduce_table || !grads) { tn_set_error("%s: reduce_table=%p grads=%p", who, (const void*)reduce_table, (void*)grads); return TNERF_EINVAL; }
    int rc = tn_check_ray_ws(who, R, ws_floats); if (rc) return rc;
    const LossArgs loss{target, target_index, (float)(1.0 / loss_denominator), g_comp_ws, nullptr};
    rc = tn_step32_core(who, d, packed, packed_x3, rs, TnStepRef{}, loss, R, S, ztab, randomized, t_rand, seed, offset, white, comp_rgb, stash, Mp,
                            job_table, n_jobs, slabs, stream);
    if (rc) return rc;
    FinishArgs f{};
    f.slabs = slabs; f.reduce_table = reduce_table; f.n_params = tnerf_param_count(d); f.grads = grads;
    f.ray_ws = g_comp_ws; f.R = R; f.inv_denom = loss.inv_denom; f.loss_out = loss_out;
    return tn_launch_finish(f, stream);
}

extern "C" int tnerf_train_step_fused(const tnerf_mlp_desc* d, const float* packed, const float* rays_o, const float* rays_d,
                                      const float* target, int64_t R, int32_t S, const float* ztab, int32_t randomized,
                                      const float* t_rand, uint64_t seed, uint64_t offset, int32_t white, double loss_denominator,
                                      float* comp_rgb, float* g_comp_ws, int64_t g_comp_ws_floats, float* loss_out, float* stash, int64_t Mp,
                                      const int32_t* job_table, int64_t n_jobs, float* slabs, const int32_t* reduce_table,
                                      float* grads, const void* packed_x3, tnerf_stream_t stream) {
    return train_step_impl("tnerf_train_step_fused", d, packed, tn_table_source(rays_o, rays_d), target, nullptr, R, S, ztab, randomized,
                           t_rand, seed, offset, white, loss_denominator, comp_rgb, g_comp_ws, g_comp_ws_floats, loss_out, stash, Mp, job_table, n_jobs,
                           slabs, reduce_table, grads, packed_x3, (hipStream_t)stream);
}

extern "C" int tnerf_train_step_fused_cam(const tnerf_mlp_desc* d, const float* packed, const tnerf_camera* cam, const float* pixels,
                                          int64_t R, int32_t S, const float* ztab, int32_t randomized, const float* t_rand,
                                          uint64_t seed, uint64_t offset, int32_t white, double loss_denominator, float* comp_rgb,
                                          float* g_comp_ws, int64_t g_comp_ws_floats, float* loss_out, float* stash, int64_t Mp, const int32_t* job_table,
                                          int64_t n_jobs, float* slabs, const int32_t* reduce_table, float* grads,
                                          const void* packed_x3, tnerf_stream_t stream) {
    RaySource rs;
    int rc = tn_camera_source("tnerf_train_step_fused_cam", cam, R, &rs); if (rc) return rc;
    if (!cam->pix_index) { tn_set_error("tnerf_train_step_fused_cam: pix_index is required (it also selects the target pixels)"); return TNERF_EINVAL; }
    return train_step_impl("tnerf_train_step_fused_cam", d, packed, rs, pixels, cam->pix_index, R, S, ztab, randomized, t_rand, seed,
                           offset, white, loss_denominator, comp_rgb, g_comp_ws, g_comp_ws_floats, loss_out, stash, Mp, job_table, n_jobs, slabs,
                           reduce_table, grads, packed_x3, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------ RCCL
// librccl.so is loaded lazily so that single-GPU users (and the CPU-only symbol test) never need it.
namespace {
typedef struct { char internal[128]; } nccl_uid;
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(void**, int, nccl_uid, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_errstr)(int);
struct Rccl { void* h; fn_get_uid get_uid; fn_init_rank init_rank; fn_destroy destroy; fn_allreduce allreduce; fn_errstr errstr; };

// The dlopen'ed function table is filled exactly once (std::call_once) and never changes afterwards.
static Rccl* rccl() {
    static Rccl r{};
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
        if (r.h) {
            r.get_uid = (fn_get_uid)dlsym(r.h, "ncclGetUniqueId");
            r.init_rank = (fn_init_rank)dlsym(r.h, "ncclCommInitRank");
            r.destroy = (fn_destroy)dlsym(r.h, "ncclCommDestroy");
            r.allreduce = (fn_allreduce)dlsym(r.h, "ncclAllReduce");
            r.errstr = (fn_errstr)dlsym(r.h, "ncclGetErrorString");
        }
    });
    if (!r.h || !r.get_uid || !r.init_rank || !r.destroy || !r.allreduce) { tn_set_error("RCCL (librccl.so) could not be loaded: %s", dlerror()); return nullptr; }
    return &r;
}
static int rccl_rc(Rccl* r, int code, const char* who) {
    if (code == 0) return TNERF_OK;
    tn_set_error("%s: RCCL error %d (%s)", who, code, r->errstr ? r->errstr(code) : "?");
    return 10000 + code;
}
}  // namespace

extern "C" int tnerf_comm_unique_id(void* id128) {
    if (!id128) { tn_set_error("tnerf_comm_unique_id: NULL"); return TNERF_EINVAL; }
    Rccl* r = rccl(); if (!r) return TNERF_EUNSUPPORTED;
    return rccl_rc(r, r->get_uid((nccl_uid*)id128), "ncclGetUniqueId");
}
extern "C" int tnerf_comm_init_rank(const void* id128, int32_t n_ranks, int32_t rank, void** comm_out) {
    if (!id128 || !comm_out || n_ranks < 1 || rank < 0 || rank >= n_ranks) { tn_set_error("tnerf_comm_init_rank: bad arguments"); return TNERF_EINVAL; }
    Rccl* r = rccl(); if (!r) return TNERF_EUNSUPPORTED;
    nccl_uid u; memcpy(&u, id128, sizeof(u));
    return rccl_rc(r, r->init_rank(comm_out, n_ranks, u, rank), "ncclCommInitRank");
}
extern "C" int tnerf_comm_destroy(void* comm) {
    if (!comm) return TNERF_OK;
    Rccl* r = rccl(); if (!r) return TNERF_EUNSUPPORTED;
    return rccl_rc(r, r->destroy(comm), "ncclCommDestroy");
}
extern "C" int tnerf_allreduce_grads(void* comm, float* grads, int64_t n, tnerf_stream_t stream) {
    if (!comm || !grads || n < 1) { tn_set_error("tnerf_allreduce_grads: comm=%p grads=%p n=%lld", comm, (void*)grads, (long long)n); return TNERF_EINVAL; }
    Rccl* r = rccl(); if (!r) return TNERF_EUNSUPPORTED;
    return rccl_rc(r, r->allreduce(grads, grads, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, (hipStream_t)stream), "ncclAllReduce");
}
