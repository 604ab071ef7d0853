// The whole train step as ONE call on device-resident state, and hipGraph capture / replay of it.
//
// tnerf_train_step_dataset runs the body of the reference loop (src/train.py:106-128) without any per-step host input:
// the image index (step % N), the pixel draw (torch.randint there) and the jitter draw (rand_like there) come from a
// device-side step counter + Philox inside the kernels, the loss gradient is formed by the forward kernel, the step counter
// advances in the weight-gradient kernel, and slab reduction + Adam + re-packing of the updated weights are one finishing
// kernel: 4 launches per step on one GPU, nothing in their arguments changes from step to step — so the step can be
// captured once into a hipGraph and replayed with a single host call.
#include "mlp16_args.hpp"

static int check_step_args(const tnerf_step_args* a) {
    if (!a) { tn_set_error("tnerf_train_step_dataset: NULL args"); return TNERF_EINVAL; }
    const bool p1 = a->phases & TNERF_PHASE_GRADIENT, p2 = a->phases & TNERF_PHASE_REDUCE, p3 = a->phases & TNERF_PHASE_UPDATE;
    if (!(p1 || p2 || p3) || (a->precision != 0 && a->precision != 1)) {
        tn_set_error("tnerf_train_step_dataset: phases=%d precision=%d", a->phases, a->precision); return TNERF_EINVAL; }
    if (p1 && (!a->poses || !a->pixels || a->n_images < 1 || a->H < 1 || a->W < 1 || !(a->focal != 0.0f) || a->n_rays < 1 ||
               a->ray_first < 0 || a->n_rays_global < a->ray_first + a->n_rays || a->n_samples < 1 || !a->ztab || !a->step ||
               !a->packed || !a->comp_rgb || !a->ray_ws || !a->stash || !a->job_table || a->n_jobs < 1 || !a->slabs ||
               !(a->loss_denominator > 0.0) || (int64_t)a->H * a->W >= ((int64_t)1 << 31))) {
        tn_set_error("tnerf_train_step_dataset: bad gradient-phase argument (poses=%p pixels=%p N=%d HxW=%dx%d rays=%lld+%lld/%lld S=%d "
                     "ztab=%p step=%p packed=%p comp=%p ray_ws=%p stash=%p jobs=%p/%lld slabs=%p denom=%g)", (const void*)a->poses,
                     (const void*)a->pixels, a->n_images, a->H, a->W, (long long)a->ray_first, (long long)a->n_rays, (long long)a->n_rays_global,
                     a->n_samples, (const void*)a->ztab, (void*)a->step, a->packed, (void*)a->comp_rgb, (void*)a->ray_ws, a->stash,
                     (const void*)a->job_table, (long long)a->n_jobs, (void*)a->slabs, a->loss_denominator);
        return TNERF_EINVAL;
    }
    if (p1) {
        int rc = tn_check_ray_ws("tnerf_train_step_dataset", a->n_rays, a->ray_ws_floats); if (rc) return rc;
        rc = a->precision == 0 ? tn_check_stash32("tnerf_train_step_dataset", &a->desc, a->n_rays * a->n_samples, a->stash_row_stride, a->stash_capacity)
                               : tn_check_stash16("tnerf_train_step_dataset", &a->desc, a->n_rays, a->n_samples, a->stash_capacity);
        if (rc) return rc;
    }
    if (p2 && (!a->slabs || !a->reduce_table || !a->grads)) { tn_set_error("tnerf_train_step_dataset: reduce phase needs slabs, reduce_table, grads"); return TNERF_EINVAL; }
    if (p3 && (!a->grads || !a->params || !a->exp_avg || !a->exp_avg_sq || !a->step || !(a->lr >= 0.0f) ||
               (a->scatter_table && (a->scatter_width < 1 || !a->packed)) ||
               (a->precision == 0 && a->packed_x3 && a->scatter_x3 && !a->pack_x3))) {
        tn_set_error("tnerf_train_step_dataset: update phase needs grads, params, exp_avg, exp_avg_sq, step (packed with a scatter table; pack_x3 with scatter_x3)"); return TNERF_EINVAL; }
    return TNERF_OK;
}

extern "C" int tnerf_train_step_dataset(const tnerf_step_args* a, tnerf_stream_t stream_) {
    int rc = check_step_args(a); if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    const int64_t n_params = tnerf_param_count(&a->desc);
    if (n_params < 0) return TNERF_EUNSUPPORTED;
    const float inv_denom = (float)(1.0 / a->loss_denominator);
    if (a->phases & TNERF_PHASE_GRADIENT) {
        RaySource rs{};
        rs.c2w = a->poses; rs.H = a->H; rs.W = a->W; rs.focal = a->focal;
        rs.step = a->step; rs.n_images = a->n_images; rs.seed = a->seed; rs.ray_first = a->ray_first; rs.rays_global = a->n_rays_global;
        const TnStepRef sr{a->step, (uint64_t)a->n_rays_global * (uint64_t)a->n_samples};
        const LossArgs loss{a->pixels, nullptr, inv_denom, a->ray_ws, a->pix_out};
        const uint64_t off0 = (uint64_t)a->ray_first * (uint64_t)a->n_samples;         // + step * per_step in the kernels
        if (a->precision == 0)
            rc = tn_step32_core("tnerf_train_step_dataset", &a->desc, static_cast<const float*>(a->packed), a->packed_x3, rs, sr, loss, a->n_rays, a->n_samples,
                                a->ztab, 1, nullptr, a->seed, off0, a->white_bkgd, a->comp_rgb, static_cast<float*>(a->stash), a->stash_row_stride,
                                a->job_table, a->n_jobs, a->slabs, stream);
        else
            rc = tn_step16_core("tnerf_train_step_dataset", &a->desc, a->packed, rs, sr, loss, a->n_rays, a->n_samples, a->ztab, 1, nullptr, a->seed,
                                off0, a->white_bkgd, a->comp_rgb, a->stash, a->job_table, a->n_jobs, a->slabs, stream);
        if (rc) return rc;
    }
    if (!(a->phases & (TNERF_PHASE_REDUCE | TNERF_PHASE_UPDATE))) return TNERF_OK;
    FinishArgs f{};
    f.n_params = n_params; f.grads = a->grads;
    if (a->phases & TNERF_PHASE_REDUCE) {
        f.slabs = a->slabs; f.reduce_table = a->reduce_table;
        if (a->loss_out && a->ray_ws) { f.ray_ws = a->ray_ws; f.R = a->n_rays; f.inv_denom = inv_denom; f.loss_out = a->loss_out; }
    }
    if (a->phases & TNERF_PHASE_UPDATE) {
        f.params = a->params; f.m = a->exp_avg; f.v = a->exp_avg_sq; f.lr = a->lr; f.b1 = a->beta1; f.b2 = a->beta2; f.eps = a->eps;
        f.gscale = 1.0f; f.step = a->step;
        f.scatter = a->scatter_table; f.width = a->scatter_width; f.packed = const_cast<void*>(a->packed);
        if (a->precision == 1 && a->scatter_table) {
            Net16 n; if ((rc = tn_build_net16(&a->desc, &n))) return rc;
            f.bf16_elems = (int64_t)(n.n_frag + n.n_bw_frag) * 512; f.bias_off_bytes = n.bias_off;
        }
        if (a->precision == 0 && a->packed_x3 && a->scatter_x3) {
            NetX3 n; if ((rc = tn_build_netx3(&a->desc, &n))) return rc;
            f.scatter3 = a->scatter_x3; f.width3 = a->scatter_x3_width; f.packed3 = const_cast<void*>(a->packed_x3);
            f.x3_elems = (int64_t)(n.n_rec + n.n_bw_rec) * n.rec_frags * 512; f.n3 = n; f.scale_floor = 16.0f * a->lr;
        }
    }
    // The finishing kernel re-scatters every weight into the x3 stream; then the scale it used is published and the layers' maxima are
    // refreshed, with headroom for the next update (tx_stats_final) — by the kernel's last workgroup for small networks, in a launch of
    // its own for large ones (k_finish).
    if (f.scatter3) f.fold_stats = (n_params + 255) / 256 <= TN_FOLD_STATS_BLOCKS;
    if ((rc = tn_launch_finish(f, stream))) return rc;
    if (f.scatter3 && !f.fold_stats) return tnx3_launch_stats(f.n3, a->params, a->pack_x3, f.packed3, 1, stream, f.scale_floor);
    return TNERF_OK;
}

// ------------------------------------------------------------------------------------------- hipGraph
// Capture whatever the caller launches on `stream` between begin and end (it must not be the NULL stream, and everything
// launched must keep its buffers alive for the life of the executable graph).
extern "C" int tnerf_graph_begin(tnerf_stream_t stream) {
    if (!stream) { tn_set_error("tnerf_graph_begin: the NULL stream cannot be captured; use a stream of your own"); return TNERF_EINVAL; }
    const hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { tn_set_error("hipStreamBeginCapture: %s", hipGetErrorString(e)); return (int)e; }
    return TNERF_OK;
}

extern "C" int tnerf_graph_end(tnerf_stream_t stream, void** graph_exec_out) {
    if (!stream || !graph_exec_out) { tn_set_error("tnerf_graph_end: NULL argument"); return TNERF_EINVAL; }
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
    if (e != hipSuccess || !g) { tn_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return e != hipSuccess ? (int)e : TNERF_EINVAL; }
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { tn_set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return (int)e; }
    *graph_exec_out = (void*)x;
    return TNERF_OK;
}

extern "C" int tnerf_graph_launch(void* graph_exec, tnerf_stream_t stream) {
    if (!graph_exec) { tn_set_error("tnerf_graph_launch: NULL graph"); return TNERF_EINVAL; }
    const hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    if (e != hipSuccess) { tn_set_error("hipGraphLaunch: %s", hipGetErrorString(e)); return (int)e; }
    return TNERF_OK;
}

extern "C" int tnerf_graph_destroy(void* graph_exec) {
    if (!graph_exec) return TNERF_OK;
    const hipError_t e = hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    if (e != hipSuccess) { tn_set_error("hipGraphExecDestroy: %s", hipGetErrorString(e)); return (int)e; }
    return TNERF_OK;
}
