"""Fused train step (reference src/train.py:114-128) on the flat parameter buffer:
   tnerf_train_step_fused (forward, MSE, backward, wgrad, slab reduce) -> [all-reduce] -> tnerf_adam_step.

`FlatAdam` is a torch.optim.Optimizer whose state_dict has torch.optim.Adam's layout (per-parameter
'step', 'exp_avg', 'exp_avg_sq'), so checkpoints written by the reference loop load here and vice versa
(reference src/train.py:83-92,142-148)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import lib as _l
from . import ops as _ops
from . import dist as _dist


class FlatAdam(torch.optim.Optimizer):
    """Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0) as ONE kernel over the model's flat buffer."""

    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self._model = model
        self._st: _ops.ModelState = model.hip_state()
        params = list(model.parameters())
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None))
        st = self._st
        self._m = torch.zeros_like(st.flat)
        self._v = torch.zeros_like(st.flat)
        self._t = 0
        self._params = params
        self._sync_state()

    def _sync_state(self):
        st = self._st
        for p, o in zip(self._params, st.offsets):
            n = p.numel()
            self.state[p] = {"step": torch.tensor(float(self._t)),
                             "exp_avg": self._m[o:o + n].view(p.shape),
                             "exp_avg_sq": self._v[o:o + n].view(p.shape)}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        st = self._st
        t = 0
        for p, o in zip(self._params, st.offsets):
            s = self.state.get(p, {})
            if "exp_avg" in s:
                n = p.numel()
                self._m[o:o + n].view(p.shape).copy_(s["exp_avg"])
                self._v[o:o + n].view(p.shape).copy_(s["exp_avg_sq"])
                t = int(float(s["step"]))
        self._t = t
        self._sync_state()

    @torch.no_grad()
    def step(self, closure=None, grads_in_flat: bool = False, grad_scale: float = 1.0):
        """grads_in_flat=True: the gradient already sits in the model's flat grad buffer (fused step);
        otherwise p.grad of every parameter is gathered into it first."""
        st = self._model.hip_state()             # re-adopts the parameters if one of them was rebound since the last step
        if st is not self._st:
            raise RuntimeError("FlatAdam: the model moved to another device after the optimizer was built")
        if not grads_in_flat:
            for p, gv in zip(self._params, st.grad_views(self._params)):
                if p.grad is None:
                    gv.zero_()
                else:
                    gv.copy_(p.grad)
        g = self.param_groups[0]
        self._t += 1
        _l.call("tnerf_adam_step", st.flat.data_ptr(), st.grad.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), st.n_params,
                float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), self._t, float(grad_scale),
                torch.cuda.current_stream(st.device).cuda_stream)
        st.generation += 1                       # (a raw kernel wrote the parameters: torch's _version counters did not move)
        st.packed_key = None                     # the packed copies (fp32 fragments, bf16 stream, x3 records) are stale now
        if st.bf16 is not None:
            st.bf16.key = None
        if st.x3 is not None:
            st.x3.key = None
        return None          # the per-parameter "step" scalars are refreshed lazily (state_dict), not 2*depth+4 times per step

    def _refresh_steps(self):
        for p in self._params:
            self.state[p]["step"].fill_(float(self._t))

    def state_dict(self):
        self._refresh_steps()
        return super().state_dict()


class FusedTrainer:
    """One object per (model, optimizer): `step()` is the body of the reference training loop."""

    def __init__(self, model, optimizer: FlatAdam, near: float, far: float, n_samples: int, white_bkgd: bool = True,
                 precision: str = "fp32"):
        """precision "bf16" (BASELINE cfg 4): bf16 weights / activations / activation gradients on MFMA with fp32
        accumulation, fp32 compositing, fp32 weight gradients, fp32 master weights and Adam state."""
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"precision must be 'fp32' or 'bf16', got {precision!r}")
        self.precision = precision
        self.model, self.opt = model, optimizer
        self.near, self.far, self.S, self.white = float(near), float(far), int(n_samples), int(bool(white_bkgd))
        self.st: _ops.ModelState = model.hip_state()
        dev = self.st.device
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self._comp = None
        self._gws = None

    @torch.no_grad()
    def step(self, rays_o, rays_d, target, t_rand: Optional[torch.Tensor] = None, philox=None,
             global_rays: Optional[int] = None, randomized: bool = True):
        """Returns (loss contribution of this rank's rays [device scalar], comp_rgb).  With torch.distributed
        initialised the flat gradient is all-reduced (SUM) before Adam; pass global_rays = total rays over all
        ranks so that the shards' losses/gradients add up to the full-batch ones."""
        st, dev = self.st, self.st.device
        rays_o, rays_d, target = _ops._f32c(rays_o), _ops._f32c(rays_d), _ops._f32c(target)
        R = rays_o.shape[0]
        if self._comp is None or self._comp.shape[0] != R:
            self._comp = torch.empty(R, 3, dtype=torch.float32, device=dev)
            self._gws = torch.empty(_l.train_ws_floats(R), dtype=torch.float32, device=dev)      # per ray: dL/dcomp_rgb + squared error
        ztab = _ops.depth_table(self.near, self.far, self.S, dev)
        rnd, tr, seed, off = _ops._rng_args(randomized, t_rand, philox)
        if tr is not None:
            tr = _ops._f32c(tr)
        denom = 3.0 * float(global_rays if global_rays is not None else R)
        if self.precision == "bf16":
            b = st.repack_bf16(tuple(p._version for p in self.model._param_list()))
            bp = b.train_plan(R, self.S)
            _l.call("tnerf_train_step_fused_bf16", C.byref(st.desc), b.packed.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(),
                    target.data_ptr(), R, self.S, ztab.data_ptr(), rnd, _ops._ptr(tr), seed, off, self.white, denom,
                    self._comp.data_ptr(), self._gws.data_ptr(), self._gws.numel(), self.loss.data_ptr(), bp.stash.data_ptr(),
                    bp.jobs.data_ptr(), bp.n_jobs, bp.slabs.data_ptr(), bp.reduce.data_ptr(), st.grad.data_ptr(),
                    torch.cuda.current_stream(dev).cuda_stream)
            _dist.all_reduce_sum_(st.grad)
            self.opt.step(grads_in_flat=True)
            return self.loss, self._comp
        self.model._ensure_packed()
        plan = st.plan(R * self.S)
        with plan.lease() as stash:               # not the buffer a pending autograd node of the same size may still hold
            _l.call("tnerf_train_step_fused", C.byref(st.desc), st.packed.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(),
                    target.data_ptr(), R, self.S, ztab.data_ptr(), rnd, _ops._ptr(tr), seed, off, self.white, denom,
                    self._comp.data_ptr(), self._gws.data_ptr(), self._gws.numel(), self.loss.data_ptr(), stash.data_ptr(), plan.Mp,
                    plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr(), plan.reduce.data_ptr(), st.grad.data_ptr(),
                    self._x3_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        _dist.all_reduce_sum_(st.grad)
        self.opt.step(grads_in_flat=True)
        return self.loss, self._comp

    def _x3_ptr(self):
        """The x3 record stream of the current weights (the chains then run on the fp16 matrix pipe, three partial products), or
        None for models the x3 kernels do not cover / when the fp32-MFMA kernels were asked for."""
        st = self.st
        if (st.desc.flags & _l.FLAG_FP32_MFMA) or not st.x3_capable:
            return None
        b = st.repack_x3(tuple(p._version for p in self.model._param_list()))
        return b.packed.data_ptr() if st.uses_x3 else None          # (an "auto" model may have left the x3 pipe's domain in that pack)

    @torch.no_grad()
    def step_camera(self, pose, H: int, W: int, focal: float, inds, pixels, t_rand: Optional[torch.Tensor] = None, philox=None,
                    global_rays: Optional[int] = None, randomized: bool = True):
        """The same step with the rays generated in the kernel from `pose` and the flat pixel indices `inds`
        (train.py:109) and the targets read as pixels[inds] (`pixels`: this image as [H*W, 3]): no (N,HW,3) ray
        tables, no gathers."""
        st, dev = self.st, self.st.device
        R = int(inds.shape[0])
        if self._comp is None or self._comp.shape[0] != R:
            self._comp = torch.empty(R, 3, dtype=torch.float32, device=dev)
            self._gws = torch.empty(_l.train_ws_floats(R), dtype=torch.float32, device=dev)      # per ray: dL/dcomp_rgb + squared error
        ztab = _ops.depth_table(self.near, self.far, self.S, dev)
        rnd, tr, seed, off = _ops._rng_args(randomized, t_rand, philox)
        if tr is not None:
            tr = _ops._f32c(tr)
        pixels = _ops._f32c(pixels)
        cam, keep = _ops.camera_struct(pose, H, W, focal, inds, 0)
        denom = 3.0 * float(global_rays if global_rays is not None else R)
        if self.precision == "bf16":
            b = st.repack_bf16(tuple(p._version for p in self.model._param_list()))
            bp = b.train_plan(R, self.S)
            _l.call("tnerf_train_step_fused_cam_bf16", C.byref(st.desc), b.packed.data_ptr(), C.byref(cam), pixels.data_ptr(), R, self.S,
                    ztab.data_ptr(), rnd, _ops._ptr(tr), seed, off, self.white, denom, self._comp.data_ptr(), self._gws.data_ptr(), self._gws.numel(),
                    self.loss.data_ptr(), bp.stash.data_ptr(), bp.jobs.data_ptr(), bp.n_jobs, bp.slabs.data_ptr(),
                    bp.reduce.data_ptr(), st.grad.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
            _dist.all_reduce_sum_(st.grad)
            self.opt.step(grads_in_flat=True)
            return self.loss, self._comp
        self.model._ensure_packed()
        plan = st.plan(R * self.S)
        with plan.lease() as stash:
            _l.call("tnerf_train_step_fused_cam", C.byref(st.desc), st.packed.data_ptr(), C.byref(cam), pixels.data_ptr(), R, self.S,
                    ztab.data_ptr(), rnd, _ops._ptr(tr), seed, off, self.white, denom, self._comp.data_ptr(), self._gws.data_ptr(), self._gws.numel(),
                    self.loss.data_ptr(), stash.data_ptr(), plan.Mp, plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr(),
                    plan.reduce.data_ptr(), st.grad.data_ptr(), self._x3_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        _dist.all_reduce_sum_(st.grad)
        self.opt.step(grads_in_flat=True)
        return self.loss, self._comp


def _scatter_table(pack_table: np.ndarray, n_params: int) -> np.ndarray:
    """Inverse of a pack table (packed[i] = params[table[i]]): row p lists the packed positions of parameter p, -1 padded.
    A weight sits in the forward AND in the transposed (dgrad) fragment stream, a bias once."""
    pos = np.nonzero(pack_table >= 0)[0].astype(np.int64)
    src = pack_table[pos].astype(np.int64)
    order = np.argsort(src, kind="stable")
    src, pos = src[order], pos[order]
    counts = np.bincount(src, minlength=n_params)
    width = max(1, int(counts.max()))
    first = np.concatenate([[0], np.cumsum(counts)[:-1]])
    out = np.full((n_params, width), -1, np.int32)
    out[src, np.arange(src.shape[0]) - first[src]] = pos.astype(np.int32)
    return out


class DatasetTrainer:
    """The reference loop body (src/train.py:106-128) on device-resident state: `step()` is ONE host call.

    The dataset (poses, pixels) lives in HBM; which image (step % N), which pixels (the reference's torch.randint) and
    which jitter (its rand_like) are drawn inside the kernels from a device-side step counter (Philox, include/tnerf.h
    "the whole step on device-resident state"); Adam and the re-packing of the updated weights finish the same call.
    On one GPU the launches of a step are captured once into a hipGraph and replayed (graph=True).  With
    torch.distributed initialised the rows of the global batch are sharded (dist.shard_bounds) and the flat gradient is
    all-reduced between the gradient and the update phase: two captured graphs around one eager collective.

    This is the speed path; FusedTrainer.step / step_camera with torch-drawn `inds` / `t_rand` is the parity path."""

    def __init__(self, model, optimizer: FlatAdam, images: torch.Tensor, poses: torch.Tensor, focal: float, n_rand: int,
                 n_samples: int, near: float, far: float, seed: int = 0, white_bkgd: bool = True, precision: str = "fp32",
                 graph: bool = True, start_step: int = 0, record_pixels: bool = False, rank: Optional[int] = None,
                 world: Optional[int] = None):
        """rank / world default to the torch.distributed process group (1 rank without one); pass them to shard by hand
        (then call gradient_phase(), exchange `model.hip_state().grad` yourself, update_phase())."""
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"precision must be 'fp32' or 'bf16', got {precision!r}")
        self.model, self.opt, self.precision = model, optimizer, precision
        st = self.st = model.hip_state()
        dev = st.device
        _ops._need_cuda(images, poses)
        N, H, W, _ = images.shape
        self.N, self.H, self.W, self.focal = int(N), int(H), int(W), float(focal)
        self.pixels = _ops._f32c(images).reshape(N, H * W, 3)
        self.poses = _ops._f32c(poses).reshape(N, 16)
        self.S, self.white, self.seed = int(n_samples), int(bool(white_bkgd)), int(seed) & (2 ** 64 - 1)
        self.ztab = _ops.depth_table(float(near), float(far), self.S, dev)
        if rank is None or world is None:
            rank, world = _dist.world()
        self.world = world
        self.R_global = int(n_rand)
        lo, hi = _dist.shard_bounds(self.R_global, rank, world)
        self.lo, self.R = lo, hi - lo
        if self.R < 1:
            raise ValueError(f"rank {rank} of {world} has no rays of a {n_rand}-ray batch")
        if int(optimizer._t) != int(start_step):
            raise ValueError(f"DatasetTrainer: the loop step ({start_step}) and Adam's step count ({optimizer._t}) must agree "
                             "(one device counter serves both); use FusedTrainer.step_camera otherwise")
        self.step_dev = torch.tensor([int(start_step)], dtype=torch.int64, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.comp = torch.empty(self.R, 3, dtype=torch.float32, device=dev)
        self.ray_ws = torch.empty(_l.train_ws_floats(self.R), dtype=torch.float32, device=dev)
        self.pix = torch.empty(self.R, dtype=torch.int32, device=dev) if record_pixels else None
        self._graph = None
        self._graph_key = None
        self._x3_packed = self._x3_scatter = None
        self._want_graph = bool(graph)
        self._graph_update = None
        self._start_step = int(start_step)
        self._calls = 0
        self._own_stream = None
        self._args_keep = None
        # packed weights + the inverse of their pack table
        if precision == "fp32":
            model._ensure_packed()
            self._packed = st.packed
            self._plan = st.plan(self.R * self.S)
            self._lease = self._plan.lease()                                  # this trainer's stash for as long as it lives
            self._stash, self._stride = self._lease.buf, self._plan.Mp
            tab = st.pack_table.cpu().numpy()
            if st.x3_capable and not (st.desc.flags & _l.FLAG_FP32_MFMA):    # forward on the x3 chain kernel; the finishing kernel keeps its stream current
                # (key None: always packed here, with headroom for the finishing kernel's first re-scatter — tnerf_mlp_pack_x3_floor)
                x3 = st.repack_x3(None, scale_floor=16.0 * float(optimizer.param_groups[0]["lr"]))
                if st.uses_x3:                                                # (an "auto" model whose weights are outside the x3 domain has just switched)
                    self._x3_packed, self._x3_table = x3.packed, x3.table
                    self._x3_scatter = torch.from_numpy(_scatter_table(x3.table.cpu().numpy(), st.n_params)).to(dev)
        else:
            b = st.repack_bf16(tuple(p._version for p in model._param_list()))
            self._packed = b.packed
            self._plan = b.train_plan(self.R, self.S)
            self._stash, self._stride = self._plan.stash, 0
            tab = b.table.cpu().numpy()
        self._scatter = torch.from_numpy(_scatter_table(tab, st.n_params)).to(dev)
        self._param_key = tuple(p._version for p in model._param_list())

    # ------------------------------------------------------------------ one step
    def _args(self, phases: int) -> "_l.StepArgs":
        st, g, plan = self.st, self.opt.param_groups[0], self._plan
        a = _l.StepArgs()
        a.desc = st.desc
        a.precision = 0 if self.precision == "fp32" else 1
        a.phases = phases
        a.poses, a.pixels = self.poses.data_ptr(), self.pixels.data_ptr()
        a.n_images, a.H, a.W, a.focal = self.N, self.H, self.W, self.focal
        a.n_rays, a.ray_first, a.n_rays_global = self.R, self.lo, self.R_global
        a.n_samples, a.white_bkgd = self.S, self.white
        a.ztab, a.seed, a.loss_denominator = self.ztab.data_ptr(), self.seed, 3.0 * self.R_global
        a.step, a.packed = self.step_dev.data_ptr(), self._packed.data_ptr()
        a.comp_rgb, a.ray_ws, a.loss_out = self.comp.data_ptr(), self.ray_ws.data_ptr(), self.loss.data_ptr()
        a.ray_ws_floats = self.ray_ws.numel()
        a.pix_out = self.pix.data_ptr() if self.pix is not None else None
        a.stash, a.stash_row_stride = self._stash.data_ptr(), self._stride
        a.stash_capacity = self._stash.numel()                # fp32: floats, bf16: bytes (a uint8 buffer)
        a.job_table, a.n_jobs, a.slabs = plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr()
        a.reduce_table, a.grads = plan.reduce.data_ptr(), st.grad.data_ptr()
        a.params, a.exp_avg, a.exp_avg_sq = st.flat.data_ptr(), self.opt._m.data_ptr(), self.opt._v.data_ptr()
        a.lr, a.beta1, a.beta2, a.eps = float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])
        a.scatter_table, a.scatter_width = self._scatter.data_ptr(), int(self._scatter.shape[1])
        if self._x3_packed is not None:
            a.packed_x3, a.scatter_x3, a.scatter_x3_width = self._x3_packed.data_ptr(), self._x3_scatter.data_ptr(), int(self._x3_scatter.shape[1])
            a.pack_x3 = self._x3_table.data_ptr()
        return a

    def _hyper_key(self):
        g = self.opt.param_groups[0]
        return (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))

    def _stream(self):
        cur = torch.cuda.current_stream(self.st.device)
        if cur.cuda_stream != 0:
            return cur, False                      # the caller already works on a stream of its own: use it
        if self._own_stream is None:
            self._own_stream = torch.cuda.Stream(self.st.device)
        return self._own_stream, True

    def _sync_params(self):
        """The kernels keep the packed copies current themselves; a parameter the CALLER wrote between two steps (p.copy_(), an
        in-place edit: its _version moved) is packed again here, into the same buffers (captured graphs stay valid)."""
        key = tuple(p._version for p in self.model._param_list())
        if key == self._param_key:
            return
        st = self.st
        if self.precision == "fp32":
            st.packed_key = None
            self.model._ensure_packed()
            if self._x3_packed is not None:
                st.repack_x3(None, scale_floor=16.0 * float(self.opt.param_groups[0]["lr"]))
        else:
            st.repack_bf16(None)
        self._param_key = key

    def _capture(self, stream, phases: int):
        """One tnerf_train_step_dataset call with `phases`, captured on `stream` -> (graph exec handle, the argument block it keeps)."""
        a = self._args(phases)
        _l.call("tnerf_graph_begin", stream.cuda_stream)
        try:
            _l.call("tnerf_train_step_dataset", C.byref(a), stream.cuda_stream)
        finally:
            gx = C.c_void_p()
            rc = _l.load().tnerf_graph_end(stream.cuda_stream, C.byref(gx))
        _l.check(rc, "tnerf_graph_end")
        return gx, a

    def _after_update(self):
        self._calls += 1
        self.opt._t += 1
        st = self.st
        st.generation += 1                       # the finishing kernel wrote the parameters (ops._check_versions)
        # the finishing kernel re-packed THIS precision's copies of the weights; the others are stale now
        if self.precision == "fp32":
            if st.bf16 is not None:
                st.bf16.key = None
            if st.x3 is not None and self._x3_packed is None:
                st.x3.key = None
        else:
            st.packed_key = None
            if st.x3 is not None:
                st.x3.key = None

    X3_CHECK_EVERY = 256
    skip_exchange_for_timing = False       # bench.py only: N ranks step WITHOUT the all-reduce (the ranks' weights diverge) to price the exchange

    def _leave_x3(self):
        """The model's flags now ask for the fp32-MFMA kernels: drop the x3 stream (the fp32 fragment copy has been kept current by every
        update) and the captured graphs (their kernels are the x3 ones)."""
        self._x3_packed = self._x3_scatter = None
        self._drop_graph()

    @torch.no_grad()
    def step(self):
        """One training step.  Returns (loss [device scalar: this rank's share of the batch MSE], comp_rgb [R,3]).
        One GPU: the whole step is one hipGraph replay.  N ranks: two replays — GRADIENT|REDUCE, then UPDATE — around one eager
        all-reduce(SUM) of the flat gradient on the same stream (a collective cannot sit inside the captured region: torch's
        process group owns its launch), so the host still issues three calls per step instead of six launches."""
        st = self.st
        if not st.owns(self.model._param_list()):
            raise RuntimeError("DatasetTrainer: a model parameter was rebound; build a new trainer")
        if int(self.opt._t) != self._calls + self._start_step:
            raise RuntimeError(f"DatasetTrainer: Adam's step count ({self.opt._t}) no longer equals the loop's ({self._calls + self._start_step}): "
                               "the optimizer was stepped outside this trainer; one device counter serves both")
        self._sync_params()
        # an "auto" model: every X3_CHECK_EVERY steps the x3 pipe's domain is checked against the weights as they are now (one small
        # kernel + a 72-byte read-back); outside it the step continues on the fp32-MFMA kernels (ops.ModelState.check_x3_domain)
        if self._x3_packed is not None and st.pipe_policy == "auto" and self._calls > 0 and self._calls % self.X3_CHECK_EVERY == 0:
            st.check_x3_domain()
        if self._x3_packed is not None and not st.uses_x3:      # (the check above, or a re-pack of caller-edited parameters, switched the model)
            self._leave_x3()
        full = _l.PHASE_GRADIENT | _l.PHASE_REDUCE | _l.PHASE_UPDATE
        if not self._want_graph:
            if self.world > 1:                      # gradient -> all-reduce -> update, six eager launches
                self.gradient_phase()
                _dist.all_reduce_sum_(st.grad)
                return self.update_phase()
            a = self._args(full)
            _l.call("tnerf_train_step_dataset", C.byref(a), torch.cuda.current_stream(st.device).cuda_stream)
            self._after_update()
            return self.loss, self.comp
        stream, own = self._stream()
        if own:
            stream.wait_stream(torch.cuda.current_stream(st.device))
        if self._graph is not None and self._graph_key != self._hyper_key():
            self._drop_graph()                 # lr / betas changed: the captured kernel arguments are stale
        if self._graph is None and self._calls >= 1:
            # capture (the first call ran eagerly: every kernel is loaded, every buffer exists)
            try:
                if self.world == 1:
                    self._graph, self._args_keep = self._capture(stream, full)
                else:
                    self._graph, keep_g = self._capture(stream, _l.PHASE_GRADIENT | _l.PHASE_REDUCE)
                    self._graph_update, keep_u = self._capture(stream, _l.PHASE_UPDATE)
                    self._args_keep = (keep_g, keep_u)
                self._graph_key = self._hyper_key()
            except (RuntimeError, ValueError) as e:      # a runtime that cannot capture here: the same launches, eagerly, from now on
                import warnings
                warnings.warn(f"DatasetTrainer: hipGraph capture failed ({e}); continuing with eager launches")
                self._drop_graph()
                self._want_graph = False
                if own:
                    torch.cuda.current_stream(st.device).wait_stream(stream)
                return self.step()
        with torch.cuda.stream(stream):
            if self.world == 1:
                if self._graph is not None:
                    _l.call("tnerf_graph_launch", self._graph, stream.cuda_stream)
                else:
                    a = self._args(full)
                    _l.call("tnerf_train_step_dataset", C.byref(a), stream.cuda_stream)
            else:
                if self._graph is not None:
                    _l.call("tnerf_graph_launch", self._graph, stream.cuda_stream)
                else:
                    a = self._args(_l.PHASE_GRADIENT | _l.PHASE_REDUCE)
                    _l.call("tnerf_train_step_dataset", C.byref(a), stream.cuda_stream)
                if not self.skip_exchange_for_timing:
                    _dist.all_reduce_sum_(st.grad)      # ordered on `stream` (torch's collective waits for / is waited on by the current stream)
                if self._graph is not None:
                    _l.call("tnerf_graph_launch", self._graph_update, stream.cuda_stream)
                else:
                    a = self._args(_l.PHASE_UPDATE)
                    _l.call("tnerf_train_step_dataset", C.byref(a), stream.cuda_stream)
        if own:
            torch.cuda.current_stream(st.device).wait_stream(stream)
        self._after_update()
        return self.loss, self.comp

    @torch.no_grad()
    def gradient_phase(self):
        """forward + loss + backward of this rank's rows -> model.hip_state().grad (its share of the batch gradient);
        advances the device step counter."""
        a = self._args(_l.PHASE_GRADIENT | _l.PHASE_REDUCE)
        _l.call("tnerf_train_step_dataset", C.byref(a), torch.cuda.current_stream(self.st.device).cuda_stream)

    @torch.no_grad()
    def update_phase(self):
        """Adam on whatever model.hip_state().grad holds now (the all-reduced gradient) + re-pack."""
        a = self._args(_l.PHASE_UPDATE)
        _l.call("tnerf_train_step_dataset", C.byref(a), torch.cuda.current_stream(self.st.device).cuda_stream)
        self._after_update()
        return self.loss, self.comp

    @property
    def steps_done(self) -> int:
        return int(self.step_dev.item())

    def _drop_graph(self):
        for name in ("_graph", "_graph_update"):
            g = getattr(self, name, None)
            if g is not None:
                _l.call("tnerf_graph_destroy", g)
                setattr(self, name, None)

    def __del__(self):
        try:
            self._drop_graph()
        except Exception:
            pass
