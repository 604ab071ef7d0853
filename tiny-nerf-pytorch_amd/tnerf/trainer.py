"""Fused train step (reference src/train.py:114-128) on the flat parameter buffer:
   tnerf_train_step_fused (forward, MSE, backward, wgrad, slab reduce) -> [all-reduce] -> tnerf_adam_step.

`FlatAdam` is a torch.optim.Optimizer whose state_dict has torch.optim.Adam's layout (per-parameter
'step', 'exp_avg', 'exp_avg_sq'), so checkpoints written by the reference loop load here and vice versa
(reference src/train.py:83-92,142-148)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

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
        st.packed_key = None                     # the packed copies (fp32 fragments, bf16 stream) are stale now
        if st.bf16 is not None:
            st.bf16.key = None
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
            self._gws = torch.empty(R, 3, dtype=torch.float32, device=dev)
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
                    self._comp.data_ptr(), self._gws.data_ptr(), self.loss.data_ptr(), bp.stash.data_ptr(),
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
                    self._comp.data_ptr(), self._gws.data_ptr(), self.loss.data_ptr(), stash.data_ptr(), plan.Mp,
                    plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr(), plan.reduce.data_ptr(), st.grad.data_ptr(),
                    torch.cuda.current_stream(dev).cuda_stream)
        _dist.all_reduce_sum_(st.grad)
        self.opt.step(grads_in_flat=True)
        return self.loss, self._comp

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
            self._gws = torch.empty(R, 3, dtype=torch.float32, device=dev)
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
                    ztab.data_ptr(), rnd, _ops._ptr(tr), seed, off, self.white, denom, self._comp.data_ptr(), self._gws.data_ptr(),
                    self.loss.data_ptr(), bp.stash.data_ptr(), bp.jobs.data_ptr(), bp.n_jobs, bp.slabs.data_ptr(),
                    bp.reduce.data_ptr(), st.grad.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
            _dist.all_reduce_sum_(st.grad)
            self.opt.step(grads_in_flat=True)
            return self.loss, self._comp
        self.model._ensure_packed()
        plan = st.plan(R * self.S)
        with plan.lease() as stash:
            _l.call("tnerf_train_step_fused_cam", C.byref(st.desc), st.packed.data_ptr(), C.byref(cam), pixels.data_ptr(), R, self.S,
                    ztab.data_ptr(), rnd, _ops._ptr(tr), seed, off, self.white, denom, self._comp.data_ptr(), self._gws.data_ptr(),
                    self.loss.data_ptr(), stash.data_ptr(), plan.Mp, plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr(),
                    plan.reduce.data_ptr(), st.grad.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        _dist.all_reduce_sum_(st.grad)
        self.opt.step(grads_in_flat=True)
        return self.loss, self._comp
