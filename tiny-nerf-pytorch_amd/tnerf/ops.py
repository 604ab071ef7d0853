"""Host side of the HIP hot path: torch tensors in, torch tensors out, every FLOP in libtnerf_hip.so.

PyTorch is used here only as plumbing (device memory, the current HIP stream, autograd bookkeeping,
the random draws the reference takes from torch's generator).  Nothing in this module computes the
path's arithmetic with torch ops, and nothing falls back to a CPU implementation: CPU tensors or a
missing library raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import lib as _l

_NULL = None
FLAG_FP32_MFMA = _l.FLAG_FP32_MFMA


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _need_cuda(*ts: torch.Tensor) -> torch.device:
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("tnerf HIP path: expected tensors on a ROCm GPU (device 'cuda'), got a CPU tensor; "
                               "there is no CPU fallback in this package")
        dev = dev or t.device
        if t.device != dev:
            raise RuntimeError(f"tnerf HIP path: tensors on different devices ({dev} vs {t.device})")
    return dev


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# --------------------------------------------------------------------------------- depth tables
_ZTAB: Dict[Tuple, torch.Tensor] = {}


def depth_table(near: float, far: float, n_samples: int, dev: torch.device) -> torch.Tensor:
    """Device copy of tnerf_sample_tables(near, far, S): [z | lo | hi], 3*S floats (bit-exact bins)."""
    key = (float(near), float(far), int(n_samples), str(dev))
    t = _ZTAB.get(key)
    if t is None:
        host = np.empty(3 * n_samples, np.float32)
        _l.call("tnerf_sample_tables", float(near), float(far), int(n_samples), host.ctypes.data_as(C.c_void_p), None)
        t = torch.from_numpy(host).to(dev)
        if len(_ZTAB) > 64:
            _ZTAB.clear()
        _ZTAB[key] = t
    return t


# ----------------------------------------------------------------------------------- stage ops
class _RayDirsFn(torch.autograd.Function):
    """rays_d of get_rays as a function of the pose (a caller that learns camera poses): tnerf_get_rays / tnerf_get_rays_bwd."""

    @staticmethod
    def forward(ctx, c2w, H, W, focal):
        dev = c2w.device
        rays_d = torch.empty(H * W, 3, dtype=torch.float32, device=dev)
        _l.call("tnerf_get_rays", int(H), int(W), float(focal), c2w.data_ptr(), None, rays_d.data_ptr(), _stream(dev))
        ctx.save_for_backward(c2w)
        ctx.cfg = (int(H), int(W), float(focal))
        return rays_d

    @staticmethod
    def backward(ctx, g_d):
        (c2w,) = ctx.saved_tensors
        H, W, focal = ctx.cfg
        dev = c2w.device
        g_d = _f32c(g_d)
        n = int(_l.load().tnerf_get_rays_bwd_scratch_floats(H, W))
        scratch = torch.empty(n, dtype=torch.float32, device=dev)
        d_c2w = torch.empty(4, 4, dtype=torch.float32, device=dev)
        _l.call("tnerf_get_rays_bwd", H, W, focal, c2w.data_ptr(), g_d.data_ptr(), scratch.data_ptr(), n, d_c2w.data_ptr(), _stream(dev))
        return d_c2w, None, None, None


def get_rays(H: int, W: int, focal: float, c2w: torch.Tensor, want_origin_copy: bool = False):
    dev = _need_cuda(c2w)
    if c2w.shape != (4, 4):
        raise ValueError(f"c2w must be (4,4), got {tuple(c2w.shape)}")
    if c2w.requires_grad and torch.is_grad_enabled():      # learned pose: directions through the kernel pair, origins through torch's expand
        cc = c2w if (c2w.dtype == torch.float32 and c2w.is_contiguous()) else c2w.float().contiguous()
        return cc[:3, 3].expand(H * W, 3), _RayDirsFn.apply(cc, int(H), int(W), float(focal))
    c2w = _f32c(c2w)
    rays_d = torch.empty(H * W, 3, dtype=torch.float32, device=dev)
    rays_o = torch.empty(H * W, 3, dtype=torch.float32, device=dev) if want_origin_copy else None
    _l.call("tnerf_get_rays", int(H), int(W), float(focal), c2w.data_ptr(), _ptr(rays_o), rays_d.data_ptr(), _stream(dev))
    if rays_o is None:
        rays_o = c2w[:3, 3].expand(H * W, 3)          # stride-0 view, as the reference returns
    return rays_o, rays_d


def _rng_args(randomized: bool, t_rand: Optional[torch.Tensor], philox: Optional[Tuple[int, int]]):
    if not randomized:
        return 0, None, 0, 0
    if t_rand is not None:
        return 1, t_rand, 0, 0
    if philox is None:
        raise ValueError("randomized sampling needs t_rand or a (seed, offset) pair")
    return 1, None, int(philox[0]), int(philox[1])


def sample_along_rays(near: float, far: float, n_samples: int, rays_o: torch.Tensor, rays_d: torch.Tensor,
                      randomized: bool, t_rand: Optional[torch.Tensor] = None, philox=None,
                      want_pts: bool = True, encode: Optional[Tuple[int, bool]] = None):
    dev = _need_cuda(rays_o, rays_d, t_rand)
    rays_o, rays_d = _f32c(rays_o), _f32c(rays_d)
    R, S = rays_o.shape[0], int(n_samples)
    ztab = depth_table(near, far, S, dev)
    rnd, tr, seed, off = _rng_args(randomized, t_rand, philox)
    if tr is not None:
        tr = _f32c(tr)
        if tr.shape != (R, S):
            raise ValueError(f"t_rand must be ({R},{S})")
    z = torch.empty(R, S, dtype=torch.float32, device=dev) if (randomized or want_pts) else None
    pts = torch.empty(R, S, 3, dtype=torch.float32, device=dev) if want_pts else None
    enc = None
    L, inc = (0, 1)
    if encode is not None:
        L, inc = int(encode[0]), int(bool(encode[1]))
        enc = torch.empty(R * S, 6 * L + 3 * inc, dtype=torch.float32, device=dev)
    _l.call("tnerf_sample_encode_fwd", rays_o.data_ptr(), rays_d.data_ptr(), R, S, ztab.data_ptr(), rnd, _ptr(tr), seed, off,
            _ptr(z), _ptr(pts), _ptr(enc), L, inc, _stream(dev))
    if not randomized:
        z = ztab[:S].expand(R, S)                      # stride-0 view, as the reference returns
    return z, pts, enc


_TTAB: Dict[Tuple, torch.Tensor] = {}


def sample_along_rays_per_ray(near, far, n_samples: int, rays_o: torch.Tensor, rays_d: torch.Tensor, randomized: bool,
                              t_rand: Optional[torch.Tensor] = None, philox=None):
    """stratified_samples with near / far given as tensors broadcastable to (R, 1) (reference src/sampling.py:8):
    every ray has its own bins.  Python floats mixed with tensors are rounded to fp32 as torch's scalar ops do."""
    dev = _need_cuda(rays_o, rays_d, t_rand)
    rays_o, rays_d = _f32c(rays_o), _f32c(rays_d)
    R, S = rays_o.shape[0], int(n_samples)

    def per_ray(v):
        t = torch.as_tensor(v, dtype=torch.float32).to(dev)
        return torch.broadcast_to(t, (R, 1)).reshape(R).contiguous()        # raises like the reference's broadcast would
    nr, fr = per_ray(near), per_ray(far)
    key = (S, str(dev))
    ttab = _TTAB.get(key)
    if ttab is None:
        z3, th = np.empty(3 * S, np.float32), np.empty(S, np.float32)
        _l.call("tnerf_sample_tables", 0.0, 1.0, S, z3.ctypes.data_as(C.c_void_p), th.ctypes.data_as(C.c_void_p))
        ttab = _TTAB[key] = torch.from_numpy(th).to(dev)
    rnd, tr, seed, off = _rng_args(randomized, t_rand, philox)
    if tr is not None:
        tr = _f32c(tr)
        if tr.shape != (R, S):
            raise ValueError(f"t_rand must be ({R},{S})")
    z = torch.empty(R, S, dtype=torch.float32, device=dev)
    pts = torch.empty(R, S, 3, dtype=torch.float32, device=dev)
    _l.call("tnerf_sample_per_ray_fwd", rays_o.data_ptr(), rays_d.data_ptr(), R, S, ttab.data_ptr(), nr.data_ptr(), fr.data_ptr(),
            rnd, _ptr(tr), seed, off, z.data_ptr(), pts.data_ptr(), _stream(dev))
    return z, pts


def posenc(x: torch.Tensor, num_freqs: int, include_input: bool) -> torch.Tensor:
    dev = _need_cuda(x)
    lead = x.shape[:-1]
    xf = x.reshape(-1, 3)
    xf = xf if (xf.dtype == torch.float32 and xf.is_contiguous()) else xf.float().contiguous()
    D = 6 * num_freqs + (3 if include_input else 0)
    return _PosEncFn.apply(xf, int(num_freqs), bool(include_input)).reshape(*lead, D)


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, sigma, z_vals, rays_d, white_bkgd):
        dev = _need_cuda(rgb, sigma, z_vals, rays_d)
        rgb, sigma, z_vals, rays_d = _f32c(rgb), _f32c(sigma), _f32c(z_vals), _f32c(rays_d)
        R, S = z_vals.shape
        comp = torch.empty(R, 3, dtype=torch.float32, device=dev)
        depth = torch.empty(R, 1, dtype=torch.float32, device=dev)
        acc = torch.empty(R, 1, dtype=torch.float32, device=dev)
        w = torch.empty(R, S, dtype=torch.float32, device=dev)
        _l.call("tnerf_composite_fwd", rgb.data_ptr(), sigma.data_ptr(), z_vals.data_ptr(), rays_d.data_ptr(), R, S,
                int(bool(white_bkgd)), comp.data_ptr(), depth.data_ptr(), acc.data_ptr(), w.data_ptr(), _stream(dev))
        ctx.save_for_backward(rgb, sigma, z_vals, rays_d)
        ctx.white = int(bool(white_bkgd))
        ctx.sigma_shape = sigma.shape
        return comp, depth, acc, w

    @staticmethod
    def backward(ctx, g_comp, g_depth, g_acc, g_w):
        rgb, sigma, z_vals, rays_d = ctx.saved_tensors
        dev = rgb.device
        R, S = z_vals.shape
        gs = [None if g is None else _f32c(g) for g in (g_comp, g_depth, g_acc, g_w)]
        d_rgb = torch.empty_like(rgb)
        d_sigma = torch.empty(ctx.sigma_shape, dtype=torch.float32, device=dev)
        need_z, need_d = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        if need_z or need_d:          # the geometry side (learned poses, resampling): dL/dz_vals, dL/drays_d
            d_z = torch.empty(R, S, dtype=torch.float32, device=dev) if need_z else None
            d_d = torch.empty(R, 3, dtype=torch.float32, device=dev) if need_d else None
            _l.call("tnerf_composite_bwd_geom", rgb.data_ptr(), sigma.data_ptr(), z_vals.data_ptr(), rays_d.data_ptr(), R, S, ctx.white,
                    _ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]), _ptr(gs[3]), d_rgb.data_ptr(), d_sigma.data_ptr(), _ptr(d_z), _ptr(d_d), _stream(dev))
            return d_rgb, d_sigma, d_z, d_d, None
        _l.call("tnerf_composite_bwd", rgb.data_ptr(), sigma.data_ptr(), z_vals.data_ptr(), rays_d.data_ptr(), R, S, ctx.white,
                _ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]), _ptr(gs[3]), d_rgb.data_ptr(), d_sigma.data_ptr(), _stream(dev))
        return d_rgb, d_sigma, None, None, None


def volume_render(rgb, sigma, z_vals, rays_d, white_bkgd=True):
    return _Composite.apply(rgb, sigma, z_vals, rays_d, white_bkgd)


class _PointsFn(torch.autograd.Function):
    """pts = o + d z (reference src/sampling.py:27) for already drawn depths: the differentiable part of stratified_samples."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, z_vals, pts):
        ctx.save_for_backward(rays_d, z_vals)
        return pts.view_as(pts)

    @staticmethod
    def backward(ctx, g_pts):
        rays_d, z_vals = ctx.saved_tensors
        dev = rays_d.device
        R, S = z_vals.shape
        g_pts = _f32c(g_pts)
        rd = _f32c(rays_d)
        zs = z_vals if z_vals.stride(-1) == 1 else z_vals.contiguous()
        stride = 0 if (zs.stride(0) == 0 or R == 1) else zs.stride(0)
        need_o, need_d, need_z = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        d_o = torch.empty(R, 3, dtype=torch.float32, device=dev) if need_o else None
        d_d = torch.empty(R, 3, dtype=torch.float32, device=dev) if need_d else None
        d_z = torch.empty(R, S, dtype=torch.float32, device=dev) if need_z else None
        _l.call("tnerf_sample_bwd", rd.data_ptr(), zs.data_ptr(), stride, g_pts.data_ptr(), R, S, _ptr(d_o), _ptr(d_d), _ptr(d_z), _stream(dev))
        return d_o, d_d, d_z, None


def attach_points_grad(rays_o, rays_d, z_vals, pts):
    """pts as computed by the sampling kernel, re-attached to autograd as o + d z when any of its inputs requires grad."""
    if torch.is_grad_enabled() and (rays_o.requires_grad or rays_d.requires_grad or z_vals.requires_grad):
        return _PointsFn.apply(rays_o, rays_d, z_vals, pts)
    return pts


class _DepthsFn(torch.autograd.Function):
    """z_vals as a function of tensor near / far bounds (reference src/sampling.py:16-25 is ordinary autograd, so bounds that
    require grad receive one).  The depths themselves are the sampling kernel's (bit-exact bins); they are LINEAR in the bounds,
    z_s = near a_s + far b_s with a = lerp of (1 - t), b = lerp of t through the same bin midpoints and jitter, so the backward is
    d near = sum_s g_s a_s, d far = sum_s g_s b_s — a handful of elementwise torch ops on (R, S) tensors, summed to the bounds' shapes."""

    @staticmethod
    def forward(ctx, near_t, far_t, z_vals, t_rand):
        ctx.shapes = (near_t.shape, far_t.shape)
        ctx.save_for_backward(t_rand if t_rand is not None else torch.empty(0, device=z_vals.device))
        ctx.randomized = t_rand is not None
        return z_vals.view_as(z_vals)

    @staticmethod
    def backward(ctx, g_z):
        (u,) = ctx.saved_tensors
        R, S = g_z.shape
        t = torch.linspace(0.0, 1.0, steps=S, device=g_z.device, dtype=torch.float32)

        def coeff(c):                                  # the reference's own arithmetic applied to the coefficient row c (S,)
            c = c.expand(R, S)
            if not ctx.randomized:
                return c
            mid = 0.5 * (c[:, :-1] + c[:, 1:])
            hi = torch.cat([mid, c[:, -1:]], dim=-1); lo = torch.cat([c[:, :1], mid], dim=-1)
            return lo + (hi - lo) * u
        g = g_z.float()
        d_near = (g * coeff(1.0 - t)).sum(-1, keepdim=True) if ctx.needs_input_grad[0] else None
        d_far = (g * coeff(t)).sum(-1, keepdim=True) if ctx.needs_input_grad[1] else None
        ns, fs = ctx.shapes
        return (d_near.sum_to_size(ns) if d_near is not None else None, d_far.sum_to_size(fs) if d_far is not None else None, None, None)


def attach_depth_grad(near, far, z_vals, t_rand):
    """z_vals as drawn by the sampling kernel, re-attached to autograd when near / far are tensors that require grad."""
    nt = near if isinstance(near, torch.Tensor) else None
    ft = far if isinstance(far, torch.Tensor) else None
    if torch.is_grad_enabled() and ((nt is not None and nt.requires_grad) or (ft is not None and ft.requires_grad)):
        dev = z_vals.device
        nt = nt if nt is not None else torch.tensor(float(near), device=dev)
        ft = ft if ft is not None else torch.tensor(float(far), device=dev)
        return _DepthsFn.apply(nt, ft, z_vals, t_rand)
    return z_vals


class _PosEncFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xf, num_freqs, include_input):
        dev = xf.device
        D = 6 * num_freqs + (3 if include_input else 0)
        out = torch.empty(xf.shape[0], D, dtype=torch.float32, device=dev)
        _l.call("tnerf_posenc_fwd", xf.data_ptr(), xf.shape[0], int(num_freqs), int(bool(include_input)), out.data_ptr(), _stream(dev))
        ctx.save_for_backward(xf)
        ctx.cfg = (int(num_freqs), int(bool(include_input)))
        return out

    @staticmethod
    def backward(ctx, g_out):
        (xf,) = ctx.saved_tensors
        g_out = _f32c(g_out)
        d_x = torch.empty_like(xf)
        _l.call("tnerf_posenc_bwd", xf.data_ptr(), xf.shape[0], ctx.cfg[0], ctx.cfg[1], g_out.data_ptr(), d_x.data_ptr(), _stream(xf.device))
        return d_x, None, None


# ------------------------------------------------------------------------------- model state
class _Plan:
    """Buffers and tables for one (model, sample count): stash, slabs, wgrad jobs, reduce table."""

    def __init__(self, st: "ModelState", M: int):
        dev = st.device
        sz = _l.PlanSizes()
        _l.call("tnerf_plan_sizes_query", C.byref(st.desc), int(M), st.n_cu, C.byref(sz))
        jobs = np.empty(sz.job_ints, np.int32)
        red = np.empty(sz.reduce_ints, np.int32)
        _l.call("tnerf_plan_fill", C.byref(st.desc), int(M), st.n_cu, None, jobs.ctypes.data_as(C.c_void_p),
                red.ctypes.data_as(C.c_void_p))
        self.M, self.Mp, self.n_jobs = int(M), int(sz.stash_row_stride), int(sz.n_jobs)
        self.jobs = torch.from_numpy(jobs).to(dev)
        self.reduce = torch.from_numpy(red).to(dev)
        self.stash = torch.empty(sz.stash_floats, dtype=torch.float32, device=dev)
        self.slabs = torch.empty(sz.slab_floats, dtype=torch.float32, device=dev)
        self._free = [self.stash]          # stash buffers nobody holds (see lease())

    def lease(self) -> "_StashLease":
        """Exclusive use of ONE stash buffer until the lease object dies.  A training forward writes the activations
        its backward needs into the stash; an autograd node keeps its lease alive in `ctx`, so a second forward of the
        same size before the first backward (two batches summed into one loss, a grad-enabled validation pass) gets a
        buffer of its own instead of overwriting the first one's activations."""
        return _StashLease(self)


class _StashLease:
    def __init__(self, plan: _Plan):
        self.plan = plan
        self.buf = plan._free.pop() if plan._free else torch.empty_like(plan.stash)

    def release(self):
        if self.buf is not None:
            if len(self.plan._free) < 2:                  # the primary buffer + one spare; larger pools go back to the allocator
                self.plan._free.append(self.buf)
            self.buf = None

    __del__ = release

    def __enter__(self):
        return self.buf

    def __exit__(self, *exc):
        self.release()


class ModelState:
    """Device-side state of one TinyNeRF: flat fp32 parameters (the nn.Parameters are views into it),
    the MFMA-fragment-packed copy of the weights and the per-batch-size plans."""

    def __init__(self, in_dim: int, hidden: int, depth: int, skip_at: int, device: torch.device, flags: Optional[int] = None):
        self.device = device
        # "auto" (matrix_pipe=None): the x3 pipe while the weights are inside its domain (x3_domain below), the fp32-MFMA kernels from the
        # moment they are not; an explicit matrix_pipe="x3" / "fp32_mfma" is never changed.
        self.pipe_policy = "auto" if flags is None else "fixed"
        self.pipe_switched: Optional[str] = None           # why an "auto" model left the x3 pipe (None: it has not)
        self.generation = 0                                # bumped by every update the framework itself makes to the flat parameters (raw kernels
                                                           # do not move torch's _version counters): FlatAdam.step, the trainers' steps
        self._x3_packs = 0
        if flags is None:      # TNERF_FP32_PIPE=mfma32 selects the plain fp32-MFMA kernels for models built from now on (A/B runs)
            env = os.environ.get("TNERF_FP32_PIPE", "").lower()
            flags = _l.FLAG_FP32_MFMA if env in ("mfma32", "fp32", "mfma") else 0
            if env in ("mfma32", "fp32", "mfma", "x3"):
                self.pipe_policy = "fixed"
        self.desc = _l.MlpDesc(int(in_dim), int(hidden), int(depth), int(skip_at), int(flags))
        n = _l.load().tnerf_param_count(C.byref(self.desc))
        if n < 0:
            _l.check(_l.EUNSUPPORTED, "TinyNeRF (HIP)")
        self.n_params = int(n)
        self.n_cu = torch.cuda.get_device_properties(device).multi_processor_count
        # the x3 chain kernels (fp32-grade products on the fp16 matrix pipe) cover the fused paths: in_dim = 6L+3
        self.x3_capable = in_dim >= 9 and (in_dim - 3) % 6 == 0
        k = 2 * depth + 4
        off = np.zeros(k, np.int64); rows = np.zeros(k, np.int64); cols = np.zeros(k, np.int64)
        _l.call("tnerf_param_layout", C.byref(self.desc), off.ctypes.data_as(C.c_void_p), rows.ctypes.data_as(C.c_void_p),
                cols.ctypes.data_as(C.c_void_p))
        self.offsets = [int(v) for v in off]
        sz = _l.PlanSizes()
        _l.call("tnerf_plan_sizes_query", C.byref(self.desc), 64, self.n_cu, C.byref(sz))
        self.packed_floats = int(sz.packed_floats)
        pack = np.empty(self.packed_floats, np.int32)
        _l.call("tnerf_plan_fill", C.byref(self.desc), 64, self.n_cu, pack.ctypes.data_as(C.c_void_p), None, None)
        self.pack_table = torch.from_numpy(pack).to(device)
        self.flat = torch.zeros(self.n_params, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.n_params, dtype=torch.float32, device=device)
        self.packed = torch.empty(self.packed_floats, dtype=torch.float32, device=device)
        self.packed_key = None
        self.plans: Dict[int, _Plan] = {}
        self.adopted_ptrs: Tuple[int, ...] = ()
        self.bf16: Optional[_Bf16State] = None          # built on first use of the bf16 mode
        self.x3: Optional["_X3State"] = None            # built on first use of the x3 chain kernels

    def plan(self, M: int) -> _Plan:
        p = self.plans.get(M)
        if p is None:
            if len(self.plans) >= 4:                      # stashes are large: keep only a few batch shapes alive
                self.plans.pop(next(iter(self.plans)))
            p = self.plans[M] = _Plan(self, M)
        return p

    def adopt(self, params) -> None:
        """Make the module's parameters views into the flat buffer (values preserved)."""
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                n = p.numel()
                view = self.flat[o:o + n].view(p.shape)
                if p.data_ptr() != view.data_ptr():
                    view.copy_(p.data)
                    p.data = view
        self.adopted_ptrs = tuple(p.data_ptr() for p in params)
        self.packed_key = None
        if self.bf16 is not None:
            self.bf16.key = None
        if self.x3 is not None:
            self.x3.key = None

    def owns(self, params) -> bool:
        """True while EVERY parameter is still the view into the flat buffer that adopt() made (2*depth+4 pointer
        compares): rebinding any one of them (`layer.weight = nn.Parameter(..)`, `p.data = ..`) is noticed."""
        return tuple(p.data_ptr() for p in params) == self.adopted_ptrs

    def repack(self, key=None) -> None:
        if key is not None and key == self.packed_key:
            return
        _l.call("tnerf_mlp_pack", self.flat.data_ptr(), self.pack_table.data_ptr(), self.packed_floats, self.packed.data_ptr(),
                _stream(self.device))
        self.packed_key = key

    def repack_bf16(self, key=None) -> "_Bf16State":
        """bf16 mode (BASELINE cfg 4): (re)build the bf16 fragment stream + fp32 biases from the flat parameters."""
        if self.bf16 is None:
            self.bf16 = _Bf16State(self)
        b = self.bf16
        if key is None or key != b.key:
            _l.call("tnerf_mlp_pack_bf16", C.byref(self.desc), self.flat.data_ptr(), b.table.data_ptr(), b.packed.data_ptr(),
                    _stream(self.device))
            b.key = key
        return b

    def repack_x3(self, key=None, scale_floor: float = 0.0) -> "_X3State":
        """x3 chain (fp32-grade products on the fp16 matrix pipe): (re)build the two-piece record stream, its scale records and the fp32 biases.
        An "auto" model checks the x3 pipe's domain on its first pack and on every 64th after it (one 72-byte read-back)."""
        if self.x3 is None:
            self.x3 = _X3State(self)
        b = self.x3
        if key is None or key != b.key:
            # scale_floor: DatasetTrainer passes 16 lr (include/tnerf.h, tnerf_mlp_pack_x3_floor)
            _l.call("tnerf_mlp_pack_x3_floor", C.byref(self.desc), self.flat.data_ptr(), b.table.data_ptr(), b.packed.data_ptr(),
                    float(scale_floor), _stream(self.device))
            b.key = key
            if self.pipe_policy == "auto" and self._x3_packs % 64 == 0:
                self.check_x3_domain()
            self._x3_packs += 1
        return b

    @property
    def uses_x3(self) -> bool:
        """True if the fused fp32 paths of this model run on the x3 kernels right now."""
        return self.x3_capable and not (self.desc.flags & _l.FLAG_FP32_MFMA)

    def x3_domain(self):
        """Per layer (index depth = the heads): (share of the nonzero weights below 2^-13 max|W_l|, share of the nonzero biases below
        2^-14 max|b_l|) for the parameters as they are now, against the maxima in the x3 stream's scale records (include/tnerf.h,
        tnerf_x3_domain_counts).  Synchronises."""
        b = self.x3 if self.x3 is not None else self.repack_x3()
        counts = torch.zeros(4 * (self.desc.depth + 1), dtype=torch.int32, device=self.device)
        _l.call("tnerf_x3_domain_counts", C.byref(self.desc), self.flat.data_ptr(), b.table.data_ptr(), b.packed.data_ptr(), counts.data_ptr(),
                _stream(self.device))
        c = counts.cpu().numpy().astype(np.int64).reshape(-1, 4)
        return [(float(r[1]) / max(1, int(r[0])), float(r[3]) / max(1, int(r[2]))) for r in c]

    X3_MAX_SMALL_SHARE = 1.0 / 64.0        # of a layer's weights (or biases) that may sit below the x3 pipe's full-precision range

    def check_x3_domain(self) -> bool:
        """True if the weights are inside the x3 pipe's domain.  Otherwise an "auto" model switches to the fp32-MFMA kernels (for good:
        desc.flags gains TNERF_FLAG_FP32_MFMA, every later call of the fused paths takes them) and says so once."""
        if not self.uses_x3:
            return True
        shares = self.x3_domain()
        bad = [(l, w, bb) for l, (w, bb) in enumerate(shares) if w > self.X3_MAX_SMALL_SHARE or bb > 0.5]
        if not bad:
            return True
        if self.pipe_policy == "auto":
            l, w, bb = bad[0]
            self.pipe_switched = (f"layer {l}: {100 * w:.1f} % of the weights are more than 2^13 below max|W| and {100 * bb:.1f} % of the biases more than "
                                  f"2^14 below max|b|")
            self.desc.flags |= _l.FLAG_FP32_MFMA
            import warnings
            warnings.warn("TinyNeRF (HIP): the weights left the domain in which the x3 matrix pipe (three fp16 partial products) is fp32-grade — "
                          + self.pipe_switched + "; this model now runs on the plain fp32-MFMA kernels (about 2.4x slower).  "
                          "matrix_pipe='x3' keeps the x3 pipe regardless.", RuntimeWarning, stacklevel=3)
        return False

    def grad_views(self, params):
        return [self.grad[o:o + p.numel()].view(p.shape) for p, o in zip(params, self.offsets)]


class _Bf16State:
    """Pack table and packed buffer of the bf16 mode for one model (include/tnerf.h, "bf16 mode")."""

    def __init__(self, st: ModelState):
        sz = _l.Bf16Sizes()
        _l.call("tnerf_bf16_plan_sizes", C.byref(st.desc), C.byref(sz))
        tab = np.empty(int(sz.pack_entries), np.int32)
        _l.call("tnerf_bf16_pack_table", C.byref(st.desc), tab.ctypes.data_as(C.c_void_p))
        self.table = torch.from_numpy(tab).to(st.device)
        self.packed = torch.empty(int(sz.packed_bytes), dtype=torch.uint8, device=st.device)
        self.n_fragments = int(sz.n_fragments)
        self.key = None
        self.st = st
        self.plans: Dict[tuple, "_Bf16TrainPlan"] = {}

    def train_plan(self, R: int, S: int) -> "_Bf16TrainPlan":
        p = self.plans.get((R, S))
        if p is None:
            if len(self.plans) >= 2:
                self.plans.pop(next(iter(self.plans)))
            p = self.plans[(R, S)] = _Bf16TrainPlan(self.st, R, S)
        return p


class _X3State:
    """Pack table and packed record stream of the x3 chain kernels for one model (include/tnerf.h, "x3")."""

    def __init__(self, st: ModelState):
        sz = _l.Bf16Sizes()
        _l.call("tnerf_x3_plan_sizes", C.byref(st.desc), C.byref(sz))
        tab = np.empty(int(sz.pack_entries), np.int32)
        _l.call("tnerf_x3_pack_table", C.byref(st.desc), tab.ctypes.data_as(C.c_void_p))
        self.table = torch.from_numpy(tab).to(st.device)
        self.packed = torch.empty(int(sz.packed_bytes), dtype=torch.uint8, device=st.device)
        self.n_fragments = int(sz.n_fragments)
        self.key = None


class _Bf16TrainPlan:
    """Stash, slabs and tables of the bf16 training step for one (rays, samples) batch shape."""

    def __init__(self, st: "ModelState", R: int, S: int):
        dev = st.device
        sz = _l.Bf16TrainPlan()
        _l.call("tnerf_bf16_train_sizes", C.byref(st.desc), int(R), int(S), st.n_cu, C.byref(sz))
        jobs = np.empty(sz.job_ints, np.int32)
        red = np.empty(sz.reduce_ints, np.int32)
        _l.call("tnerf_bf16_train_fill", C.byref(st.desc), int(R), int(S), st.n_cu, jobs.ctypes.data_as(C.c_void_p),
                red.ctypes.data_as(C.c_void_p))
        self.R, self.S, self.n_tiles, self.n_jobs = int(R), int(S), int(sz.n_tiles), int(sz.n_jobs)
        self.jobs = torch.from_numpy(jobs).to(dev)
        self.reduce = torch.from_numpy(red).to(dev)
        self.stash = torch.empty(int(sz.stash_bytes), dtype=torch.uint8, device=dev)
        self.slabs = torch.empty(int(sz.slab_floats), dtype=torch.float32, device=dev)


# ---------------------------------------------------------------------------------- MLP op
def _again(ctx, lease, x3: bool) -> None:
    """A SECOND backward over the same forward (retain_graph=True).  The x3 dgrad kernel keeps running maxima of the activation
    gradients (the weight-gradient kernel's scales) in the stash's bound words TNB_DZ(0) .. TNB_DZH (csrc/tnerf_internal.h: words 17 ..
    33 of the last 64 floats); only a forward clears them, so a repeated backward would scale by the maxima of BOTH gradients and
    not be bit-identical to a first one.  Cleared here, on the (rare) repeat."""
    if getattr(ctx, "n_backward", 0) and x3:
        lease.buf[-47:-30].zero_()
    ctx.n_backward = getattr(ctx, "n_backward", 0) + 1


def _check_versions(ctx, who: str) -> None:
    """The backward reads the packed weights of NOW against the activations of the forward: refuse, like autograd does for saved
    tensors, when a parameter was modified in place in between (an optimizer step before a second backward(retain_graph=True))."""
    # (torch's _version counters see in-place edits made through torch; the framework's own updates — FlatAdam.step, the trainers' steps —
    #  write the flat buffer with raw kernels and bump ModelState.generation instead)
    if tuple(p._version for p in ctx.params) != ctx.versions or ctx.st.generation != ctx.generation:
        raise RuntimeError(f"{who}: one of the parameters needed for gradient computation has been modified by an inplace operation "
                           "since the forward (e.g. optimizer.step() between two backward passes over the same graph)")


class _MlpFn(torch.autograd.Function):
    """TinyNeRF.forward on x[M, in_dim] with parameter gradients (no gradient w.r.t. x)."""

    @staticmethod
    def forward(ctx, st: ModelState, x: torch.Tensor, train: bool, *params):
        dev = st.device
        M = x.shape[0]
        rgb = torch.empty(M, 3, dtype=torch.float32, device=dev)
        sigma = torch.empty(M, 1, dtype=torch.float32, device=dev)
        plan = st.plan(M) if train else None
        lease = plan.lease() if train else None
        # the x3 chain (fp32-grade results, products on the fp16 matrix pipe) whenever the input is a 6L+3 encoding
        x3 = None
        if st.x3_capable and not (st.desc.flags & _l.FLAG_FP32_MFMA):
            x3 = st.repack_x3(("pack", st.packed_key) if st.packed_key is not None else None)
            if not st.uses_x3:                     # an "auto" model has just left the x3 pipe's domain (ModelState.check_x3_domain)
                x3 = None
        _l.call("tnerf_mlp_fwd_x3" if x3 is not None else "tnerf_mlp_fwd", C.byref(st.desc),
                (x3 if x3 is not None else st).packed.data_ptr(), x.data_ptr(), M, rgb.data_ptr(), sigma.data_ptr(),
                lease.buf.data_ptr() if train else None, plan.Mp if train else 0, _stream(dev))
        ctx.st, ctx.plan, ctx.M, ctx.lease, ctx.x3 = st, plan, M, lease, x3
        ctx.shapes = [p.shape for p in params]
        ctx.params, ctx.versions, ctx.generation = params, tuple(p._version for p in params), st.generation
        return rgb, sigma

    @staticmethod
    def backward(ctx, g_rgb, g_sigma):
        st, plan, M, lease = ctx.st, ctx.plan, ctx.M, ctx.lease
        if plan is None:
            raise RuntimeError("TinyNeRF (HIP): backward through a forward that ran without grad enabled")
        _check_versions(ctx, "TinyNeRF (HIP)")
        _again(ctx, lease, ctx.x3 is not None)
        dev = st.device
        g_rgb = torch.zeros(M, 3, dtype=torch.float32, device=dev) if g_rgb is None else _f32c(g_rgb)
        g_sigma = torch.zeros(M, 1, dtype=torch.float32, device=dev) if g_sigma is None else _f32c(g_sigma)
        x3 = ctx.x3
        _l.call("tnerf_mlp_bwd_x3" if x3 is not None else "tnerf_mlp_bwd", C.byref(st.desc),
                (x3 if x3 is not None else st).packed.data_ptr(), M, g_rgb.data_ptr(), g_sigma.data_ptr(),
                lease.buf.data_ptr(), plan.Mp, plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr(), plan.reduce.data_ptr(),
                st.grad.data_ptr(), _stream(dev))
        grads = [st.grad[o:o + int(np.prod(s))].view(s).clone() for s, o in zip(ctx.shapes, st.offsets)]
        return (None, None, None, *grads)          # the stash stays leased until this node dies: backward(retain_graph=True) may come again


def mlp_forward(st: ModelState, x: torch.Tensor, params) -> Tuple[torch.Tensor, torch.Tensor]:
    _need_cuda(x)
    if x.requires_grad:      # the chain kernels form no dL/dx (the reference's points carry no grad); nerf.TinyNeRF.forward routes
        raise NotImplementedError("mlp_forward: no gradient w.r.t. x on the chain kernels; use mlp_forward_generic")   # such calls there
    if x.dim() != 2 or x.shape[1] != st.desc.in_dim:
        raise RuntimeError(f"TinyNeRF (HIP): expected x of shape (N, {st.desc.in_dim}), got {tuple(x.shape)}")
    train = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    return _MlpFn.apply(st, _f32c(x), train, *params)


# ----------------------------------------------------------- any width / gradient w.r.t. the input: layer by layer
class _MlpGenericFn(torch.autograd.Function):
    """TinyNeRF.forward through tnerf_mlp_fwd_generic / tnerf_mlp_bwd_generic (one hipBLAS SGEMM per layer): the shapes the chain
    kernels do not cover (hidden > 256, in_dim > 64) and, for any shape, the gradient w.r.t. x.  The parameter tensors are used
    where they are (no flat buffer, no packed copy); torch lends its hipBLAS handle."""

    @staticmethod
    def forward(ctx, desc, x: torch.Tensor, train: bool, *params):
        dev = x.device
        M = x.shape[0]
        ps = [_f32c(p.detach()) for p in params]
        ptrs = (C.c_void_p * len(ps))(*[p.data_ptr() for p in ps])
        rgb = torch.empty(M, 3, dtype=torch.float32, device=dev)
        sigma = torch.empty(M, 1, dtype=torch.float32, device=dev)
        n_acts = int(_l.load().tnerf_mlp_generic_acts_floats(C.byref(desc), M))
        if n_acts < 0:
            _l.check(n_acts, "tnerf_mlp_generic_acts_floats")
        acts = torch.empty(n_acts, dtype=torch.float32, device=dev)
        _l.call("tnerf_mlp_fwd_generic", C.byref(desc), torch.cuda.current_blas_handle(), ptrs, x.data_ptr(), M, rgb.data_ptr(),
                sigma.data_ptr(), acts.data_ptr(), n_acts, _stream(dev))
        if train:
            ctx.desc, ctx.M = desc, M
            ctx.save_for_backward(x, rgb, sigma, acts, *ps)
            ctx.need_dx = bool(x.requires_grad)
        return rgb, sigma

    @staticmethod
    def backward(ctx, g_rgb, g_sigma):
        x, rgb, sigma, acts, *ps = ctx.saved_tensors
        desc, M, dev = ctx.desc, ctx.M, x.device
        g_rgb = torch.zeros(M, 3, dtype=torch.float32, device=dev) if g_rgb is None else _f32c(g_rgb)
        g_sigma = torch.zeros(M, 1, dtype=torch.float32, device=dev) if g_sigma is None else _f32c(g_sigma)
        n_scr = int(_l.load().tnerf_mlp_generic_scratch_floats(C.byref(desc), M))
        scratch = torch.empty(n_scr, dtype=torch.float32, device=dev)
        grads = [torch.empty_like(p) for p in ps]
        pptr = (C.c_void_p * len(ps))(*[p.data_ptr() for p in ps])
        gptr = (C.c_void_p * len(ps))(*[g.data_ptr() for g in grads])
        dx = torch.empty_like(x) if ctx.need_dx else None
        _l.call("tnerf_mlp_bwd_generic", C.byref(desc), torch.cuda.current_blas_handle(), pptr, x.data_ptr(), M, rgb.data_ptr(),
                sigma.data_ptr(), g_rgb.data_ptr(), g_sigma.data_ptr(), acts.data_ptr(), acts.numel(), scratch.data_ptr(), n_scr, gptr,
                _ptr(dx), _stream(dev))
        return (None, dx, None, *grads)


def mlp_forward_generic(in_dim: int, hidden: int, depth: int, skip_at: int, x: torch.Tensor, params):
    """rgb [M,3], sigma [M,1] of TinyNeRF(in_dim, hidden, depth, skip_at) on x [M, in_dim]; gradients w.r.t. the parameters and,
    when x.requires_grad, w.r.t. x."""
    _need_cuda(x, *params)
    if x.dim() != 2 or x.shape[1] != in_dim:
        raise RuntimeError(f"TinyNeRF (HIP): expected x of shape (N, {in_dim}), got {tuple(x.shape)}")
    desc = _l.MlpDesc(int(in_dim), int(hidden), int(depth), int(skip_at), 0)
    train = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    xc = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
    return _MlpGenericFn.apply(desc, xc, train, *params)


# ------------------------------------------------------------------------------- fused rays op
class _FusedRaysFn(torch.autograd.Function):
    """sample -> encode -> MLP -> composite for a batch of rays in one kernel (+ backward)."""

    @staticmethod
    def forward(ctx, st: ModelState, rays_o, rays_d, ztab, S, rnd, t_rand, seed, off, white, train, *params):
        dev = st.device
        R = rays_o.shape[0]
        comp = torch.empty(R, 3, dtype=torch.float32, device=dev)
        # the x3 chain (fp32-grade results, products on the fp16 matrix pipe) whenever the model allows it; its record stream
        # follows the fp32 pack's key (the caller packed for the current parameter versions)
        x3 = None
        if st.x3_capable and not (st.desc.flags & _l.FLAG_FP32_MFMA):
            x3 = st.repack_x3(("pack", st.packed_key) if st.packed_key is not None else None)
            if not st.uses_x3:                     # an "auto" model has just left the x3 pipe's domain (ModelState.check_x3_domain)
                x3 = None
        if train:
            plan = st.plan(R * S)
            lease = plan.lease()
            if x3 is not None:
                _l.call("tnerf_train_fwd_fused_x3", C.byref(st.desc), x3.packed.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(), R, S,
                        ztab.data_ptr(), rnd, _ptr(t_rand), seed, off, white, comp.data_ptr(), lease.buf.data_ptr(), plan.Mp,
                        _stream(dev))
            else:
                _l.call("tnerf_train_fwd_fused", C.byref(st.desc), st.packed.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(), R, S,
                        ztab.data_ptr(), rnd, _ptr(t_rand), seed, off, white, comp.data_ptr(), lease.buf.data_ptr(), plan.Mp,
                        _stream(dev))
            ctx.save_for_backward(rays_o, rays_d, ztab, t_rand if t_rand is not None else ztab)
            ctx.args = (st, plan, R, S, rnd, t_rand is not None, seed, off, white)
            ctx.st = st
            ctx.x3 = x3
            ctx.lease = lease
            ctx.shapes = [p.shape for p in params]
            ctx.params, ctx.versions, ctx.generation = params, tuple(p._version for p in params), st.generation
            return comp, None, None
        depth = torch.empty(R, 1, dtype=torch.float32, device=dev)
        acc = torch.empty(R, 1, dtype=torch.float32, device=dev)
        _l.call("tnerf_render_fused_x3" if x3 is not None else "tnerf_render_fused", C.byref(st.desc),
                (x3 if x3 is not None else st).packed.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(), R, S,
                ztab.data_ptr(), rnd, _ptr(t_rand), seed, off, white, comp.data_ptr(), depth.data_ptr(), acc.data_ptr(), _stream(dev))
        ctx.args = None
        return comp, depth, acc

    @staticmethod
    def backward(ctx, g_comp, g_depth, g_acc):
        if ctx.args is None:
            raise RuntimeError("fused render (HIP): backward through an inference forward")
        st, plan, R, S, rnd, has_tr, seed, off, white = ctx.args
        rays_o, rays_d, ztab, t_rand = ctx.saved_tensors
        dev = st.device
        g_comp = _f32c(g_comp)
        lease = ctx.lease
        _check_versions(ctx, "fused render (HIP)")
        _again(ctx, lease, ctx.x3 is not None)
        _l.call("tnerf_train_bwd_fused", C.byref(st.desc), st.packed.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(), R, S,
                ztab.data_ptr(), rnd, t_rand.data_ptr() if has_tr else None, seed, off, white, g_comp.data_ptr(),
                lease.buf.data_ptr(), plan.Mp, plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr(), plan.reduce.data_ptr(),
                st.grad.data_ptr(), ctx.x3.packed.data_ptr() if ctx.x3 is not None else None, _stream(dev))
        grads = [st.grad[o:o + int(np.prod(s))].view(s).clone() for s, o in zip(ctx.shapes, st.offsets)]
        return (None,) * 11 + tuple(grads)         # the stash stays leased until this node dies (retain_graph)


def render_rays_fused(st: ModelState, params, rays_o, rays_d, near, far, n_samples, randomized, white_bkgd=True,
                      t_rand=None, philox=None):
    """Returns (comp_rgb [R,3], depth [R,1] | None, acc [R,1] | None).  depth/acc only without grad."""
    dev = _need_cuda(rays_o, rays_d, t_rand)
    rays_o, rays_d = _f32c(rays_o), _f32c(rays_d)
    S = int(n_samples)
    ztab = depth_table(near, far, S, dev)
    rnd, tr, seed, off = _rng_args(randomized, t_rand, philox)
    if tr is not None:
        tr = _f32c(tr)
        if tuple(tr.shape) != (rays_o.shape[0], S):
            raise ValueError(f"t_rand must be ({rays_o.shape[0]},{S})")
    train = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    return _FusedRaysFn.apply(st, rays_o, rays_d, ztab, S, rnd, tr, seed, off, int(bool(white_bkgd)), train, *params)


def camera_struct(c2w: torch.Tensor, H: int, W: int, focal: float, pix_index: Optional[torch.Tensor] = None, pix_first: int = 0):
    """tnerf_camera for the *_cam entry points (keeps the tensors alive through the returned tuple)."""
    _need_cuda(c2w, pix_index)
    c2w = _f32c(c2w)
    if c2w.shape != (4, 4):
        raise ValueError(f"c2w must be (4,4), got {tuple(c2w.shape)}")
    if pix_index is not None:
        if pix_index.dtype != torch.int64:
            pix_index = pix_index.long()
        pix_index = pix_index.contiguous()
    cam = _l.Camera(c2w.data_ptr(), int(H), int(W), float(focal), _ptr(pix_index), int(pix_first))
    return cam, (c2w, pix_index)


@torch.no_grad()
def render_camera_fused(st: ModelState, c2w, H, W, focal, pix_first, n_rays, near, far, n_samples, white_bkgd=True,
                        randomized=False, t_rand=None, philox=None, x3_key=None):
    """Inference render of pixels pix_first .. pix_first+n_rays-1 of one pose: rays are generated in the kernel
    (no get_rays launch, no ray tables).  Returns (comp_rgb, depth, acc).  Runs the x3 chain kernel (fp32 results,
    the matrix work on the bf16 pipe) unless the model's flags select the fp32-MFMA kernels; x3_key names the parameter
    versions its record stream was packed for (None: the key st.packed was packed with)."""
    dev = st.device
    S = int(n_samples)
    ztab = depth_table(near, far, S, dev)
    rnd, tr, seed, off = _rng_args(randomized, t_rand, philox)
    cam, keep = camera_struct(c2w, H, W, focal, None, pix_first)
    comp = torch.empty(n_rays, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(n_rays, 1, dtype=torch.float32, device=dev)
    acc = torch.empty(n_rays, 1, dtype=torch.float32, device=dev)
    if st.x3_capable and not (st.desc.flags & _l.FLAG_FP32_MFMA):
        if x3_key is None and st.packed_key is not None:
            x3_key = ("pack", st.packed_key)
        b = st.repack_x3(x3_key)              # fp32-grade results, products on the fp16 matrix pipe (x3)
    if st.uses_x3:                            # (an "auto" model may have left the x3 pipe's domain in that pack)
        _l.call("tnerf_render_fused_cam_x3", C.byref(st.desc), b.packed.data_ptr(), C.byref(cam), int(n_rays), S, ztab.data_ptr(), rnd,
                _ptr(tr), seed, off, int(bool(white_bkgd)), comp.data_ptr(), depth.data_ptr(), acc.data_ptr(), _stream(dev))
        return comp, depth, acc
    _l.call("tnerf_render_fused_cam", C.byref(st.desc), st.packed.data_ptr(), C.byref(cam), int(n_rays), S, ztab.data_ptr(), rnd,
            _ptr(tr), seed, off, int(bool(white_bkgd)), comp.data_ptr(), depth.data_ptr(), acc.data_ptr(), _stream(dev))
    return comp, depth, acc


# ------------------------------------------------------------------------------------- bf16 mode
@torch.no_grad()
def render_rays_fused_bf16(st: ModelState, rays_o, rays_d, near, far, n_samples, randomized=False, white_bkgd=True,
                           t_rand=None, philox=None, key=None):
    """render_rays_fused (inference) with bf16 weights/activations on MFMA, fp32 accumulate and compositing
    (BASELINE cfg 4).  Returns (comp_rgb [R,3], depth [R,1], acc [R,1])."""
    dev = _need_cuda(rays_o, rays_d, t_rand)
    rays_o, rays_d = _f32c(rays_o), _f32c(rays_d)
    R, S = rays_o.shape[0], int(n_samples)
    ztab = depth_table(near, far, S, dev)
    rnd, tr, seed, off = _rng_args(randomized, t_rand, philox)
    if tr is not None:
        tr = _f32c(tr)
    b = st.repack_bf16(key)
    comp = torch.empty(R, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(R, 1, dtype=torch.float32, device=dev)
    acc = torch.empty(R, 1, dtype=torch.float32, device=dev)
    _l.call("tnerf_render_fused_bf16", C.byref(st.desc), b.packed.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(), R, S,
            ztab.data_ptr(), rnd, _ptr(tr), seed, off, int(bool(white_bkgd)), comp.data_ptr(), depth.data_ptr(), acc.data_ptr(),
            _stream(dev))
    return comp, depth, acc


@torch.no_grad()
def render_camera_fused_bf16(st: ModelState, c2w, H, W, focal, pix_first, n_rays, near, far, n_samples, white_bkgd=True,
                             randomized=False, t_rand=None, philox=None, key=None):
    """render_camera_fused in bf16 mode."""
    dev = st.device
    S = int(n_samples)
    ztab = depth_table(near, far, S, dev)
    rnd, tr, seed, off = _rng_args(randomized, t_rand, philox)
    cam, keep = camera_struct(c2w, H, W, focal, None, pix_first)
    b = st.repack_bf16(key)
    comp = torch.empty(n_rays, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(n_rays, 1, dtype=torch.float32, device=dev)
    acc = torch.empty(n_rays, 1, dtype=torch.float32, device=dev)
    _l.call("tnerf_render_fused_cam_bf16", C.byref(st.desc), b.packed.data_ptr(), C.byref(cam), int(n_rays), S, ztab.data_ptr(), rnd,
            _ptr(tr), seed, off, int(bool(white_bkgd)), comp.data_ptr(), depth.data_ptr(), acc.data_ptr(), _stream(dev))
    return comp, depth, acc


# ------------------------------------------------------------------------------------- x3 chain
@torch.no_grad()
def render_rays_fused_x3(st: ModelState, rays_o, rays_d, near, far, n_samples, randomized=False, white_bkgd=True,
                         t_rand=None, philox=None, key=None):
    """render_rays_fused (inference) with the MLP's products on the fp16 matrix pipe (x3: two-piece operands, three partial
    products): fp32-grade results.  Returns (comp_rgb [R,3], depth [R,1], acc [R,1])."""
    dev = _need_cuda(rays_o, rays_d, t_rand)
    rays_o, rays_d = _f32c(rays_o), _f32c(rays_d)
    R, S = rays_o.shape[0], int(n_samples)
    ztab = depth_table(near, far, S, dev)
    rnd, tr, seed, off = _rng_args(randomized, t_rand, philox)
    if tr is not None:
        tr = _f32c(tr)
    b = st.repack_x3(key)
    comp = torch.empty(R, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(R, 1, dtype=torch.float32, device=dev)
    acc = torch.empty(R, 1, dtype=torch.float32, device=dev)
    _l.call("tnerf_render_fused_x3", C.byref(st.desc), b.packed.data_ptr(), rays_o.data_ptr(), rays_d.data_ptr(), R, S,
            ztab.data_ptr(), rnd, _ptr(tr), seed, off, int(bool(white_bkgd)), comp.data_ptr(), depth.data_ptr(), acc.data_ptr(),
            _stream(dev))
    return comp, depth, acc


@torch.no_grad()
def render_camera_fused_x3(st: ModelState, c2w, H, W, focal, pix_first, n_rays, near, far, n_samples, white_bkgd=True,
                           randomized=False, t_rand=None, philox=None, key=None):
    """render_camera_fused on the x3 chain kernels."""
    dev = st.device
    S = int(n_samples)
    ztab = depth_table(near, far, S, dev)
    rnd, tr, seed, off = _rng_args(randomized, t_rand, philox)
    cam, keep = camera_struct(c2w, H, W, focal, None, pix_first)
    b = st.repack_x3(key)
    comp = torch.empty(n_rays, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(n_rays, 1, dtype=torch.float32, device=dev)
    acc = torch.empty(n_rays, 1, dtype=torch.float32, device=dev)
    _l.call("tnerf_render_fused_cam_x3", C.byref(st.desc), b.packed.data_ptr(), C.byref(cam), int(n_rays), S, ztab.data_ptr(), rnd,
            _ptr(tr), seed, off, int(bool(white_bkgd)), comp.data_ptr(), depth.data_ptr(), acc.data_ptr(), _stream(dev))
    return comp, depth, acc
