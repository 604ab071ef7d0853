"""
TEST INFRASTRUCTURE — NOT PRODUCT CODE.

CPU fp32 restatement (PyTorch CPU ops) of the TinyNeRF render/train hot path of
avihaig/tiny-nerf-pytorch.  It exists to *check* the HIP path:

  * tests/           compare the HIP C-ABI results against these functions,
  * __graft_entry__.smoke()   one tiny check on cuda:0,
  * bench.py         the `cpu_baseline` leg only (kind "port").

Nothing under tiny-nerf-pytorch_amd/ imports this module; the product path fails
loudly if the HIP library is missing instead of falling back to it.

Pinning: the reference publishes no tests / golden vectors for this path
(SURVEY.md §4).  The restatement is pinned by fixtures generated *here* from the
reference itself (tests/golden/make_golden.py imports /root/reference/src and
stores inputs + outputs as .npz); tests/test_oracle_golden.py checks every
function below against them.

Each function cites the reference file:line it restates (paths relative to
/root/reference).  Everything is written as a pure function over explicit
tensors (weights are passed as a flat list, random numbers are passed in) so
that the same inputs can be fed to the HIP entry points.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------- rays
def pinhole_rays(H: int, W: int, focal: float, c2w: Tensor) -> Tuple[Tensor, Tensor]:
    """One ray per pixel, flat index p = row*W + col.   [src/rays.py:3-33]

    Camera looks down -z; no half-pixel offset (rays.py:21-25); directions are
    rotated by the upper-left 3x3 (rays.py:28-30), unit-normalised (rays.py:31)
    and the origin is the pose translation repeated (rays.py:32).
    """
    c2w = c2w.to(torch.float32)
    col = torch.arange(W, dtype=torch.int64).repeat(H)              # p % W
    row = torch.arange(H, dtype=torch.int64).repeat_interleave(W)   # p // W
    cam = torch.empty(H * W, 3, dtype=torch.float32)
    cam[:, 0] = (col - W * 0.5) / focal
    cam[:, 1] = -(row - H * 0.5) / focal
    cam[:, 2] = -1.0
    world = cam @ c2w[:3, :3].T
    rays_d = torch.nn.functional.normalize(world, dim=-1)
    rays_o = c2w[:3, 3].expand(H * W, 3)
    return rays_o, rays_d


# ------------------------------------------------------------------- novel views
def spiral_poses(c2w_ref: Tensor, n_frames: int = 60, radius: float = 0.3) -> Tensor:
    """Camera path of the GIF: frame k is the reference pose moved by (radius cos a_k, radius sin a_k, 0) in its OWN
    frame, a = linspace(0, 2 pi, n_frames) — both end points included, so the last frame repeats the first.
    [src/camera.py:4-12]   Returns (n_frames, 4, 4)."""
    frames = []
    for a in torch.linspace(0, 2 * math.pi, n_frames):               # camera.py:7
        shift = torch.eye(4, dtype=c2w_ref.dtype)
        shift[:3, 3] = torch.tensor([radius * torch.cos(a), radius * torch.sin(a), 0.0], dtype=c2w_ref.dtype)   # camera.py:8-10
        frames.append(c2w_ref @ shift)                               # camera.py:11
    return torch.stack(frames, dim=0)


def deal_frames(n_frames: int, rank: int, world: int) -> List[int]:
    """Pose-parallel dealing of the novel-view frames (new): rank r renders frames r, r+world, ..."""
    return list(range(rank, n_frames, world))


# ----------------------------------------------------------------------- sampling
def depth_bins(near, far, n_samples: int) -> Tensor:
    """Un-jittered depths z_i = near(1-t_i) + far t_i, t = linspace(0,1,S).  [src/sampling.py:16-17]"""
    t = torch.linspace(0.0, 1.0, steps=n_samples)
    return near * (1.0 - t) + far * t


def stratified(near, far, n_samples: int, rays_o: Tensor, rays_d: Tensor,
               t_rand: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Depths + 3-D points along each ray.   [src/sampling.py:3-28]

    `t_rand` (R,S) in [0,1) replaces the reference's `torch.rand_like` draw
    (sampling.py:24); None means randomized=False (sampling.py:20).  near / far: floats or
    tensors broadcastable to (R,1) (sampling.py:8).
    """
    R = rays_o.shape[0]
    z = depth_bins(near, far, n_samples).expand(R, n_samples)
    if t_rand is not None:
        centre = 0.5 * (z[:, :-1] + z[:, 1:])                        # sampling.py:21
        hi = torch.cat([centre, z[:, -1:]], dim=-1)                  # sampling.py:22
        lo = torch.cat([z[:, :1], centre], dim=-1)                   # sampling.py:23
        z = lo + (hi - lo) * t_rand                                  # sampling.py:25
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]     # sampling.py:27
    return z, pts


# ----------------------------------------------------------------------- encoding
def posenc_dim(num_freqs: int, include_input: bool = True) -> int:
    """[src/encoding.py:16-19]"""
    return 6 * num_freqs + (3 if include_input else 0)


def posenc(x: Tensor, num_freqs: int, include_input: bool = True) -> Tensor:
    """[x, sin(2^k x), cos(2^k x)]_{k<L}; frequency-major, sin then cos, xyz innermost.
    [src/encoding.py:14,26-33]"""
    if x.shape[-1] != 3:
        raise AssertionError("PositionalEncoding expects (..., 3)")
    cols = [x] if include_input else []
    for k in range(num_freqs):
        f = float(2.0 ** k)
        cols += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(cols, dim=-1)


# ---------------------------------------------------------------------------- MLP
def mlp_shapes(in_dim: int, hidden: int, depth: int, skip_at: int) -> List[Tuple[int, ...]]:
    """Parameter shapes in state_dict order: layers.i.{weight,bias}, sigma.0.*, rgb.0.*.
    [src/nerf.py:18-27]"""
    shapes: List[Tuple[int, ...]] = []
    fan_in = in_dim
    for i in range(depth):
        shapes += [(hidden, fan_in), (hidden,)]
        fan_in = hidden + in_dim if i == skip_at - 1 else hidden
    shapes += [(1, hidden), (1,), (3, hidden), (3,)]
    return shapes


def mlp_init(in_dim: int, hidden: int, depth: int, skip_at: int, generator: Optional[torch.Generator] = None) -> List[Tensor]:
    """nn.Linear default init, U(+-1/sqrt(fan_in)) for weight and bias.  (Own generator, so this is
    *distributionally* the reference init, not bit-identical — fixtures carry real reference weights.)"""
    out = []
    shapes = mlp_shapes(in_dim, hidden, depth, skip_at)
    for wi in range(0, len(shapes), 2):
        bound = 1.0 / math.sqrt(shapes[wi][1])
        out.append((torch.rand(shapes[wi], generator=generator) * 2 - 1) * bound)
        out.append((torch.rand(shapes[wi + 1], generator=generator) * 2 - 1) * bound)
    return out


def mlp_forward(params: Sequence[Tensor], x: Tensor, skip_at: int) -> Tuple[Tensor, Tensor]:
    """ReLU MLP with one skip concat; sigmoid rgb head, ReLU sigma head.   [src/nerf.py:29-41]

    `params` in state_dict order (see mlp_shapes).  Returns rgb (M,3), sigma (M,1).
    """
    depth = (len(params) - 4) // 2
    h = x
    for i in range(depth):
        h = torch.relu(torch.nn.functional.linear(h, params[2 * i], params[2 * i + 1]))   # nerf.py:36
        if i == skip_at - 1:
            h = torch.cat([h, x], dim=-1)                                                  # nerf.py:38
    w_s, b_s, w_c, b_c = params[2 * depth: 2 * depth + 4]
    rgb = torch.sigmoid(torch.nn.functional.linear(h, w_c, b_c))                           # nerf.py:39
    sigma = torch.relu(torch.nn.functional.linear(h, w_s, b_s))                            # nerf.py:40
    return rgb, sigma


# ----------------------------------------------------------------------- composite
def composite(rgb: Tensor, sigma: Tensor, z_vals: Tensor, rays_d: Tensor, white_bkgd: bool = True):
    """Alpha compositing; returns the reference's 4-tuple (comp_rgb, depth, acc, weights).
    [src/volume.py:18-44]"""
    gap = z_vals[..., 1:] - z_vals[..., :-1]
    gap = torch.cat([gap, torch.full_like(gap[..., :1], 1e10)], dim=-1)          # volume.py:18-21
    gap = gap * torch.linalg.norm(rays_d[:, None, :], dim=-1)                     # volume.py:23
    alpha = 1.0 - torch.exp(-sigma.squeeze(-1) * gap)                             # volume.py:27
    survive = torch.cumprod(1.0 - alpha + 1e-10, dim=-1)                          # volume.py:30-31
    trans = torch.cat([torch.ones_like(survive[..., :1]), survive[..., :-1]], dim=-1)  # volume.py:32
    w = alpha * trans                                                             # volume.py:34
    colour = (w[..., None] * rgb).sum(dim=-2)                                     # volume.py:36
    depth = (w * z_vals).sum(dim=-1, keepdim=True)                                # volume.py:37
    acc = w.sum(dim=-1, keepdim=True)                                             # volume.py:38
    if white_bkgd:
        colour = colour + (1.0 - acc)                                             # volume.py:42
    return colour, depth, acc, w


def psnr_from_mse(mse: Tensor) -> Tensor:
    """[src/utils.py:14-15]"""
    return -10.0 * torch.log10(mse.clamp_min(1e-10))


# ------------------------------------------------------------------ render / train
def render_rays(params, skip_at, num_freqs, rays_o, rays_d, near, far, n_samples, t_rand=None,
                include_input=True, white_bkgd=True):
    """sample -> encode -> MLP -> composite for a batch of rays (the body shared by
    src/train.py:51-56 and src/train.py:114-121)."""
    R = rays_o.shape[0]
    z, pts = stratified(near, far, n_samples, rays_o, rays_d, t_rand)
    rgb, sigma = mlp_forward(params, posenc(pts.reshape(-1, 3), num_freqs, include_input), skip_at)
    return composite(rgb.reshape(R, n_samples, 3), sigma.reshape(R, n_samples, 1), z, rays_d, white_bkgd)


# ----------------------------------------------------------------------- bf16 mode (BASELINE cfg 4)
def bf16_round(x: Tensor) -> Tensor:
    """fp32 -> nearest-even bf16 -> fp32 (what v_cvt_pk_bf16_f32 does to an MFMA operand)."""
    return x.to(torch.bfloat16).to(x.dtype)


def mlp_forward_bf16(params: Sequence[Tensor], x: Tensor, skip_at: int) -> Tuple[Tensor, Tensor]:
    """mlp_forward with the numerics of the bf16 mode (SURVEY.md 8d cfg 4): weights, the network input and every
    hidden activation rounded to bf16, products accumulated in fp32 (fp64 here: products of bf16 pairs are exact, only
    the summation order differs), biases / sigmoid / head ReLU in fp32.  Not a reference function: the reference has no
    reduced-precision path; this restates tiny-nerf-pytorch_amd/csrc/mlp16_core.hpp on the CPU for the parity tests."""
    depth = (len(params) - 4) // 2
    xb = bf16_round(x)
    h = xb

    def lin(inp, w, b):
        return (inp.double() @ bf16_round(w).double().t()).float() + b

    for i in range(depth):
        h = bf16_round(torch.relu(lin(h, params[2 * i], params[2 * i + 1])))
        if i == skip_at - 1:
            h = torch.cat([h, xb], dim=-1)
    w_s, b_s, w_c, b_c = params[2 * depth: 2 * depth + 4]
    return torch.sigmoid(lin(h, w_c, b_c)), torch.relu(lin(h, w_s, b_s))


def render_rays_bf16(params, skip_at, num_freqs, rays_o, rays_d, near, far, n_samples, t_rand=None,
                     include_input=True, white_bkgd=True):
    """render_rays with mlp_forward_bf16: sampling, encoding and compositing stay fp32."""
    R = rays_o.shape[0]
    z, pts = stratified(near, far, n_samples, rays_o, rays_d, t_rand)
    rgb, sigma = mlp_forward_bf16(params, posenc(pts.reshape(-1, 3), num_freqs, include_input), skip_at)
    return composite(rgb.reshape(R, n_samples, 3), sigma.reshape(R, n_samples, 1), z, rays_d, white_bkgd)


@torch.no_grad()
def render_image(params, skip_at, num_freqs, H, W, focal, pose, n_samples=64, near=2.0, far=6.0,
                 chunk=8192) -> Tensor:
    """Chunked full-image render, clamp to [0,1].   [src/train.py:36-59]"""
    rays_o, rays_d = pinhole_rays(H, W, focal, pose)
    rows = []
    for s in range(0, H * W, chunk):
        c, _, _, _ = render_rays(params, skip_at, num_freqs, rays_o[s:s + chunk], rays_d[s:s + chunk],
                                 near, far, n_samples, None)
        rows.append(c)
    return torch.cat(rows, dim=0).reshape(H, W, 3).clamp(0.0, 1.0)


def loss_and_grads(params, skip_at, num_freqs, rays_o, rays_d, target, near, far, n_samples, t_rand,
                   loss_denominator: Optional[int] = None):
    """Forward + backward of one minibatch: MSE over R*3 elements.   [src/train.py:114-126]

    `loss_denominator` (default R*3) lets a ray shard use the *global* normaliser so that the SUM of
    shard gradients equals the full-batch gradient.
    Returns (loss, psnr, [grad per param]).
    """
    leaves = [p.detach().clone().requires_grad_(True) for p in params]
    comp, _, _, _ = render_rays(leaves, skip_at, num_freqs, rays_o, rays_d, near, far, n_samples, t_rand)
    sq = (comp - target) ** 2
    loss = sq.mean() if loss_denominator is None else sq.sum() / float(loss_denominator)   # train.py:122
    grads = torch.autograd.grad(loss, leaves)
    return loss.detach(), psnr_from_mse(loss.detach()), [g.detach() for g in grads]


def loss_and_grads_bf16(params, skip_at, num_freqs, rays_o, rays_d, target, near, far, n_samples, t_rand,
                        loss_denominator: Optional[int] = None):
    """loss_and_grads with the numerics of the bf16 training kernels (mlp16_fwd.hip / mlp16_bwd.hip), written out by
    hand: bf16 weights, inputs, hidden activations H_l and activation gradients dZ_l (incl. the head gradients), every
    contraction accumulated exactly (fp64 here, fp32 on the GPU), compositing forward/backward in fp32, fp32 weight
    gradients.  Not a reference function (the reference has no reduced-precision path).
    Returns (loss, psnr, [grad per param])."""
    depth = (len(params) - 4) // 2
    R = rays_o.shape[0]
    z, pts = stratified(near, far, n_samples, rays_o, rays_d, t_rand)
    xb = bf16_round(posenc(pts.reshape(-1, 3), num_freqs, True))
    Wb = [bf16_round(params[2 * i]) for i in range(depth)]
    w_s, b_s, w_c, b_c = params[2 * depth: 2 * depth + 4]
    Wh = bf16_round(torch.cat([w_c, w_s], 0))                                   # rows r,g,b,sigma
    bh = torch.cat([b_c, b_s], 0)

    def mm(a, b):                                                               # exact products, exact-enough sums
        return (a.double() @ b.double()).float()

    ins, hs = [], []
    inp = xb
    for i in range(depth):
        ins.append(inp)
        h = bf16_round(torch.relu(mm(inp, Wb[i].t()) + params[2 * i + 1]))
        hs.append(h)
        inp = torch.cat([h, xb], -1) if i == skip_at - 1 else h
    zh = mm(hs[-1], Wh.t()) + bh
    rgb = torch.sigmoid(zh[:, :3]).requires_grad_(True)
    sigma = torch.relu(zh[:, 3:4]).requires_grad_(True)
    comp = composite(rgb.reshape(R, n_samples, 3), sigma.reshape(R, n_samples, 1), z, rays_d, True)[0]
    sq = (comp - target) ** 2
    loss = sq.mean() if loss_denominator is None else sq.sum() / float(loss_denominator)
    d_rgb, d_sigma = torch.autograd.grad(loss, [rgb, sigma])
    rgb, sigma = rgb.detach(), sigma.detach()
    dzh = bf16_round(torch.cat([d_rgb * (rgb * (1.0 - rgb)), d_sigma * (sigma > 0).float()], -1))
    grads = [None] * len(params)
    gWh = mm(dzh.t(), hs[-1]); gbh = dzh.double().sum(0).float()
    grads[2 * depth] = gWh[3:4]; grads[2 * depth + 1] = gbh[3:4]
    grads[2 * depth + 2] = gWh[:3]; grads[2 * depth + 3] = gbh[:3]
    hidden = params[1].numel()
    dz = bf16_round(mm(dzh, Wh) * (hs[-1] > 0).float())
    for i in range(depth - 1, -1, -1):
        grads[2 * i] = mm(dz.t(), ins[i]); grads[2 * i + 1] = dz.double().sum(0).float()
        if i > 0:
            dz = bf16_round(mm(dz, Wb[i][:, :hidden]) * (hs[i - 1] > 0).float())
    return loss.detach(), psnr_from_mse(loss.detach()), grads


class AdamState:
    """torch.optim.Adam defaults as used by the reference (lr from Config, betas (0.9,0.999),
    eps 1e-8, no weight decay).   [src/train.py:80]   Written out explicitly so the fused HIP
    Adam can be compared against something that is not itself torch.optim."""

    def __init__(self, params: Sequence[Tensor], lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, betas[0], betas[1], eps
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = 0

    @torch.no_grad()
    def step(self, params: Sequence[Tensor], grads: Sequence[Tensor]) -> None:
        self.t += 1
        c1 = 1.0 - self.b1 ** self.t
        c2 = 1.0 - self.b2 ** self.t
        for p, g, m, v in zip(params, grads, self.m, self.v):
            m.lerp_(g, 1.0 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            denom = (v.sqrt() / math.sqrt(c2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-self.lr / c1)


# ------------------------------------------------------------------- fp64 variants
def to64(ts):
    return [t.double() for t in ts]
