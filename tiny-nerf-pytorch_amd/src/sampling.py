"""Drop-in for the reference's src/sampling.py (tnerf_sample_encode_fwd, csrc/stage_kernels.hip)."""
import torch

from _hip import ops


def stratified_samples(near, far, n_samples, rays_o, rays_d, randomized=True):
    """n_samples depths in [near, far] per ray (+ jitter inside each bin when randomized) and the
    points o + d*z.  Returns (z_vals (R,S), pts (R,S,3)).            [reference src/sampling.py:3-28]

    The sample bins are bit-exact with the reference's CPU arithmetic (host-built depth table, un-fused
    lerp in the kernel).  The jitter is drawn with torch.rand on the rays' device — the same generator
    call the reference makes with rand_like (sampling.py:24).
    """
    t_rand = None
    if randomized:
        t_rand = torch.rand(rays_o.shape[0], int(n_samples), dtype=torch.float32, device=rays_o.device)
    if isinstance(near, torch.Tensor) or isinstance(far, torch.Tensor):      # "tensors broadcastable to (N_rays, 1)", sampling.py:8
        z_vals, pts = ops.sample_along_rays_per_ray(near, far, int(n_samples), rays_o, rays_d, bool(randomized), t_rand)
    else:
        z_vals, pts, _ = ops.sample_along_rays(float(near), float(far), int(n_samples), rays_o, rays_d, bool(randomized), t_rand)
    # tensor bounds that require grad: the depths are linear in them (reference sampling.py:17,25 is plain autograd)
    z_vals = ops.attach_depth_grad(near, far, z_vals, t_rand)
    # rays that require grad (a learned pose): pts = o + d z is differentiable in o and d — and, through z, in the bounds — as in
    # the reference (sampling.py:27)
    return z_vals, ops.attach_points_grad(rays_o, rays_d, z_vals, pts)
