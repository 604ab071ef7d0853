"""Drop-in for the reference's src/volume.py (tnerf_composite_fwd / _bwd, csrc/stage_kernels.hip)."""
from _hip import ops


def volume_render(rgb, sigma, z_vals, rays_d, white_bkgd=True):
    """Alpha compositing along each ray: one ray per wavefront, exclusive transmittance scan by wave
    shuffles.  rgb (R,S,3), sigma (R,S,1), z_vals (R,S), rays_d (R,3).
    Returns the reference's 4-tuple (comp_rgb (R,3), depth (R,1), acc (R,1), weights (R,S)).
    [reference src/volume.py:3-44]"""
    return ops.volume_render(rgb, sigma, z_vals, rays_d, white_bkgd)
