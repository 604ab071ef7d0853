#!/usr/bin/env python3
"""
Round-2 additions to the golden fixtures, generated FROM THE REFERENCE ITSELF like make_golden.py (same rules: build
container only, reference modules imported on CPU fp32, fixtures are inputs + outputs, no source text):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r2.py

  sampling_per_ray.npz  stratified_samples with near / far given as tensors broadcastable to (N_rays, 1)
                        (reference src/sampling.py:8,17): (R,1) tensors, a 0-dim tensor mixed with a float, randomized and not
  novel_views.npz       the novel-view path (reference src/make_gif.py:22-27): spiral_poses(poses[0], 60, 0.3) and three of its
                        frames rendered by the reference functions in render_one's order (src/train.py:45-58) with the
                        weights_4x128 fixture weights at 40x40
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G          # noqa: E402  (sets up the reference import path; nothing runs at import)
from make_golden import get_rays, stratified_samples, PositionalEncoding, TinyNeRF, volume_render, camera, save, FOCAL, poses3  # noqa: E402


def fx_sampling_per_ray():
    g = torch.Generator().manual_seed(111)
    ro, rd = get_rays(100, 100, FOCAL, poses3()[1])
    pick = torch.randperm(10000, generator=g)[:80]
    ro, rd = ro[pick].contiguous(), rd[pick].contiguous()
    near = 1.5 + torch.rand(80, 1, generator=g)                 # (R,1)
    far = 5.0 + 2.0 * torch.rand(80, 1, generator=g)
    out = {"rays_o": ro, "rays_d": rd, "near": near, "far": far}
    for S in (64, 33):
        z, pts = stratified_samples(near, far, S, ro, rd, randomized=False)
        out[f"z_det_{S}"], out[f"pts_det_{S}"] = z.contiguous(), pts[:16].contiguous()
        torch.manual_seed(2000 + S)
        z, pts = stratified_samples(near, far, S, ro, rd, randomized=True)
        torch.manual_seed(2000 + S)
        out[f"u_{S}"], out[f"z_rand_{S}"], out[f"pts_rand_{S}"] = torch.rand_like(z), z, pts[:16].contiguous()
    # a 0-dim tensor mixed with a python float
    near0 = torch.tensor(2.25)
    torch.manual_seed(31)
    z, pts = stratified_samples(near0, 6.5, 64, ro, rd, randomized=True)
    torch.manual_seed(31)
    out["near0"], out["far0"] = near0, np.float64(6.5)
    out["u_mixed"], out["z_mixed"], out["pts_mixed"] = torch.rand_like(z), z, pts[:16].contiguous()
    save("sampling_per_ray", **out)


def fx_novel_views():
    w = np.load(os.path.join(HERE, "weights_4x128.npz"))
    L, hidden, depth, skip_at = (int(v) for v in w["cfg"])
    enc = PositionalEncoding(L, True)
    model = TinyNeRF(enc.out_dim, hidden, depth, skip_at)
    with torch.no_grad():
        for i, p in enumerate(model.parameters()):
            p.copy_(torch.from_numpy(w[f"p{i:02d}"]))
    ref = poses3()[1]
    path = camera.spiral_poses(ref, n_frames=60, radius=0.3)                 # make_gif.py:22
    H = W = 40
    focal = FOCAL * W / 100.0
    frames, ks = [], (0, 17, 44, 59)
    with torch.no_grad():
        for k in ks:                                                          # render_one, train.py:45-58
            rays_o, rays_d = get_rays(H, W, focal, path[k])
            outs = []
            for i in range(0, H * W, 8192):
                zz, pp = stratified_samples(2.0, 6.0, 64, rays_o[i:i + 8192], rays_d[i:i + 8192], randomized=False)
                c_rgb, c_sig = model(enc(pp.reshape(-1, 3)))
                comp, _, _, _ = volume_render(c_rgb.reshape(pp.shape[0], 64, 3), c_sig.reshape(pp.shape[0], 64, 1), zz, rays_d[i:i + 8192])
                outs.append(comp)
            frames.append(torch.cat(outs, 0).reshape(H, W, 3).clamp(0., 1.))
    save("novel_views", ref=ref, path=path, H=np.int64(H), W=np.int64(W), focal=np.float64(focal),
         frame_index=np.array(ks), frames=torch.stack(frames),
         frames_u8=(torch.stack(frames).numpy() * 255).astype(np.uint8))     # make_gif.py:26


if __name__ == "__main__":
    print("torch", torch.__version__, "reference at", G.REF)
    fx_sampling_per_ray(); fx_novel_views()
