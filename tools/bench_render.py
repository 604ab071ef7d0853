#!/usr/bin/env python3
"""Full-image render throughput of the fused kernel (BASELINE.json configs 3 and 5 shapes, one GPU's share):
   python tools/bench_render.py            -> 100x100/S=64, 400x400/S=128, 800x800/S=256 with the 8x256 L=6 model."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"))
import nerf, train
from encoding import PositionalEncoding
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = PositionalEncoding(6, True).to(dev)
model = nerf.TinyNeRF(39, 256, 8, 4).to(dev)
with torch.no_grad():
    model.sigma[0].bias += 0.5
pose = torch.eye(4, device=dev); pose[2, 3] = 4.0
MACS = 39 * 256 + 7 * 256 * 256 + 39 * 256 + 4 * 256
out = []
for (H, S, chunk) in ((100, 64, 8192), (400, 128, 8192), (400, 128, 160000), (800, 256, 80000)):
    focal = 138.88887889922103 * H / 100
    for _ in range(2):
        img = train.render_one(model, enc, H, H, focal, pose, dev, n_samples=S, chunk=chunk)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 3
    for _ in range(n):
        img = train.render_one(model, enc, H, H, focal, pose, dev, n_samples=S, chunk=chunk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    rays = H * H
    tf = 2 * MACS * S * rays / dt / 1e12
    out.append(dict(image=f"{H}x{H}", samples=S, chunk=chunk, ms=dt * 1e3, rays_per_s=rays / dt, tflops=tf, mfma_frac=tf / 157.3,
                    finite=bool(torch.isfinite(img).all())))
    print(json.dumps(out[-1]))
