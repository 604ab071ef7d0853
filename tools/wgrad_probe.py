#!/usr/bin/env python3
"""Diagnostic: per-workgroup duration of k_wgrad by job class (needs the TN_STAMPS build: TNERF_LIB=...libtnerf_hip_stamps.so)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"))
from tnerf import ops, trainer
import nerf
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = nerf.TinyNeRF(39, 256, 8, 4).to(dev)
with torch.no_grad(): model.sigma[0].bias += 0.5
opt = trainer.FlatAdam(model, lr=5e-4); tr = trainer.FusedTrainer(model, opt, 2.0, 6.0, 64)
R = 4096
g = torch.Generator().manual_seed(1)
d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
o = (-4.0 * d).to(dev); d = d.to(dev); tgt = torch.rand(R, 3, generator=g).to(dev); u = torch.rand(R, 64, generator=g).to(dev)
for _ in range(4):
    tr.step(o, d, tgt, t_rand=u)
torch.cuda.synchronize()
plan = model.hip_state().plan(R * 64)
jobs = plan.jobs.cpu().numpy().reshape(-1, 16)
dt = (jobs[:, 14].astype(np.int64) & 0xffffffff) | (jobs[:, 15].astype(np.int64) << 32)
for c in sorted(set(jobs[:, 10])):
    m = jobs[:, 10] == c
    print(f"class {c}: {m.sum():3d} WGs, tiles {jobs[m][0][4]}x{jobs[m][0][5]} blocks/WG {jobs[m][:,8].min()}-{jobs[m][:,8].max()}  cycles median {np.median(dt[m]):.0f} max {dt[m].max()}  per block {np.median(dt[m] / jobs[m][:,8]):.0f}")
print("kernel critical path (max WG):", dt.max())
