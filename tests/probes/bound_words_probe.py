#!/usr/bin/env python3
"""Diagnostic: which of the stash's 64 bound words a DatasetTrainer step zeroes / writes (the stash is filled with a marker first)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"))
import torch
from tnerf import trainer
import nerf, data
dev = torch.device("cuda:0")
sc = data.make_synthetic_scene(n_images=5, H=20, W=20, focal=138.88887889922103 * 0.2, seed=4)
images, poses, focal = torch.from_numpy(sc["images"]).to(dev), torch.from_numpy(sc["poses"]).to(dev), float(sc["focal"])
for arch, graph in (((39, 200, 3, 2), False), ((39, 256, 8, 4), False), ((39, 256, 8, 4), True)):
    torch.manual_seed(1)
    m = nerf.TinyNeRF(*arch).to(dev)
    tr = trainer.DatasetTrainer(m, trainer.FlatAdam(m, lr=5e-4), images, poses, focal, 64, 40, 2.0, 6.0, seed=9, precision="fp32", graph=graph)
    nb = tr._stash.numel() - 64
    print(arch, "graph" if graph else "eager", "stash floats", tr._stash.numel(), "plan says", tr._plan.stash.numel())
    for s in range(3):
        tr._stash[nb - 8:].fill_(777.0); torch.cuda.synchronize()
        tr.step(); torch.cuda.synchronize()
        b = tr._stash[nb - 8:].cpu()
        print(f"  step {s}: 8 words before:", [f"{float(x):.3g}" for x in b[:8]])
        print("     bound words:", " ".join(f"{i}:{float(x):.3g}" for i, x in enumerate(b[8:])))
