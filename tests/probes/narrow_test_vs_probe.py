#!/usr/bin/env python3
"""Diagnostic: call the pytest function test_hidden_widths_other_than_128_and_256 directly and record both trajectories it compares."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_round2 as T
from oracle import tnerf_oracle as O
import rays, sampling, encoding, nerf, volume, utils, train, camera, data, make_gif
from tnerf import ops, trainer, lib
mods = dict(rays=rays, sampling=sampling, encoding=encoding, nerf=nerf, volume=volume, utils=utils, train=train, camera=camera, data=data, make_gif=make_gif, ops=ops, trainer=trainer, lib=lib)
dev = torch.device("cuda:0")
log = []
orig_step = trainer.DatasetTrainer.step
def step(self, *a, **k):
    out = orig_step(self, *a, **k); torch.cuda.synchronize()
    st = self.model.hip_state() if hasattr(self, "model") else self.st
    log.append(("hip", self.precision, st.grad.cpu().clone(), st.flat.cpu().clone(), float(out[0])))
    return out
trainer.DatasetTrainer.step = step
orig_adam = O.AdamState.step
def astep(self, ps, grads):
    r = orig_adam(self, ps, grads)
    log.append(("oracle", None, torch.cat([g.reshape(-1) for g in grads]).clone(), torch.cat([p.reshape(-1) for p in ps]).clone(), None))
    return r
O.AdamState.step = astep
arch = (39, 200, 3, 2)
try:
    T.test_hidden_widths_other_than_128_and_256.__wrapped__(mods, dev, arch) if hasattr(T.test_hidden_widths_other_than_128_and_256, "__wrapped__") else T.test_hidden_widths_other_than_128_and_256(mods, dev, arch)
    print("test body passed")
except AssertionError as e:
    print("test body failed:", str(e)[:300])
hip = [l for l in log if l[0] == "hip" and l[1] == "fp32"]; ora = [l for l in log if l[0] == "oracle"]
for s in range(min(len(hip), 3)):
    gh, wh = hip[s][2], hip[s][3]; go, wo = ora[s][2], ora[s][3]
    print(f"step {s}: loss {hip[s][4]:.6f}  grad relmax {float((gh - go).abs().max() / go.abs().max()):.2e}  weights after: max dev {float((wh - wo).abs().max()):.3e}  #dev>1e-5 {int(((wh - wo).abs() > 1e-5).sum())} of {wh.numel()}")
