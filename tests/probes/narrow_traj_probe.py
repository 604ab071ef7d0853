#!/usr/bin/env python3
"""Diagnostic: three DatasetTrainer steps of a narrow net against the oracle on the emulated draws, per matrix pipe: where the
largest weight deviation sits and what the gradients / Adam updates there were."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import tnerf_oracle as O
from tnerf import ops, trainer
import nerf, data
from test_gpu_round2 import _emulated_draws
dev = torch.device("cuda:0")
sc = data.make_synthetic_scene(n_images=5, H=20, W=20, focal=138.88887889922103 * 0.2, seed=4)
images, poses, focal = torch.from_numpy(sc["images"]), torch.from_numpy(sc["poses"]), float(sc["focal"])
N, H, W, _ = images.shape; pixs = images.reshape(N, H * W, 3)
S, Rg, seed = 40, 64, 9
for arch in ((39, 200, 3, 2), (63, 64, 4, 2)):
    in_dim, hidden, depth, skip = arch; L = (in_dim - 3) // 6
    for pipe in ("x3", "fp32_mfma"):
        if os.environ.get("POISON"):            # make every torch.empty below return NaN-filled memory: finds reads of unwritten workspace
            junk = [torch.full((n,), float("nan"), device=dev) for n in (1 << 26, 1 << 22, 1 << 18, 1 << 14, 1 << 10)]
            del junk
        torch.manual_seed(1)
        m = nerf.TinyNeRF(in_dim, hidden, depth, skip, matrix_pipe=pipe).to(dev)
        with torch.no_grad(): m.sigma[0].bias += 0.5
        params = [p.detach().cpu().clone() for p in m.parameters()]
        opt = trainer.FlatAdam(m, lr=5e-4)
        tr = trainer.DatasetTrainer(m, opt, images.to(dev), poses.to(dev), focal, Rg, S, 2.0, 6.0, seed=seed, precision="fp32", record_pixels=True)
        ps = [p.clone() for p in params]; adam = O.AdamState(ps, lr=5e-4)
        hist = []
        for s in range(3):
            loss, _ = tr.step(); torch.cuda.synchronize()
            pix, u = _emulated_draws(seed, s, Rg, S, H * W)
            ro, rd = O.pinhole_rays(H, W, focal, poses[s % N])
            lo_, _, grads = O.loss_and_grads(ps, skip, L, ro[pix], rd[pix], pixs[s % N, pix], 2.0, 6.0, S, u)
            gh = m.hip_state().grad.cpu().clone(); go = torch.cat([x.reshape(-1) for x in grads])
            hist.append((gh, go))
            nb = m.hip_state().flat.isnan().nonzero().flatten().cpu()
            if bool(gh.isnan().any()) or nb.numel():
                print(f"    step {s}: NaN gradients at {gh.isnan().nonzero().flatten()[:8].tolist()} ({int(gh.isnan().sum())}), NaN weights at {nb[:8].tolist()} ({nb.numel()}), loss {float(loss)}")
            adam.step(ps, grads)
        wh = torch.cat([p.detach().cpu().reshape(-1) for p in m.parameters()]); wo = torch.cat([q.reshape(-1) for q in ps])
        e = (wh - wo).abs(); k = int(e.argmax())
        print(f"{arch} {pipe}: max weight deviation {float(e.max()):.2e} at flat index {k}; #elements > 1e-4: {int((e > 1e-4).sum())}")
        for s, (gh, go) in enumerate(hist):
            print(f"    step {s}: grad there hip {float(gh[k]):+.3e} oracle {float(go[k]):+.3e}   (grad relmax whole vector {float((gh - go).abs().max() / go.abs().max()):.1e})")
