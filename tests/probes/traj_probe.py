import os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import golden_params, load_golden
from oracle import tnerf_oracle as O
from tnerf import ops, trainer, lib
import nerf
dev = torch.device("cuda:0")
for tag in ("4x128", "8x256"):
    cfg, params = golden_params(tag)
    g = load_golden(f"step_{tag}")
    images, poses, focal = g["images"], g["poses"], g["focal"]
    N, H, W, _ = images.shape
    rays = [O.pinhole_rays(H, W, focal, poses[i]) for i in range(N)]
    S = g["u"].shape[-1]
    for flags in (0, lib.FLAG_FP32_MFMA):
        os.environ["TNERF_FP32_PIPE"] = "mfma32" if flags else ""
        m = nerf.TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"]).to(dev)
        with torch.no_grad():
            for p, v in zip(m.parameters(), params): p.copy_(v.to(dev))
        opt = trainer.FlatAdam(m, lr=5e-4); tr = trainer.FusedTrainer(m, opt, 2.0, 6.0, S)
        devs = []
        for step in range(10):
            i = step % N; inds = g["inds"][step]
            ro, rd, tgt = rays[i][0][inds].contiguous(), rays[i][1][inds].contiguous(), images.reshape(N, H * W, 3)[i, inds].contiguous()
            loss, _ = tr.step(ro.to(dev), rd.to(dev), tgt.to(dev), t_rand=g["u"][step].to(dev))
            devs.append(float(loss) / float(g["loss"][step]) - 1)
        print(tag, "flags", m.hip_state().desc.flags, " ".join(f"{d:+.1e}" for d in devs))
