#!/usr/bin/env python3
"""Diagnostic: per-tensor gradient errors against fp64 of the ragged 8x256 / 130 rays x 128 samples case of
tests/test_gpu_parity.py::test_fused_ragged_shapes_forward_and_gradients for both matrix pipes and for the CPU fp32 oracle, and
which samples carry the worst element (the 1e10 tail sample makes d sigma ill-conditioned)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import golden_params
from oracle import tnerf_oracle as O
from tnerf import ops
import nerf
dev = torch.device("cuda:0")
relmax = lambda a, b: float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)
for tag, R, S in (("8x256", 130, 128), ("8x256", 37, 50), ("4x128", 66, 256)):
    cfg, params = golden_params(tag)
    gen = torch.Generator().manual_seed(R * 1000 + S)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
    o = -4.0 * d + 0.3 * torch.randn(R, 3, generator=gen)
    tgt, u = torch.rand(R, 3, generator=gen), torch.rand(R, S, generator=gen)
    _, _, g32 = O.loss_and_grads(params, cfg["skip_at"], cfg["L"], o, d, tgt, 2.0, 6.0, S, u)
    _, _, g64 = O.loss_and_grads([p.double() for p in params], cfg["skip_at"], cfg["L"], o.double(), d.double(), tgt.double(), 2.0, 6.0, S, u.double())
    rows = {"cpu fp32": [relmax(a.double(), b) for a, b in zip(g32, g64)]}
    for pipe in ("x3", "fp32_mfma"):
        m = nerf.TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"], matrix_pipe=pipe).to(dev)
        with torch.no_grad():
            for p, v in zip(m.parameters(), params): p.copy_(v.to(dev))
        st, plist = m._ensure_packed(), m._param_list()
        comp, _, _ = ops.render_rays_fused(st, plist, o.to(dev), d.to(dev), 2.0, 6.0, S, True, t_rand=u.to(dev))
        torch.mean((comp - tgt.to(dev)) ** 2).backward()
        rows[pipe] = [relmax(p.grad.cpu().double(), b) for p, b in zip(plist, g64)]
    print(f"{tag} R={R} S={S}: per-tensor max|d| / max|g| against fp64 (weights then bias, layer by layer)")
    for k, v in rows.items():
        print(f"   {k:10s} worst {max(v):.2e} | " + " ".join(f"{x:.1e}" for x in v))
