#!/usr/bin/env python3
"""Diagnostic: first-step gradient of narrow random-init nets (the shapes of test_hidden_widths_other_than_128_and_256) against
fp64 for both matrix pipes: whole vector, worst element relative to the tensor maximum, and the elements that matter to Adam
(|g| around eps = 1e-8)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import tnerf_oracle as O
from tnerf import ops
import nerf
dev = torch.device("cuda:0")
for (in_dim, hidden, depth, skip) in ((39, 200, 3, 2), (63, 64, 4, 2), (39, 100, 2, 0), (27, 31, 3, 1)):
    L = (in_dim - 3) // 6; R, S = 64, 40
    gen = torch.Generator().manual_seed(3)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1); o = -4.0 * d + 0.3 * torch.randn(R, 3, generator=gen)
    tgt, u = torch.rand(R, 3, generator=gen), torch.rand(R, S, generator=gen)
    out = {}
    for pipe in ("x3", "fp32_mfma"):
        torch.manual_seed(1)
        m = nerf.TinyNeRF(in_dim, hidden, depth, skip, matrix_pipe=pipe).to(dev)
        with torch.no_grad(): m.sigma[0].bias += 0.5
        params = [p.detach().cpu().clone() for p in m.parameters()]
        st, plist = m._ensure_packed(), m._param_list()
        comp, _, _ = ops.render_rays_fused(st, plist, o.to(dev), d.to(dev), 2.0, 6.0, S, True, t_rand=u.to(dev))
        torch.mean((comp - tgt.to(dev)) ** 2).backward()
        out[pipe] = torch.cat([p.grad.cpu().reshape(-1).double() for p in plist])
    _, _, g32 = O.loss_and_grads(params, skip, L, o, d, tgt, 2.0, 6.0, S, u)
    _, _, g64 = O.loss_and_grads([p.double() for p in params], skip, L, o.double(), d.double(), tgt.double(), 2.0, 6.0, S, u.double())
    ref = torch.cat([x.reshape(-1) for x in g64]); out["cpu fp32"] = torch.cat([x.reshape(-1).double() for x in g32])
    small = (ref.abs() > 1e-9) & (ref.abs() < 1e-7)
    print(f"{in_dim}-{hidden}x{depth} skip {skip}: max|g| {float(ref.abs().max()):.2e}, {int(small.sum())} elements with 1e-9 < |g| < 1e-7")
    for k, v in out.items():
        e = (v - ref).abs()
        print(f"   {k:10s} L2 {float(e.norm() / ref.norm()):.2e}  worst/max {float(e.max() / ref.abs().max()):.2e}  abs err on the small elements: max {float(e[small].max()) if small.any() else 0:.2e} median {float(e[small].median()) if small.any() else 0:.2e}")
