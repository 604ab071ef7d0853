#!/usr/bin/env python3
"""Diagnostic: per-layer activations written by the x3 chain forward and by the fp32-MFMA forward against an fp64 evaluation of
the same network on the same encoded inputs (the stash holds the encoder output, so the comparison isolates the MLP)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import golden_params
from tnerf import ops, lib
import nerf
dev = torch.device("cuda:0")
for tag in ("4x128", "8x256"):
    cfg, params = golden_params(tag)
    m = nerf.TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"]).to(dev)
    with torch.no_grad():
        for p, v in zip(m.parameters(), params): p.copy_(v.to(dev))
    st = m._ensure_packed(); x3 = st.repack_x3(1)
    M = 4096
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, cfg["in_dim"], generator=g)
    plan = st.plan(M)
    H, depth, NE = (128 if cfg["hidden"] <= 128 else 256), cfg["depth"], (32 if cfg["L"] == 10 else 20)
    rows = 2 * NE + depth * H + 4 + depth * H + 4
    # fp64 layers, and dL/dz_l for a random upstream gradient
    P = [p.double() for p in params]
    h = x.double(); acts = []; zs = []
    for l in range(depth):
        z = h @ P[2 * l].T + P[2 * l + 1]; z.requires_grad_(True) if not z.requires_grad else None
        zs.append(z); h = torch.relu(z); acts.append(h.detach())
        if cfg["skip_at"] > 0 and l == cfg["skip_at"] - 1: h = torch.cat([h, x.double()], -1)
    sig64 = torch.relu(h @ P[2 * depth].T + P[2 * depth + 1]); rgb64 = torch.sigmoid(h @ P[2 * depth + 2].T + P[2 * depth + 3])
    g_rgb = torch.randn(M, 3, generator=g) * 0.1; g_sig = torch.randn(M, 1, generator=g) * 0.1
    P0 = P[0].clone().requires_grad_(True)            # (zs[0] is a leaf-like tensor only through P0; rebuild with grads enabled)
    h = x.double(); zs = []
    for l in range(depth):
        W = P0 if l == 0 else P[2 * l]
        z = h @ W.T + P[2 * l + 1]; z.retain_grad(); zs.append(z); h = torch.relu(z)
        if cfg["skip_at"] > 0 and l == cfg["skip_at"] - 1: h = torch.cat([h, x.double()], -1)
    sig64 = torch.relu(h @ P[2 * depth].T + P[2 * depth + 1]); rgb64 = torch.sigmoid(h @ P[2 * depth + 2].T + P[2 * depth + 3])
    ((rgb64 * g_rgb.double()).sum() + (sig64 * g_sig.double()).sum()).backward()
    dz64 = [z.grad for z in zs]
    for name, fn, packed in (("x3", "tnerf_mlp_fwd_x3", x3.packed), ("fp32-MFMA", "tnerf_mlp_fwd", st.packed)):
        s = torch.zeros_like(plan.stash); rgb = torch.empty(M, 3, device=dev); sig = torch.empty(M, 1, device=dev)
        lib.call(fn, C.byref(st.desc), packed.data_ptr(), x.to(dev).data_ptr(), M, rgb.data_ptr(), sig.data_ptr(), s.data_ptr(), plan.Mp, torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
        A = s[: (plan.Mp // 32 + 1) * rows * 32].view(-1, rows, 32)[: M // 32].cpu().double()
        out = []
        r0 = 2 * NE
        for l in range(depth):
            a = A[:, r0:r0 + H].permute(0, 2, 1).reshape(M, H)[:, : cfg["hidden"]]
            ref = acts[l]
            out.append(f"{float((a - ref).norm() / ref.norm()):.1e}")
            r0 += H
        print(f"{tag} {name:10s} rel L2 error of H[l] vs fp64: " + " ".join(out), flush=True)
        # backward on this forward's stash
        jobs = plan
        bw = "tnerf_mlp_bwd_x3" if name == "x3" else "tnerf_mlp_bwd"
        lib.call(bw, C.byref(st.desc), packed.data_ptr(), M, g_rgb.to(dev).data_ptr(), g_sig.to(dev).data_ptr(), s.data_ptr(), plan.Mp,
                 plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr(), plan.reduce.data_ptr(), st.grad.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
        A = s[: (plan.Mp // 32 + 1) * rows * 32].view(-1, rows, 32)[: M // 32].cpu().double()
        out = []
        r0 = 2 * NE + depth * H + 4
        for l in range(depth):
            a = A[:, r0:r0 + H].permute(0, 2, 1).reshape(M, H)[:, : cfg["hidden"]]
            out.append(f"{float((a - dz64[l]).norm() / dz64[l].norm()):.1e}")
            r0 += H
        print(f"{tag} {name:10s} rel L2 error of dZ[l] vs fp64: " + " ".join(out), flush=True)
