#!/usr/bin/env python3
"""Diagnostic: error of the MLP parameter gradients against an fp64 evaluation for every combination of forward and dgrad
kernel (x3 = split-bf16 chain, m32 = fp32 MFMA); the weight-gradient kernel is the same in all of them."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_golden, golden_params
from oracle import tnerf_oracle as O
from tnerf import ops, lib
import nerf
dev = torch.device("cuda:0")
for tag in ("4x128", "8x256"):
    cfg, params = golden_params(tag); g = load_golden(f"mlp_{tag}")
    m = nerf.TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"]).to(dev)
    with torch.no_grad():
        for p, v in zip(m.parameters(), params): p.copy_(v.to(dev))
    st = m._ensure_packed(); x3 = st.repack_x3(1)
    x = g["x"].to(dev); M = x.shape[0]; plan = st.plan(M)
    gr, gs = g["g_rgb"].to(dev).contiguous(), g["g_sigma"].to(dev).contiguous()
    leaves = [p.double().requires_grad_(True) for p in params]
    r64, s64 = O.mlp_forward(leaves, g["x"].double(), cfg["skip_at"])
    g64 = torch.cat([t.reshape(-1) for t in torch.autograd.grad((r64 * g["g_rgb"].double()).sum() + (s64 * g["g_sigma"].double()).sum(), leaves)])
    gref = torch.cat([g[f"g{i:02d}"].reshape(-1).double() for i in range(len(params))])
    sp = torch.cuda.current_stream(dev).cuda_stream
    print(f"{tag} M={M}: CPU fp32 (fixture) rel L2 error vs fp64 {float((gref - g64).norm() / g64.norm()):.2e}")
    for fw in ("x3", "m32"):
        for bw in ("x3", "m32"):
            s = torch.zeros_like(plan.stash); rgb = torch.empty(M, 3, device=dev); sig = torch.empty(M, 1, device=dev); grads = torch.zeros(st.n_params, device=dev)
            lib.call("tnerf_mlp_fwd_x3" if fw == "x3" else "tnerf_mlp_fwd", C.byref(st.desc), (x3 if fw == "x3" else st).packed.data_ptr(), x.data_ptr(), M,
                     rgb.data_ptr(), sig.data_ptr(), s.data_ptr(), plan.Mp, sp)
            lib.call("tnerf_mlp_bwd_x3" if bw == "x3" else "tnerf_mlp_bwd", C.byref(st.desc), (x3 if bw == "x3" else st).packed.data_ptr(), M, gr.data_ptr(), gs.data_ptr(),
                     s.data_ptr(), plan.Mp, plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr(), plan.reduce.data_ptr(), grads.data_ptr(), sp)
            torch.cuda.synchronize()
            gh = grads.cpu().double()
            print(f"   forward {fw:3s} dgrad {bw:3s}: rel L2 error vs fp64 {float((gh - g64).norm() / g64.norm()):.2e}   worst element / max {float((gh - g64).abs().max() / g64.abs().max()):.2e}", flush=True)
