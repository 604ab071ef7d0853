#!/usr/bin/env python3
"""Diagnostic sweep: the x3 chain kernels against the fp32-MFMA kernels (same C ABI, same inputs) over odd shapes — ray / sample
counts that are not multiples of the tile sizes, every depth, skip positions, padded widths.  Prints the worst deviations."""
import ctypes as C, itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"))
import torch
from tnerf import ops, lib
import nerf
dev = torch.device("cuda:0")
cases = [(6, 256, 2, 0, 1, 1), (6, 256, 2, 1, 1, 33), (6, 256, 16, 8, 5, 64), (10, 128, 2, 0, 3, 31), (10, 128, 16, 5, 9, 65),
         (4, 200, 3, 2, 7, 129), (1, 17, 5, 3, 2, 96), (10, 256, 8, 4, 2, 1000), (6, 128, 8, 7, 1025, 16), (2, 64, 4, 1, 4097, 7)]
worst = 0.0
for (L, hidden, depth, skip, R, S) in cases:
    torch.manual_seed(L * 1000 + hidden + depth)
    m = nerf.TinyNeRF(6 * L + 3, hidden, depth, skip).to(dev)
    with torch.no_grad(): m.sigma[0].bias += 0.5
    st = m._ensure_packed(); x3 = st.repack_x3(("e", 0))
    g = torch.Generator().manual_seed(R + S)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev); o = (-4.0 * d + 0.1).contiguous()
    u = torch.rand(R, S, generator=g).to(dev); gc = (torch.randn(R, 3, generator=g) / (3 * R)).to(dev)
    plan = st.plan(R * S); ztab = ops.depth_table(2.0, 6.0, S, dev)
    sA, sB = torch.zeros_like(plan.stash), torch.zeros_like(plan.stash)
    cA, cB, cI = (torch.empty(R, 3, device=dev) for _ in range(3)); dep = torch.empty(R, 1, device=dev); acc = torch.empty(R, 1, device=dev)
    sp = torch.cuda.current_stream(dev).cuda_stream
    common = (o.data_ptr(), d.data_ptr(), R, S, ztab.data_ptr(), 1, u.data_ptr(), 0, 0, 1)
    lib.call("tnerf_train_fwd_fused", C.byref(st.desc), st.packed.data_ptr(), *common, cA.data_ptr(), sA.data_ptr(), plan.Mp, sp)
    lib.call("tnerf_train_fwd_fused_x3", C.byref(st.desc), x3.packed.data_ptr(), *common, cB.data_ptr(), sB.data_ptr(), plan.Mp, sp)
    lib.call("tnerf_render_fused_x3", C.byref(st.desc), x3.packed.data_ptr(), *common, cI.data_ptr(), dep.data_ptr(), acc.data_ptr(), sp)
    gA, gB, gC = (torch.zeros(st.n_params, device=dev) for _ in range(3))
    sC = sA.clone()                                                  # x3 dgrad on the fp32-MFMA forward's stash: no sign flips involved
    for (s_, g_, p3) in ((sA, gA, None), (sB, gB, x3.packed.data_ptr()), (sC, gC, x3.packed.data_ptr())):
        lib.call("tnerf_train_bwd_fused", C.byref(st.desc), st.packed.data_ptr(), *common, gc.data_ptr(), s_.data_ptr(), plan.Mp,
                 plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr(), plan.reduce.data_ptr(), g_.data_ptr(), p3, sp)
    torch.cuda.synchronize()
    e_c = float((cA - cB).abs().max()); e_i = float((cI - cB).abs().max())
    e_g = float((gA - gB).norm() / gA.norm().clamp_min(1e-30))
    e_d = float((gA - gC).norm() / gA.norm().clamp_min(1e-30))
    body = plan.stash.numel() - sA[:0].numel()
    flips = int((sA.view(torch.int32) != sB.view(torch.int32)).sum())
    fin = bool(torch.isfinite(cB).all() and torch.isfinite(gB).all())
    worst = max(worst, e_c, e_g)
    print(f"L={L:2d} {depth:2d}x{hidden:3d} skip {skip} R={R:5d} S={S:4d}: |comp x3 - fp32mfma| {e_c:.1e}  |inference - training comp| {e_i:.1e}  grad rel L2 {e_g:.1e} (same stash: {e_d:.1e})  finite {fin}", flush=True)
    assert fin and e_c <= 5e-6 and e_i <= 1e-6 and e_d <= 2e-6 and e_g <= 1e-3
print("worst", worst)
