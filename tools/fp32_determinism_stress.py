"""Stress: the fp32 train step launched many times on identical inputs must give bit-identical gradients every time
(LDS-DMA staged k_wgrad: a missed wait / early slot reuse would show up as an occasional mismatch)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")]
import torch
import nerf, rays as rays_mod
from tnerf import ops, lib as L

dev = torch.device("cuda:0")
N = int(os.environ.get("N", "200"))
for (L_, hidden, depth, skip, R, S) in ((6, 256, 8, 4, 4096, 64), (10, 128, 4, 2, 2048, 64), (6, 256, 8, 4, 777, 100), (6, 128, 3, 0, 1001, 33)):
    torch.manual_seed(0)
    m = nerf.TinyNeRF(6 * L_ + 3, hidden, depth, skip).to(dev)
    with torch.no_grad():
        m.sigma[0].bias += 0.5
    st = m._ensure_packed()
    pose = torch.eye(4, device=dev); pose[2, 3] = 4.0
    ro, rd = rays_mod.get_rays(100, 100, 138.88887889922103, pose)
    idx = torch.randint(0, 10000, (R,), device=dev)
    o, d = ro[idx].contiguous(), rd[idx].contiguous()
    tgt = torch.rand(R, 3, device=dev); t = torch.rand(R, S, device=dev)
    plan = st.plan(R * S)
    ztab = ops.depth_table(2.0, 6.0, S, dev)
    comp = torch.empty(R, 3, device=dev); gws = torch.empty(R, 4, device=dev); loss = torch.zeros(1, device=dev)
    s_ = torch.cuda.current_stream(dev).cuda_stream
    ref = None; bad = 0
    for i in range(N):
        L.call("tnerf_train_step_fused", C.byref(st.desc), st.packed.data_ptr(), o.data_ptr(), d.data_ptr(), tgt.data_ptr(), R, S,
               ztab.data_ptr(), 1, t.data_ptr(), 0, 0, 1, float(3 * R), comp.data_ptr(), gws.data_ptr(), gws.numel(), loss.data_ptr(), plan.stash.data_ptr(), plan.Mp,
               plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr(), plan.reduce.data_ptr(), st.grad.data_ptr(),
               st.repack_x3(1).packed.data_ptr() if st.x3_capable else None, s_)       # forward on the x3 chain kernel
        g = st.grad.clone()
        if ref is None:
            ref = g
        elif not torch.equal(g, ref):
            bad += 1
    torch.cuda.synchronize()
    print(f"{depth}x{hidden} L={L_} R={R} S={S}: {N} launches, {bad} differ from the first; |grad| {float(ref.norm()):.4e} finite {bool(torch.isfinite(ref).all())}", flush=True)
