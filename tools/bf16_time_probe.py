"""Per-kernel times of the bf16 training step at BASELINE cfg 2 shapes (TNERF_LIB selects a diagnostic build)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"), os.path.join(ROOT, "tests")]
import torch
import rays as rays_mod
import nerf
from tnerf import ops, lib as L

dev = torch.device("cuda:0")
m = nerf.TinyNeRF(39, 256, 8, 4).to(dev)
with torch.no_grad():
    m.sigma[0].bias += 0.5
st = m._ensure_packed()
R, S = int(os.environ.get("R", "4096")), 64
pose = torch.eye(4); pose[2, 3] = 4.0
ro, rd = rays_mod.get_rays(100, 100, 138.88887889922103, pose.to(dev))
idx = torch.randint(0, 10000, (R,), generator=torch.Generator().manual_seed(0)).to(dev)
o, d = ro[idx].contiguous(), rd[idx].contiguous()
tgt = torch.rand(R, 3, device=dev); t = torch.rand(R, S, device=dev)
b = st.repack_bf16(); bp = b.train_plan(R, S)
ztab = ops.depth_table(2.0, 6.0, S, dev)
comp = torch.empty(R, 3, device=dev); gws = torch.rand(R, 3, device=dev) * 1e-3; sws = torch.empty(4 * R, device=dev); loss = torch.zeros(1, device=dev)
dep = torch.empty(R, 1, device=dev); acc = torch.empty(R, 1, device=dev)
s_ = torch.cuda.current_stream(dev).cuda_stream
calls = {
    "render": lambda: L.call("tnerf_render_fused_bf16", C.byref(st.desc), b.packed.data_ptr(), o.data_ptr(), d.data_ptr(), R, S, ztab.data_ptr(), 1, t.data_ptr(), 0, 0, 1, comp.data_ptr(), dep.data_ptr(), acc.data_ptr(), s_),
    "fwd": lambda: L.call("tnerf_train_fwd_fused_bf16", C.byref(st.desc), b.packed.data_ptr(), o.data_ptr(), d.data_ptr(), R, S, ztab.data_ptr(), 1, t.data_ptr(), 0, 0, 1, comp.data_ptr(), bp.stash.data_ptr(), s_),
    "dgrad": lambda: L.call("tnerf_train_dgrad_fused_bf16", C.byref(st.desc), b.packed.data_ptr(), o.data_ptr(), d.data_ptr(), R, S, ztab.data_ptr(), 1, t.data_ptr(), 0, 0, 1, gws.data_ptr(), bp.stash.data_ptr(), s_),
    "wgrad": lambda: L.call("tnerf_wgrad_bf16", C.byref(st.desc), bp.stash.data_ptr(), bp.n_tiles, bp.jobs.data_ptr(), bp.n_jobs, bp.slabs.data_ptr(), s_),
    "step": lambda: L.call("tnerf_train_step_fused_bf16", C.byref(st.desc), b.packed.data_ptr(), o.data_ptr(), d.data_ptr(), tgt.data_ptr(), R, S, ztab.data_ptr(), 1, t.data_ptr(), 0, 0, 1, float(3 * R), comp.data_ptr(), sws.data_ptr(), sws.numel(), loss.data_ptr(), bp.stash.data_ptr(), bp.jobs.data_ptr(), bp.n_jobs, bp.slabs.data_ptr(), bp.reduce.data_ptr(), st.grad.data_ptr(), s_),
}
out = []
for name, fn in calls.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    out.append(f"{name} {e0.elapsed_time(e1) / 20:.4f}")
print(os.environ.get("TNERF_LIB", "default"), " | ".join(out), "ms", flush=True)

# per-workgroup stamps of k_wgrad16 (a -DTN_STAMPS build of mlp16_bwd.hip): is the job split balanced?
import numpy as np
calls["wgrad"](); torch.cuda.synchronize()
jobs = bp.jobs.cpu().numpy().reshape(-1, 16)
dt = (jobs[:, 14].astype(np.int64) & 0xffffffff) | (jobs[:, 15].astype(np.int64) << 32)
if dt.max() > 0:
    print(f"   k_wgrad16: longest workgroup {dt.max()} cycles, median {np.median(dt):.0f}")
    for c in sorted(set(jobs[:, 10])):
        m = jobs[:, 10] == c
        print(f"   class {c}: {m.sum():3d} WGs, tiles {jobs[m][0][4]}x{jobs[m][0][5]} sample tiles/WG {jobs[m][:,8].min()}-{jobs[m][:,8].max()}  cycles median {np.median(dt[m]):.0f} max {dt[m].max()}  per tile {np.median(dt[m] / jobs[m][:,8]):.0f}")
