#!/usr/bin/env python3
"""Diagnostic: time the weight-gradient kernel (both matrix pipes) of every library given on the command line
(TNERF_LIB-style paths; diagnostic variants from tools/build_variant.sh), per-workgroup cycles by job class when the
library was built with -DTN_STAMPS.   python tools/wgrad_x3_probe.py lib1.so lib2.so ...   (one subprocess per library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        print(f"=== {lib}", flush=True)
        subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, TNERF_LIB=os.path.abspath(lib)))
    sys.exit(0)
SHAPE = tuple(int(v) for v in os.environ.get("TNERF_PROBE_SHAPE", "6,256,8,4,4096,64").split(","))      # L, hidden, depth, skip, rays, samples
import ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"))
from tnerf import ops, trainer, lib
import nerf
dev = torch.device("cuda:0")
torch.manual_seed(0)
Lf, HID, DEPTH, SKIP, R, S = SHAPE
model = nerf.TinyNeRF(6 * Lf + 3, HID, DEPTH, SKIP).to(dev)
with torch.no_grad(): model.sigma[0].bias += 0.5
opt = trainer.FlatAdam(model, lr=5e-4); tr = trainer.FusedTrainer(model, opt, 2.0, 6.0, S)
g = torch.Generator().manual_seed(1)
d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
o = (-4.0 * d).to(dev); d = d.to(dev); tgt = torch.rand(R, 3, generator=g).to(dev); u = torch.rand(R, S, generator=g).to(dev)
for _ in range(3):
    tr.step(o, d, tgt, t_rand=u)
torch.cuda.synchronize()
st = model.hip_state(); plan = st.plan(R * S)
sp = torch.cuda.current_stream(dev).cuda_stream
for name, flags in (("x3", 0), ("fp32-mfma", lib.FLAG_FP32_MFMA)):
    dsc = lib.MlpDesc(6 * Lf + 3, HID, DEPTH, SKIP, flags)
    fn = lambda: lib.call("tnerf_wgrad", C.byref(dsc), plan.stash.data_ptr(), plan.Mp, R * S, plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr(), sp)
    fn(); torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    print(f"{name:10s} {ms:.3f} ms")
    jobs = plan.jobs.cpu().numpy().reshape(-1, 16)
    dt = (jobs[:, 14].astype(np.int64) & 0xffffffff) | (jobs[:, 15].astype(np.int64) << 32)
    if dt.max() > 0:
        print(f"   longest workgroup {dt.max()} cycles, median {np.median(dt):.0f}")
        for c in sorted(set(jobs[:, 10])):
            m = jobs[:, 10] == c
            print(f"   class {c}: {m.sum():3d} WGs, tiles {jobs[m][0][4]}x{jobs[m][0][5]} blocks/WG {jobs[m][:,8].min()}-{jobs[m][:,8].max()}  cycles median {np.median(dt[m]):.0f} max {dt[m].max()}  per block {np.median(dt[m] / jobs[m][:,8]):.0f}")
