#!/usr/bin/env python3
"""Diagnostic: launches each x3 chain kernel and the x3 weight-gradient kernel a few times (8x256, 4096 x 64) so that a rocprofv3 --pmc pass sees them alone:
   rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d out -- python3 tools/x3_pmc_run.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"))
import torch
from tnerf import ops, lib
import nerf
dev = torch.device("cuda:0")
L, hidden, depth, skip, R, S = 6, 256, 8, 4, 4096, 64
torch.manual_seed(0)
m = nerf.TinyNeRF(6 * L + 3, hidden, depth, skip).to(dev)
with torch.no_grad(): m.sigma[0].bias += 0.5
st = m._ensure_packed(); x3 = st.repack_x3(1); plan = st.plan(R * S)
g = torch.Generator().manual_seed(1)
d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev); o = (-4.0 * d).contiguous(); u = torch.rand(R, S, generator=g).to(dev)
ztab = ops.depth_table(2.0, 6.0, S, dev); comp = torch.empty(R, 3, device=dev); dep = torch.empty(R, 1, device=dev); acc = torch.empty(R, 1, device=dev)
gws = torch.full((R, 3), 1e-4, device=dev); sp = torch.cuda.current_stream(dev).cuda_stream
cx = (C.byref(st.desc), x3.packed.data_ptr(), o.data_ptr(), d.data_ptr(), R, S, ztab.data_ptr(), 1, u.data_ptr(), 0, 0, 1)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    lib.call("tnerf_render_fused_x3", *cx, comp.data_ptr(), dep.data_ptr(), acc.data_ptr(), sp)
    lib.call("tnerf_train_fwd_fused_x3", *cx, comp.data_ptr(), plan.stash.data_ptr(), plan.Mp, sp)
    lib.call("tnerf_train_dgrad_fused_x3", *cx, gws.data_ptr(), plan.stash.data_ptr(), plan.Mp, sp)
    lib.call("tnerf_wgrad", C.byref(st.desc), plan.stash.data_ptr(), plan.Mp, R * S, plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr(), sp)
torch.cuda.synchronize()
