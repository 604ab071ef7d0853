#!/usr/bin/env python3
"""Diagnostic: HIP-event times of the three x3 chain kernels for every library given (TNERF_LIB-style paths; variants from
tools/build_variant.sh).   python tools/x3_time_probe.py lib1.so lib2.so ...   (one subprocess per library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for lib in sys.argv[1:]:
        print(f"=== {lib}", flush=True)
        subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, TNERF_LIB=os.path.abspath(lib)))
    sys.exit(0)
import ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"))
from tnerf import ops, lib
import nerf
dev = torch.device("cuda:0")
for (L, hidden, depth, skip, R, S) in ((6, 256, 8, 4, 4096, 64), (10, 128, 4, 2, 2048, 64)):
    torch.manual_seed(0)
    m = nerf.TinyNeRF(6 * L + 3, hidden, depth, skip).to(dev)
    with torch.no_grad(): m.sigma[0].bias += 0.5
    st = m._ensure_packed(); x3 = st.repack_x3(1); plan = st.plan(R * S)
    g = torch.Generator().manual_seed(1)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev); o = (-4.0 * d).contiguous(); u = torch.rand(R, S, generator=g).to(dev)
    ztab = ops.depth_table(2.0, 6.0, S, dev); comp = torch.empty(R, 3, device=dev); dep = torch.empty(R, 1, device=dev); acc = torch.empty(R, 1, device=dev)
    gws = torch.full((R, 3), 1e-4, device=dev); sp = torch.cuda.current_stream(dev).cuda_stream
    cx = (C.byref(st.desc), x3.packed.data_ptr(), o.data_ptr(), d.data_ptr(), R, S, ztab.data_ptr(), 1, u.data_ptr(), 0, 0, 1)
    calls = {"render": lambda: lib.call("tnerf_render_fused_x3", *cx, comp.data_ptr(), dep.data_ptr(), acc.data_ptr(), sp),
             "train_fwd": lambda: lib.call("tnerf_train_fwd_fused_x3", *cx, comp.data_ptr(), plan.stash.data_ptr(), plan.Mp, sp),
             "dgrad": lambda: lib.call("tnerf_train_dgrad_fused_x3", *cx, gws.data_ptr(), plan.stash.data_ptr(), plan.Mp, sp)}
    out = []
    for n, fn in calls.items():
        for _ in range(30): fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        out.append(f"{n} {np.mean([a.elapsed_time(b) for a, b in ev]) * 1e3:.1f} us")
    print(f"L={L} {depth}x{hidden} R={R} S={S}: " + "  ".join(out), flush=True)
