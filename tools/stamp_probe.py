#!/usr/bin/env python3
"""Diagnostic: where does a wave of the fused forward kernel spend its cycles?  Builds nothing; expects
tiny-nerf-pytorch_amd/tnerf/libtnerf_hip_stamps.so (csrc/build_stamps.sh).  Stamp values never leave the debug buffer."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"))
from tnerf import ops, lib
import nerf
dev = torch.device("cuda:0")
dbg = C.CDLL(os.path.join(ROOT, "tiny-nerf-pytorch_amd", "tnerf", "libtnerf_hip_stamps.so"))
torch.manual_seed(0)
model = nerf.TinyNeRF(39, 256, 8, 4).to(dev)
with torch.no_grad(): model.sigma[0].bias += 0.5
st = model._ensure_packed()
R, S = 4096, 64
g = torch.Generator().manual_seed(1)
d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
o = (-4.0 * d).to(dev).contiguous(); d = d.to(dev).contiguous()
ztab = ops.depth_table(2.0, 6.0, S, dev)
comp = torch.empty(R, 3, device=dev)
stamps = torch.zeros(R * 32, dtype=torch.int64, device=dev)
for train in (False, True):
    plan = st.plan(R * S) if train else None
    for it in range(3):
        stamps.zero_()
        rc = dbg.tnerf_debug_render_stamps(C.byref(st.desc), C.c_void_p(st.packed.data_ptr()), C.c_void_p(o.data_ptr()), C.c_void_p(d.data_ptr()),
                                           C.c_int64(R), C.c_int32(S), C.c_void_p(ztab.data_ptr()), C.c_void_p(comp.data_ptr()),
                                           C.c_void_p(plan.stash.data_ptr() if train else None), C.c_int64(plan.Mp if train else 0),
                                           C.c_void_p(stamps.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(R, 32).astype(np.int64)
    names = ["ray start->tile2 encode done", "layer0", "layer1", "layer2", "layer3", "layer4(skip)", "layer5", "layer6", "layer7", None, None, "heads", "composite", "epilogue"]
    print(f"--- train={train}: median cycles over {R} waves (second 32-sample tile of each ray)")
    tot = np.median(s[:, 14] - s[:, 0])
    print(f"whole ray (2 tiles): {tot:.0f} cycles;  ideal MFMA: {2*7616*64}")
    seq = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 14]
    lab = ["[first tile + 2nd tile's encode]", "layer0", "layer1", "layer2", "layer3", "layer4(skip)", "layer5", "layer6", "layer7", "heads", "composite(2 tiles)", "final reduce+store"]
    for (a, b), n in zip(zip(seq[:-1], seq[1:]), lab):
        dlt = s[:, b] - s[:, a]
        print(f"  {n:34s} {np.median(dlt):9.0f}  (p10 {np.percentile(dlt,10):.0f}, p90 {np.percentile(dlt,90):.0f})")
