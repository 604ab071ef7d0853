#!/usr/bin/env python3
"""Diagnostic: where does a wave of the fused forward kernel spend its cycles?   python tools/stamp_probe.py [L hidden depth skip R]
Needs a -DTN_STAMPS build of mlp_fwd.hip: tools/build_variant.sh fwdstamps mlp_fwd -DTN_STAMPS  (loaded from
tiny-nerf-pytorch_amd/tnerf/libtnerf_variant_fwdstamps.so).  Stamp values never leave the debug buffer."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"))
LIB = os.path.join(ROOT, "tiny-nerf-pytorch_amd", "tnerf", "libtnerf_variant_fwdstamps.so")
os.environ["TNERF_LIB"] = LIB
from tnerf import ops, lib
import nerf
L, hidden, depth, skip, R = (int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (6, 256, 8, 4, 4096)))
dev = torch.device("cuda:0")
dbg = lib.load()
dbg.tnerf_debug_render_stamps.restype = C.c_int
torch.manual_seed(0)
model = nerf.TinyNeRF(6 * L + 3, hidden, depth, skip).to(dev)
with torch.no_grad(): model.sigma[0].bias += 0.5
st = model._ensure_packed()
S = 64
g = torch.Generator().manual_seed(1)
d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
o = (-4.0 * d).to(dev).contiguous(); d = d.to(dev).contiguous()
ztab = ops.depth_table(2.0, 6.0, S, dev)
comp = torch.empty(R, 3, device=dev)
stamps = torch.zeros(R * 32, dtype=torch.int64, device=dev)
in_dim = 6 * L + 3
macs = in_dim * hidden + (depth - 1) * hidden * hidden + (in_dim * hidden if skip else 0) + 4 * hidden
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
    print(f"--- train={train}: median cycles over {R} waves (second 32-sample tile of each ray); L={L} {depth}x{hidden} skip {skip}")
    tot = np.median(s[:, 14] - s[:, 0])
    print(f"whole ray (2 tiles): {tot:.0f} cycles;  MFMA pipe time of one wave: {2 * macs * 32 * 2 // 4096 * 64}")
    rt = (s[:, 21] - s[:, 20]).astype(np.float64)           # 100 MHz ticks
    ghz = (s[:, 14] - s[:, 0]) / np.maximum(rt, 1) * 0.1
    print(f"shader clock while the waves ran (s_memtime / s_memrealtime): median {np.median(ghz):.3f} GHz; wave lifetime median {np.median(rt) / 100:.1f} us, "
          f"first start -> last end {(s[:, 21].max() - s[:, 20].min()) / 100:.1f} us")
    seq = [0] + list(range(1, depth + 2)) + [12, 13, 14]
    lab = ["[first tile + 2nd tile's encode]"] + [f"layer{i}" + ("(skip)" if skip and i == skip else "") for i in range(depth)] + ["heads", "composite(2 tiles)", "final reduce+store"]
    for (a, b), n in zip(zip(seq[:-1], seq[1:]), lab):
        dlt = s[:, b] - s[:, a]
        print(f"  {n:34s} {np.median(dlt):9.0f}  (p10 {np.percentile(dlt,10):.0f}, p90 {np.percentile(dlt,90):.0f})")
