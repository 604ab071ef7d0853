#!/usr/bin/env python3
"""Diagnostic: cycles between the stage boundaries of wave 0's first tile (8x256 inference and training forward).
Needs: tools/build_variant.sh stagestamps mlpx3 -DTN_STAMPS -DTN_STAGE_STAMPS        python tools/x3_stage_probe.py [variant]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"))
os.environ["TNERF_LIB"] = os.path.join(ROOT, "tiny-nerf-pytorch_amd", "tnerf", f"libtnerf_variant_{sys.argv[1] if len(sys.argv) > 1 else 'stagestamps'}.so")
from tnerf import ops, lib
import nerf
dev = torch.device("cuda:0")
dbg = lib.load()
dbg.tnerf_debug_renderx3_stamps.restype = C.c_int
L, hidden, depth, skip, R, S = 6, 256, 8, 4, 4096, 64
torch.manual_seed(0)
model = nerf.TinyNeRF(6 * L + 3, hidden, depth, skip, matrix_pipe="x3").to(dev)
with torch.no_grad(): model.sigma[0].bias += 0.5
st = model._ensure_packed(); x3 = st.repack_x3(1)
g = torch.Generator().manual_seed(1)
d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
o = (-4.0 * d).to(dev).contiguous(); d = d.to(dev).contiguous()
ztab = ops.depth_table(2.0, 6.0, S, dev)
comp = torch.empty(R, 3, device=dev)
n_wave = 256 * 4
for train in (False, True):
    plan = st.plan(R * S) if train else None
    stamps = torch.zeros(n_wave * 8 + 64 + 512, dtype=torch.int64, device=dev)
    for it in range(20):
        stamps.zero_()
        rc = dbg.tnerf_debug_renderx3_stamps(C.byref(st.desc), C.c_void_p(x3.packed.data_ptr()), C.c_void_p(o.data_ptr()), C.c_void_p(d.data_ptr()),
                                             C.c_int64(R), C.c_int32(S), C.c_void_p(ztab.data_ptr()), C.c_void_p(comp.data_ptr()),
                                             C.c_void_p(plan.stash.data_ptr() if train else None), C.c_int64(plan.Mp if train else 0),
                                             C.c_void_p(stamps.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    sm = stamps.cpu().numpy()[n_wave * 8 + 64:]
    sm = sm[sm > 0]
    dlt = np.diff(sm)
    # stages of one tile: layer 0: 2 + 2; hidden layer: 8 + 8 (+ 2 + 2 for the skip layer); heads: 2
    per_tile = 4 + 7 * 16 + 4 + 2
    print(f"train={train}: {len(sm)} boundaries; cycles between consecutive stage boundaries, first tile ({per_tile} stages), 8 per hidden half-pass:")
    row = dlt[:per_tile]
    i = 0
    def take(n, label):
        global i
        seg = row[i:i + n]; i += n
        print(f"   {label:24s} " + " ".join(f"{int(v):5d}" for v in seg) + f"   sum {int(seg.sum())}")
    take(2, "layer 0 pass A"); take(2, "layer 0 pass B")
    for l in range(1, depth):
        take(8, f"layer {l} pass A")
        if l == skip: take(2, f"layer {l} pass A skip")
        take(8, f"layer {l} pass B")
        if l == skip: take(2, f"layer {l} pass B skip")
    take(2, "heads")
