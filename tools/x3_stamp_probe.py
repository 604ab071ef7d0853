#!/usr/bin/env python3
"""Diagnostic: shader clock and walk/epilogue split of the x3 forward kernel.   python tools/x3_stamp_probe.py
Needs: tools/build_variant.sh x3stamps mlpx3 -DTN_STAMPS"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"))
os.environ["TNERF_X3_UNITS"] = "rays"            # the stamps live in the ray kernels
os.environ["TNERF_LIB"] = os.path.join(ROOT, "tiny-nerf-pytorch_amd", "tnerf", f"libtnerf_variant_{sys.argv[1] if len(sys.argv) > 1 else 'x3stamps'}.so")
from tnerf import ops, lib
import nerf
dev = torch.device("cuda:0")
dbg = lib.load()
dbg.tnerf_debug_renderx3_stamps.restype = C.c_int
for (L, hidden, depth, skip, R, S) in ((6, 256, 8, 4, 4096, 64), (10, 128, 4, 2, 2048, 64)):
    torch.manual_seed(0)
    model = nerf.TinyNeRF(6 * L + 3, hidden, depth, skip).to(dev)
    with torch.no_grad(): model.sigma[0].bias += 0.5
    st = model._ensure_packed(); x3 = st.repack_x3(1)
    g = torch.Generator().manual_seed(1)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    o = (-4.0 * d).to(dev).contiguous(); d = d.to(dev).contiguous()
    ztab = ops.depth_table(2.0, 6.0, S, dev)
    comp = torch.empty(R, 3, device=dev)
    nw_inf = (int(sys.argv[2]) if len(sys.argv) > 2 else 8) if hidden <= 128 else 4      # waves per workgroup of the inference kernel (TxCfg::RENDER; argv[2]: a -DTX_NW128 build)
    n_wave = 256 * 8
    stamps = torch.zeros(n_wave * 8 + 64, dtype=torch.int64, device=dev)
    H = 128 if hidden <= 128 else 256
    in_pad = 16 * ((6 * L + 3 + 15) // 16 + (0 if (6 * L + 3) % 16 else 0))
    for train in (False, True):
        nw = 4 if train else nw_inf                              # (the ray training kernel: TxCfg::RAY)
        stamps.zero_()
        plan = st.plan(R * S) if train else None
        for it in range(60):
            rc = dbg.tnerf_debug_renderx3_stamps(C.byref(st.desc), C.c_void_p(x3.packed.data_ptr()), C.c_void_p(o.data_ptr()), C.c_void_p(d.data_ptr()),
                                                 C.c_int64(R), C.c_int32(S), C.c_void_p(ztab.data_ptr()), C.c_void_p(comp.data_ptr()),
                                                 C.c_void_p(plan.stash.data_ptr() if train else None), C.c_int64(plan.Mp if train else 0),
                                                 C.c_void_p(stamps.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
        torch.cuda.synchronize()
        marks = stamps.cpu().numpy()[256 * nw * 8:]
        s = stamps.cpu().numpy()[: 256 * nw * 8].reshape(256 * nw, 8).astype(np.float64)
        s = s[s[:, 0] > 0]
        cyc, rt, walk, epi = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
        tiles = R * S / 32 / len(s)
        n_mfma = tiles * 3 * (H // 32) * ((depth - 1) * (H // 16) + (2 if skip else 1) * 4) + tiles * 3 * (H // 16)
        print(f"L={L} {depth}x{hidden} R={R} S={S} train={train}: waves {len(s)}, clock {np.median(cyc / rt) * 0.1:.3f} GHz, wave lifetime {np.median(rt) / 100:.1f} us")
        print(f"   cycles/wave {np.median(cyc):.0f}: layer walks {np.median(walk):.0f} ({np.median(walk / cyc) * 100:.1f}%), epilogues {np.median(epi):.0f} ({np.median(epi / cyc) * 100:.1f}%), "
              f"rest {np.median(cyc - walk - epi):.0f};  MFMA issue floor 32 x {n_mfma:.0f} = {32 * n_mfma:.0f} ({32 * n_mfma / np.median(cyc) * 100:.1f}% of the wave, {32 * n_mfma / np.median(walk) * 100:.1f}% of the walks)")
        t_in, t_p, t_out = s[:, 4], s[:, 5], s[:, 6]
        if t_in.max() > 0:
            z = t_in.min(); q = lambda a: " ".join(f"{v / 100:.1f}" for v in np.percentile(a, [0, 5, 50, 95, 100]))
            print(f"   absolute times in us from the first wave's entry (min p5 p50 p95 max): entry {q(t_in - z)} | prologue done {q(t_p - z)} | exit {q(t_out - z)} | lifetime {q(t_out - t_in)}")
            xcd = (np.arange(len(s)) // nw) % 8
            print("   per XCD (workgroup % 8) median exit us: " + " ".join(f"{np.median((t_out - z)[xcd == x]) / 100:.1f}" for x in range(8)) + "; median clock GHz: " + " ".join(f"{np.median((cyc / rt)[xcd == x]) * 0.1:.3f}" for x in range(8)))
        nm = int(marks[63])
        if nm > 1:
            print("   first tile of wave 0, cycles between pass marks: " + " ".join(str(int(marks[i + 1] - marks[i])) for i in range(nm - 1)))
