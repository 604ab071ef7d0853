#!/usr/bin/env python3
"""Diagnostic: TinyNeRF.forward (MLP-only kernel, hidden 128: <=256 registers/wave) at 2 workgroups/CU vs the same
kernel capped to 1 workgroup/CU by a dummy dynamic-LDS request (TNERF_LIB selects the build)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"))
import nerf
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = nerf.TinyNeRF(63, 128, 4, 2).to(dev)
x = torch.randn(262144, 63, device=dev)
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); m(x); b.record()
    torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
fl = 2 * (63 * 128 + 3 * 128 * 128 + 63 * 128 + 4 * 128) * 262144
print(os.environ.get("TNERF_LIB", "default"), f"mlp_fwd 4x128 on 262144 rows: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
