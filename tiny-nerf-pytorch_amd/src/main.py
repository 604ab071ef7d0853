"""Drop-in for the reference's src/main.py: the "does my GPU + data work" smoke render of an UNTRAINED
model for pose 0, timed, written to outputs/preview.png.            [reference src/main.py:36-65]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from data import load_scene                       # noqa: E402
from encoding import PositionalEncoding            # noqa: E402
from nerf import TinyNeRF                          # noqa: E402
from train import render_one, write_png            # noqa: E402


def main():
    os.makedirs("outputs", exist_ok=True)
    torch.manual_seed(0)
    if not torch.cuda.is_available():
        raise RuntimeError("main.py (HIP): no ROCm GPU visible; this package has no CPU path")
    device = torch.device("cuda")
    print(f"[device] {device} torch={torch.__version__}")
    d = load_scene("data/tiny_nerf_data.npz")
    images, poses, focal = d["images"], torch.from_numpy(d["poses"]), float(d["focal"])
    N, H, W, _ = images.shape
    print(f"[data] N={N}, H={H}, W={W}, focal={focal:.2f}")
    encoder = PositionalEncoding(num_freqs=10, include_input=True).to(device)
    model = TinyNeRF(in_dim=encoder.out_dim, hidden=128, depth=4, skip_at=2).to(device)
    t0 = time.time()
    img = render_one(model, encoder, H, W, focal, poses[0], device, n_samples=64, near=2.0, far=6.0, chunk=8192)
    torch.cuda.synchronize()
    print(f"[render] one frame {H}x{W} in {time.time() - t0:.3f}s (includes first-call setup)")
    write_png("outputs/preview.png", (img.cpu().numpy() * 255).astype(np.uint8))
    print("[save] outputs/preview.png")


if __name__ == "__main__":
    main()
