"""Drop-in for the reference's src/make_gif.py: load the checkpoint, render the 60-pose spiral around the
first camera, write outputs/novel_views.gif (imageio if present, else numbered PNG frames).
[reference src/make_gif.py:9-33]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from camera import spiral_poses                    # noqa: E402
from data import load_scene                        # noqa: E402
from encoding import PositionalEncoding            # noqa: E402
from nerf import TinyNeRF                          # noqa: E402
from train import render_one, write_png            # noqa: E402
from _hip import dist as _dist                     # noqa: E402


def main(ckpt_path="checkpoints/tinynerf_latest.pth", out_dir="outputs", precision="fp32", data_path="data/tiny_nerf_data.npz",
         n_frames=60, radius=0.3):
    """Arguments are additions (the reference hard-codes them, make_gif.py:11-23); precision="bf16" renders the frames
    with the bf16-MFMA kernel (BASELINE cfg 4), ~9x faster.  Returns the list of uint8 frames (rank 0; None elsewhere)."""
    device = torch.device("cuda", torch.cuda.current_device())
    d = load_scene(data_path)                        # a missing npz raises, as in the reference (make_gif.py:11)
    H, W, focal = d["images"].shape[1], d["images"].shape[2], float(d["focal"])
    poses = torch.from_numpy(d["poses"]).to(device)
    ckpt = torch.load(ckpt_path, map_location=device)
    in_dim = int(ckpt.get("in_dim", 63))            # the reference assumes L=10 (make_gif.py:17)
    encoder = PositionalEncoding(num_freqs=(in_dim - 3) // 6, include_input=True).to(device)
    model = TinyNeRF(in_dim=in_dim, **ckpt.get("cfg", dict(hidden=128, depth=4, skip_at=2))).to(device)   # make_gif.py:19
    model.load_state_dict(ckpt["model"])
    # Under torch.distributed (one process per GPU) the 60 poses are dealt round-robin to the ranks — whole frames
    # need no communication; single process: all frames here.
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    frames = []
    n_frames = int(n_frames)
    path = spiral_poses(poses[0], n_frames=n_frames, radius=float(radius))
    for k in _dist.deal_round_robin(n_frames, rank, world):
        pose = path[k]
        img = render_one(model, encoder, H, W, focal, pose, device, n_samples=64, near=2.0, far=6.0, precision=precision)
        frames.append((img.cpu().numpy() * 255).astype(np.uint8))
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, frames)
        frames = _dist.merge_round_robin(gathered, n_frames)
        if rank != 0:
            return None
    os.makedirs(out_dir, exist_ok=True)
    try:
        import imageio.v2 as imageio
        imageio.mimsave(os.path.join(out_dir, "novel_views.gif"), frames, fps=15, loop=0)      # make_gif.py:32
        print(f"[save] {out_dir}/novel_views.gif")
    except ImportError:
        for i, f in enumerate(frames):
            write_png(os.path.join(out_dir, f"novel_view_{i:03d}.png"), f)
        print(f"[save] {out_dir}/novel_view_000..{len(frames) - 1:03d}.png (imageio not installed: no GIF)")
    return frames


if __name__ == "__main__":
    main(*sys.argv[1:])
