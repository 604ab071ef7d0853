"""Drop-in for the reference's src/camera.py: novel-view pose path (host-side, 60 4x4 matrices)."""
import math

import torch


def spiral_poses(c2w_ref: torch.Tensor, n_frames: int = 60, radius: float = 0.3):
    """n_frames poses c2w_ref @ T(radius cos t, radius sin t, 0), t in linspace(0, 2pi).
    [reference src/camera.py:4-12]"""
    t = torch.linspace(0, 2 * math.pi, n_frames, device=c2w_ref.device)
    shift = torch.eye(4, device=c2w_ref.device, dtype=c2w_ref.dtype).repeat(n_frames, 1, 1)
    shift[:, 0, 3] = (radius * torch.cos(t)).to(c2w_ref.dtype)
    shift[:, 1, 3] = (radius * torch.sin(t)).to(c2w_ref.dtype)
    return torch.matmul(c2w_ref.unsqueeze(0), shift)
