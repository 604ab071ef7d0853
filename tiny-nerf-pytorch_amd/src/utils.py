"""Drop-in for the reference's src/utils.py.  PSNR of a scalar is not hot-path work: it stays a torch op."""
import torch


def mse2psnr(mse: torch.Tensor) -> torch.Tensor:
    """PSNR in dB = -10 log10(max(mse, 1e-10)).                      [reference src/utils.py:14-15]"""
    return torch.log10(torch.clamp_min(mse, 1e-10)) * -10.0
