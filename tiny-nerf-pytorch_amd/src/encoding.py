"""Drop-in for the reference's src/encoding.py (tnerf_posenc_fwd, csrc/stage_kernels.hip)."""
import torch
import torch.nn as nn

from _hip import ops


class PositionalEncoding(nn.Module):
    """gamma(x) = [x, sin(2^k x), cos(2^k x)]_{k<L}: frequency-major, sin then cos, xyz innermost.
    [reference src/encoding.py:4-33]"""

    def __init__(self, num_freqs: int = 10, include_input: bool = True):
        super().__init__()
        self.num_freqs = num_freqs
        self.include_input = include_input
        self.register_buffer("freq_bands", 2.0 ** torch.arange(num_freqs).float())

    @property
    def out_dim(self) -> int:
        return 6 * self.num_freqs + (3 if self.include_input else 0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape[-1] == 3, "PositionalEncoding expects (..., 3)"
        return ops.posenc(x, self.num_freqs, self.include_input)
