"""Drop-in for the reference's src/nerf.py: the same nn.Module surface (constructor, attributes,
submodule names, state_dict keys, forward signature) in front of the MFMA kernels
(tnerf_mlp_fwd / tnerf_mlp_bwd, csrc/mlp_fwd.hip, mlp_bwd.hip, wgrad.hip)."""
import torch
import torch.nn as nn

from _hip import ops


class TinyNeRF(nn.Module):
    """ReLU MLP over encoded xyz with one skip concat; heads: rgb = sigmoid(Linear(h,3)),
    sigma = relu(Linear(h,1)).                                        [reference src/nerf.py:4-41]

    The nn.Linear submodules are created in the reference's order (layers, sigma, rgb), so
    torch.manual_seed(s) gives the same initial weights and the checkpoint keys are identical.  On the
    first forward the parameters become views into one flat fp32 buffer (what the fused optimizer and
    the gradient all-reduce operate on).
    """

    def __init__(self, in_dim: int, hidden: int = 128, depth: int = 4, skip_at: int = 2, *, matrix_pipe=None):
        """matrix_pipe (keyword only, not in the reference): how the fp32 products are formed —
        "x3"  (default): three partial products of two-piece fp16 operands on the fp16 matrix pipe (22-23 significant bits per
                         operand, fp32 accumulation; measured at or below the reference's own fp32 error);
        "fp32_mfma"    : v_mfma_f32_32x32x2_f32, plain fp32 fma chains (TNERF_FLAG_FP32_MFMA), ~2.5x slower.
        None reads the environment variable TNERF_FP32_PIPE (mfma32 -> "fp32_mfma") and otherwise means "x3"."""
        super().__init__()
        if matrix_pipe not in (None, "x3", "fp32_mfma"):
            raise ValueError(f"matrix_pipe must be None, 'x3' or 'fp32_mfma', got {matrix_pipe!r}")
        self.matrix_pipe = matrix_pipe
        self.in_dim, self.hidden, self.depth, self.skip_at = in_dim, hidden, depth, skip_at
        self.layers = nn.ModuleList()
        width = in_dim
        for i in range(depth):
            self.layers.append(nn.Linear(width, hidden))
            width = hidden + in_dim if i == skip_at - 1 else hidden
        self.sigma = nn.Sequential(nn.Linear(hidden, 1), nn.ReLU(inplace=True))
        self.rgb = nn.Sequential(nn.Linear(hidden, 3), nn.Sigmoid())
        self._hip = None

    # ------------------------------------------------------------------ HIP state
    def _param_list(self):
        return list(self.parameters())

    def hip_state(self) -> "ops.ModelState":
        params = self._param_list()
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("TinyNeRF (HIP): parameters are on the CPU; move the model to a ROCm GPU "
                               "(model.to('cuda')).  There is no CPU fallback.")
        st = self._hip
        if st is None or st.device != dev:
            skip = self.skip_at if 1 <= self.skip_at <= self.depth - 1 else 0
            if self.skip_at == self.depth:
                raise RuntimeError("TinyNeRF: skip_at == depth feeds hidden+in_dim features to the heads")
            flags = None if self.matrix_pipe is None else (ops.FLAG_FP32_MFMA if self.matrix_pipe == "fp32_mfma" else 0)
            st = self._hip = ops.ModelState(self.in_dim, self.hidden, self.depth, skip, dev, flags)
        if not st.owns(params):          # first use, or some parameter was rebound since: re-adopt all of them
            st.adopt(params)
        return st

    def _ensure_packed(self):
        st = self.hip_state()
        st.repack(tuple(p._version for p in self._param_list()))
        return st

    # ------------------------------------------------------------------ reference surface
    def _chain_kernels_cover(self) -> bool:
        """The register-resident MFMA kernels take hidden <= 256 and in_dim <= 64 (L <= 10); anything else the constructor accepts
        (reference src/nerf.py:10) runs layer by layer (ops.mlp_forward_generic: one library SGEMM per layer)."""
        return self.hidden <= 256 and self.in_dim <= 64

    def forward(self, x):
        """x: (N, in_dim) encoded coordinates -> rgb (N,3) in [0,1], sigma (N,1) >= 0.  Differentiable in the parameters and in x."""
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        if not self._chain_kernels_cover() or (x2.requires_grad and torch.is_grad_enabled()):
            if self.skip_at == self.depth:
                raise RuntimeError("TinyNeRF: skip_at == depth feeds hidden+in_dim features to the heads")
            skip = self.skip_at if 1 <= self.skip_at <= self.depth - 1 else 0
            rgb, sigma = ops.mlp_forward_generic(self.in_dim, self.hidden, self.depth, skip, x2, self._param_list())
        else:
            st = self._ensure_packed()
            rgb, sigma = ops.mlp_forward(st, x2, self._param_list())
        return rgb.reshape(*lead, 3), sigma.reshape(*lead, 1)
