#!/usr/bin/env python3
"""Diagnostic: the stash's magnitude-bound words after an x3 forward + backward against the true row maxima, and the per-tensor
gradient error."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_golden, golden_params
from tnerf import ops, lib
import nerf
dev = torch.device("cuda:0")
tag = "8x256"
cfg, params = golden_params(tag); g = load_golden(f"mlp_{tag}")
m = nerf.TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"]).to(dev)
with torch.no_grad():
    for p, v in zip(m.parameters(), params): p.copy_(v.to(dev))
st = m._ensure_packed(); x3 = st.repack_x3(1)
x = g["x"].to(dev); M = x.shape[0]; plan = st.plan(M)
H, depth, NE = 256, cfg["depth"], 20
rows = 2 * NE + depth * H + 4 + depth * H + 4
s = torch.zeros_like(plan.stash); rgb = torch.empty(M, 3, device=dev); sig = torch.empty(M, 1, device=dev); grads = torch.zeros(st.n_params, device=dev)
sp = torch.cuda.current_stream(dev).cuda_stream
gr, gs = g["g_rgb"].to(dev).contiguous(), g["g_sigma"].to(dev).contiguous()
lib.call("tnerf_mlp_fwd_x3", C.byref(st.desc), x3.packed.data_ptr(), x.data_ptr(), M, rgb.data_ptr(), sig.data_ptr(), s.data_ptr(), plan.Mp, sp)
lib.call("tnerf_mlp_bwd_x3", C.byref(st.desc), x3.packed.data_ptr(), M, gr.data_ptr(), gs.data_ptr(), s.data_ptr(), plan.Mp, plan.jobs.data_ptr(), plan.n_jobs,
         plan.slabs.data_ptr(), plan.reduce.data_ptr(), grads.data_ptr(), sp)
torch.cuda.synchronize()
boff = rows * (plan.Mp + 32) + depth * (plan.Mp + 32) * (H // 32)
b = s[boff: boff + 64].cpu()
print("stash floats", s.numel(), "bound offset", boff, "rows", rows)
A = s[: (plan.Mp // 32 + 1) * rows * 32].view(-1, rows, 32)[: M // 32].cpu()
print("enc  bound %.4g true max %.4g" % (float(b[16]), float(A[:, :2 * NE].abs().max())))
r0 = 2 * NE
for l in range(depth):
    print("H[%d] bound %.4g true max %.4g" % (l, float(b[l]), float(A[:, r0:r0 + H].abs().max()))); r0 += H
r0 += 4
for l in range(depth):
    print("dZ[%d] bound %.4g true max %.4g" % (l, float(b[17 + l]), float(A[:, r0:r0 + H].abs().max()))); r0 += H
print("dZh bound %.4g true max %.4g" % (float(b[33]), float(A[:, r0:r0 + 4].abs().max())))
jobs = plan.jobs.cpu().view(-1, 16)
print("job bound indices (class, a, b):", sorted(set((int(j[10]), int(j[12]), int(j[13])) for j in jobs)))
gref = torch.cat([g[f"g{i:02d}"].reshape(-1) for i in range(len(params))])
gh = grads.cpu()
o = 0
for i, p in enumerate(params):
    n = p.numel(); a, r = gh[o:o + n], gref[o:o + n]; o += n
    print(f"tensor {i}: max|d|/max|g| = {float((a - r).abs().max() / r.abs().max()):.2e}  finite={bool(torch.isfinite(a).all())}")
o = 0
for i, p in enumerate(params[:6]):
    n = p.numel(); a, r = gh[o:o + n], gref[o:o + n]; o += n
    if p.dim() == 2:
        k = int(r.abs().argmax()); print(f"tensor {i}: largest ref elem {float(r[k]):.4e} hip {float(a[k]):.4e} ratio {float(a[k] / r[k]):.4f};  corr {float((a * r).sum() / (a.norm() * r.norm())):.4f}  norm ratio {float(a.norm() / r.norm()):.4f}")
