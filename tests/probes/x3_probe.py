import os, sys, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import golden_params
from oracle import tnerf_oracle as O
from tnerf import ops
import nerf
dev = torch.device("cuda:0")
for tag in ("8x256", "4x128"):
    cfg, params = golden_params(tag)
    m = nerf.TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"]).to(dev)
    with torch.no_grad():
        for p, v in zip(m.parameters(), params): p.copy_(v.to(dev))
    st = m._ensure_packed()
    g = torch.Generator().manual_seed(1)
    for R, S in ((37, 50), (512, 64), (130, 128)):
        d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
        o = -4.0 * d + 0.3 * torch.randn(R, 3, generator=g); u = torch.rand(R, S, generator=g)
        co, do_, ao, _ = O.render_rays(params, cfg["skip_at"], cfg["L"], o, d, 2.0, 6.0, S, u)
        c3, d3, a3 = ops.render_rays_fused_x3(st, o.to(dev), d.to(dev), 2.0, 6.0, S, randomized=True, t_rand=u.to(dev))
        with torch.no_grad():
            c32, d32, a32 = ops.render_rays_fused(st, m._param_list(), o.to(dev), d.to(dev), 2.0, 6.0, S, True, t_rand=u.to(dev))
        torch.cuda.synchronize()
        print(tag, R, S, "x3 vs oracle", float((c3.cpu() - co).abs().max()), "fp32 kernel vs oracle", float((c32.cpu() - co).abs().max()),
              "x3 vs fp32 kernel", float((c3 - c32).abs().max()), "acc", float((a3.cpu() - ao).abs().max()), "depth", float((d3.cpu() - do_).abs().max()))
# timing at cfg 2 size
torch.manual_seed(0)
m = nerf.TinyNeRF(39, 256, 8, 4).to(dev)
with torch.no_grad(): m.sigma[0].bias += 0.5
st = m._ensure_packed()
R, S = 4096, 64
g = torch.Generator().manual_seed(1)
d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev); o = (-4.0 * d).contiguous()
def ev(fn, reps=10):
    fn(); torch.cuda.synchronize()
    es = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in es:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in es]))
with torch.no_grad():
    t32 = ev(lambda: ops.render_rays_fused(st, m._param_list(), o, d, 2.0, 6.0, S, False))
t3 = ev(lambda: ops.render_rays_fused_x3(st, o, d, 2.0, 6.0, S))
t16 = ev(lambda: ops.render_rays_fused_bf16(st, o, d, 2.0, 6.0, S))
print(f"inference 4096x64 8x256: fp32-mfma {t32:.3f} ms, x3 {t3:.3f} ms, bf16 {t16:.3f} ms")
