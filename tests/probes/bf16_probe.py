"""Checker script (uses the CPU oracle, hence under tests/; run by hand on the GPU box, not collected by pytest).
GPU probe of the bf16 render kernel: parity against the CPU restatement and timing against the fp32 kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"), os.path.join(ROOT, "tests")]
import torch
from conftest import load_golden, golden_params
from oracle import tnerf_oracle as O
import nerf
from tnerf import ops

dev = torch.device("cuda:0")
for tag in ("4x128", "8x256"):
    cfg, params = golden_params(tag)
    m = nerf.TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"]).to(dev)
    with torch.no_grad():
        for p, v in zip(m.parameters(), params):
            p.copy_(v.to(dev))
    st = m._ensure_packed()
    g = load_golden(f"render_{tag}")
    ro, rd = O.pinhole_rays(g["H"], g["W"], g["focal"], g["pose"])
    for (R, S) in ((8, 64), (1000, 64), (333, 32), (200, 128), (64, 48), (17, 100)):
        o, d = ro[:R].contiguous(), rd[:R].contiguous()
        want16 = O.render_rays_bf16(params, cfg["skip_at"], cfg["L"], o, d, 2.0, 6.0, S)
        want32 = O.render_rays(params, cfg["skip_at"], cfg["L"], o, d, 2.0, 6.0, S)
        got = ops.render_rays_fused_bf16(st, o.to(dev), d.to(dev), 2.0, 6.0, S)
        torch.cuda.synchronize()
        e16 = float((got[0].cpu() - want16[0]).abs().max()); e32 = float((got[0].cpu() - want32[0]).abs().max())
        eo = float((want16[0] - want32[0]).abs().max())
        print(f"{tag} R={R} S={S}: |hip16-cpu16|={e16:.2e} |hip16-cpu32|={e32:.2e} |cpu16-cpu32|={eo:.2e} "
              f"depth {float((got[1].cpu()-want16[1]).abs().max()):.2e} acc {float((got[2].cpu()-want16[2]).abs().max()):.2e}", flush=True)

# timing at BASELINE render sizes (8x256, L=6)
cfg, params = golden_params("8x256")
m = nerf.TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"]).to(dev)
st = m._ensure_packed()
pose = load_golden("render_8x256")["pose"].to(dev)
for (H, S) in ((100, 64), (400, 128), (800, 256)):
    n = H * H
    focal = 138.88887889922103 * H / 100
    for name, fn in (("fp32", ops.render_camera_fused), ("bf16", ops.render_camera_fused_bf16)):
        for _ in range(2):
            fn(st, pose, H, H, focal, 0, n, 2.0, 6.0, S)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            out = fn(st, pose, H, H, focal, 0, n, 2.0, 6.0, S)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        fl = n * S * 959488
        print(f"{H}x{H} S={S} {name}: {dt*1e3:.3f} ms  {n/dt/1e6:.2f} Mrays/s  {fl/dt/1e12:.1f} TFLOP/s", flush=True)
