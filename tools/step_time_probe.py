"""Train-step time of arbitrary model shapes in both precisions (e.g. the reference's default 4x128 / L=10)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")]
import torch
from tnerf import ops, trainer
import nerf as nerf_mod

dev = torch.device("cuda:0")
H = W = 100; focal = 138.88887889922103
pose = torch.eye(4, device=dev); pose[2, 3] = 4.0
pixels = torch.rand(H * W, 3, device=dev)
for (L, hidden, depth, skip, R, S) in ((10, 128, 4, 2, 2048, 64), (10, 128, 4, 2, 4096, 64), (6, 256, 8, 4, 4096, 64), (10, 256, 8, 4, 4096, 128)):
    macs = (6 * L + 3) * hidden + (depth - 1) * hidden * hidden + ((6 * L + 3) * hidden if skip else 0) + 4 * hidden
    flop = 3 * 2 * macs * R * S
    for prec in ("fp32", "bf16"):
        torch.manual_seed(0)
        model = nerf_mod.TinyNeRF(6 * L + 3, hidden, depth, skip).to(dev)
        with torch.no_grad():
            model.sigma[0].bias += 0.5
        opt = trainer.FlatAdam(model, lr=5e-4)
        tr = trainer.FusedTrainer(model, opt, 2.0, 6.0, S, precision=prec)
        gen = torch.Generator(device=dev); gen.manual_seed(1)
        def step():
            inds = torch.randint(0, H * W, (R,), device=dev, generator=gen)
            u = torch.rand(R, S, device=dev, generator=gen)
            tr.step_camera(pose, H, W, focal, inds, pixels, t_rand=u)
        for _ in range(5):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        peak = 157.3e12 if prec == "fp32" else 2516e12
        print(f"L={L} {depth}x{hidden} skip {skip} R={R} S={S} {prec}: {dt * 1e3:.3f} ms/step  {R / dt / 1e3:.0f} k rays/s  {flop / dt / 1e12:.1f} TFLOP/s ({100 * flop / dt / peak:.0f} % of MFMA peak)", flush=True)
