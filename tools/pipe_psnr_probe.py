"""The same training run (BASELINE cfg 2: 8x256, 4096 rays x 64, torch-generator draws) on the x3 chain kernels, on the
fp32-MFMA kernels and in bf16 mode: held-out PSNR along the way.   STEPS=8000 python tools/pipe_psnr_probe.py
(one subprocess per pipe: TNERF_FP32_PIPE is read when a model is built)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for pipe, prec in (("", "fp32"), ("mfma32", "fp32"), ("", "bf16")):
        subprocess.run([sys.executable, __file__, prec], env=dict(os.environ, TNERF_FP32_PIPE=pipe), check=True)
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")]
import torch
from tnerf import trainer
from data import make_synthetic_scene
import nerf as nerf_mod, train as train_mod
from encoding import PositionalEncoding
from utils import mse2psnr
prec = sys.argv[1]
name = prec if prec == "bf16" else ("fp32 on the fp32-MFMA kernels" if os.environ.get("TNERF_FP32_PIPE") else "fp32 on the x3 kernels (fp16, 3 products)")
dev = torch.device("cuda:0")
STEPS = int(os.environ.get("STEPS", "8000"))
scene = make_synthetic_scene(seed=0)
images = torch.from_numpy(scene["images"]).to(dev); poses = torch.from_numpy(scene["poses"]).to(dev); focal = float(scene["focal"])
N, H, W, _ = images.shape
pixels = images.view(N, H * W, 3)
enc = PositionalEncoding(6, True).to(dev)
torch.manual_seed(0)
model = nerf_mod.TinyNeRF(39, 256, 8, 4).to(dev)
with torch.no_grad():
    model.sigma[0].bias += 0.5
opt = trainer.FlatAdam(model, lr=5e-4)
tr = trainer.FusedTrainer(model, opt, 2.0, 6.0, 64, precision=prec)
gen = torch.Generator(device=dev); gen.manual_seed(1234)
held = list(range(N - 8, N))                               # 8 held-out views
torch.cuda.synchronize(); t0 = time.perf_counter(); t_train = 0.0
out = []
for s in range(STEPS):
    i = s % (N - 8)
    inds = torch.randint(0, H * W, (4096,), device=dev, generator=gen)
    u = torch.rand(4096, 64, device=dev, generator=gen)
    loss, _ = tr.step_camera(poses[i], H, W, focal, inds, pixels[i], t_rand=u)
    if (s + 1) in (500, 1000, 2000, 4000, 6000, STEPS):
        torch.cuda.synchronize()
        ps = []
        for v in held:
            img = train_mod.render_one(model, enc, H, W, focal, poses[v], dev, n_samples=64, near=2.0, far=6.0, precision=prec)
            ps.append(float(mse2psnr(torch.mean((img - images[v]) ** 2))))
        out.append(f"{s + 1}: {sum(ps) / len(ps):.3f}")
print(f"{name:32s} held-out PSNR (8 views, dB) after " + "  ".join(out) + f"   [{(time.perf_counter() - t0):.1f} s wall incl. evaluation]", flush=True)
