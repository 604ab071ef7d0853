"""fp32 vs bf16 training on the synthetic scene (BASELINE cfg 2 / cfg 4): step time and held-out PSNR after N steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")]
import torch
from tnerf import ops, trainer
from data import make_synthetic_scene
import nerf as nerf_mod, train as train_mod
from encoding import PositionalEncoding
from utils import mse2psnr

dev = torch.device("cuda:0")
STEPS = int(os.environ.get("STEPS", "2000"))
scene = make_synthetic_scene(seed=0)
images = torch.from_numpy(scene["images"]).to(dev); poses = torch.from_numpy(scene["poses"]).to(dev); focal = float(scene["focal"])
N, H, W, _ = images.shape
pixels = images.view(N, H * W, 3)
enc = PositionalEncoding(6, True).to(dev)
res = {}
for prec, seed in ((("fp32", 1234), ("bf16", 1234)) if os.environ.get("ONE_SEED") else (("fp32", 1234), ("bf16", 1234), ("fp32", 99), ("bf16", 99))):
    torch.manual_seed(0)
    model = nerf_mod.TinyNeRF(39, 256, 8, 4).to(dev)
    with torch.no_grad():
        model.sigma[0].bias += 0.5
    opt = trainer.FlatAdam(model, lr=5e-4)
    tr = trainer.FusedTrainer(model, opt, 2.0, 6.0, 64, precision=prec)
    gen = torch.Generator(device=dev); gen.manual_seed(seed)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(STEPS):
        i = s % (N - 1)                                    # hold out the last view
        inds = torch.randint(0, H * W, (4096,), device=dev, generator=gen)
        u = torch.rand(4096, 64, device=dev, generator=gen)
        loss, _ = tr.step_camera(poses[i], H, W, focal, inds, pixels[i], t_rand=u)
        if (s + 1) in (100, 500, 1000, 2000, 4000, STEPS):
            torch.cuda.synchronize()
            img = train_mod.render_one(model, enc, H, W, focal, poses[N - 1], dev, n_samples=64, near=2.0, far=6.0)
            ps = float(mse2psnr(torch.mean((img - images[N - 1]) ** 2)))
            print(f"{prec} seed {seed} step {s + 1}: minibatch psnr {float(mse2psnr(loss)):.3f} held-out {ps:.3f} dB  ({(time.perf_counter() - t0) / (s + 1) * 1e3:.3f} ms/step incl. eval)", flush=True)
    res[(prec, seed)] = ps
print({f"{k[0]}/{k[1]}": round(v, 3) for k, v in res.items()})
