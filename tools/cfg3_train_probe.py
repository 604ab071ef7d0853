"""BASELINE cfg 3 on one GPU's share: 400x400 scene (analytically re-rendered synthetic stand-in, 24 views), 128
samples/ray, 4096 rays per step, 8x256 L=6: train-step time, held-out PSNR after N steps, full-image render time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")]
import torch
from tnerf import trainer
from data import make_synthetic_scene
import nerf as nerf_mod, train as train_mod
from encoding import PositionalEncoding
from utils import mse2psnr

dev = torch.device("cuda:0")
STEPS = int(os.environ.get("STEPS", "1000"))
scene = make_synthetic_scene(n_images=24, H=400, W=400, focal=4 * 138.88887889922103, seed=0)
images = torch.from_numpy(scene["images"]).to(dev); poses = torch.from_numpy(scene["poses"]).to(dev); focal = float(scene["focal"])
N, H, W, _ = images.shape
pixels = images.view(N, H * W, 3)
enc = PositionalEncoding(6, True).to(dev)
for prec in ("fp32", "bf16"):
    torch.manual_seed(0)
    model = nerf_mod.TinyNeRF(39, 256, 8, 4).to(dev)
    with torch.no_grad():
        model.sigma[0].bias += 0.5
    opt = trainer.FlatAdam(model, lr=5e-4)
    tr = trainer.FusedTrainer(model, opt, 2.0, 6.0, 128, precision=prec)
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    def step(s):
        i = s % (N - 1)
        inds = torch.randint(0, H * W, (4096,), device=dev, generator=gen)
        u = torch.rand(4096, 128, device=dev, generator=gen)
        return tr.step_camera(poses[i], H, W, focal, inds, pixels[i], t_rand=u)
    for s in range(10):
        step(s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(10, 10 + STEPS):
        loss, _ = step(s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / STEPS
    t1 = time.perf_counter()
    img = train_mod.render_one(model, enc, H, W, focal, poses[N - 1], dev, n_samples=128, near=2.0, far=6.0, chunk=20000, precision=prec)
    torch.cuda.synchronize(); tr_ms = (time.perf_counter() - t1) * 1e3
    ps = float(mse2psnr(torch.mean((img - images[N - 1]) ** 2)))
    print(f"cfg3 {prec}: {dt * 1e3:.3f} ms/step = {4096 / dt / 1e3:.0f} k rays/s per GPU; after {10 + STEPS} steps minibatch {float(mse2psnr(loss)):.2f} dB, "
          f"held-out 400x400 view {ps:.2f} dB; full-image render (8 chunks of 20000 rays) {tr_ms:.1f} ms = {H * W / tr_ms / 1e3:.2f} M rays/s", flush=True)
