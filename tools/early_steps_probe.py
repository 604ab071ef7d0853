"""Why are the first steps of a run slower?  20 timed steps after 5 warm-up steps (the driver's protocol) on (a) a freshly initialised
network and (b) the same network after 300 training steps, with a NEW trainer (new graph capture, new buffers) in both cases: if (b) is as
slow as (a) the cost is start-up (graph upload, clocks), if (b) is fast it is the data (power drawn by the matrix pipe on untrained
activations -> clock).   python tools/early_steps_probe.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")]
import torch
from tnerf import trainer
from data import make_synthetic_scene
import nerf as nerf_mod
dev = torch.device("cuda:0"); torch.cuda.set_stream(torch.cuda.Stream(dev))
scene = make_synthetic_scene(seed=0)
images = torch.from_numpy(scene["images"]).to(dev); poses = torch.from_numpy(scene["poses"]).to(dev); focal = float(scene["focal"])
torch.manual_seed(0)
m = nerf_mod.TinyNeRF(39, 256, 8, 4).to(dev)
with torch.no_grad(): m.sigma[0].bias += 0.5
opt = trainer.FlatAdam(m, lr=5e-4)

def protocol(tag, start_step=0):
    tr = trainer.DatasetTrainer(m, opt, images, poses, focal, 4096, 64, 2.0, 6.0, seed=1234, start_step=start_step)
    for _ in range(5): tr.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): tr.step()
    torch.cuda.synchronize(); print(f"{tag}: {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms/step", flush=True)
    return tr

tr = protocol("fresh network, steps 6-25")
for _ in range(300): tr.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): tr.step()
torch.cuda.synchronize(); print(f"same trainer, steps 326-345: {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms/step", flush=True)
n = int(opt._t)
del tr
protocol("trained network, NEW trainer, its steps 6-25", start_step=n)
