"""A few hundred dataset steps at a small ray count (run under rocprofv3 --kernel-trace --stats to see where a 512-ray step goes).
   python tools/small_step_probe.py [rays] [steps]        TNERF_X3_UNITS=rays|tiles selects the work unit of the x3 chain kernels"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")]
import torch
from tnerf import trainer
from data import make_synthetic_scene
import nerf as nerf_mod

R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
scene = make_synthetic_scene(seed=0)
images = torch.from_numpy(scene["images"]).to(dev); poses = torch.from_numpy(scene["poses"]).to(dev); focal = float(scene["focal"])
torch.manual_seed(0)
model = nerf_mod.TinyNeRF(39, 256, 8, 4).to(dev)
with torch.no_grad():
    model.sigma[0].bias += 0.5
tr = trainer.DatasetTrainer(model, trainer.FlatAdam(model, lr=5e-4), images, poses, focal, R, 64, 2.0, 6.0, seed=1234)
for _ in range(20):
    tr.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n):
    tr.step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"{R} rays x 64, units={os.environ.get('TNERF_X3_UNITS', 'auto')}: {dt * 1e3:.4f} ms/step", flush=True)
