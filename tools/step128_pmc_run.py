"""PMC target: 40 train steps of the reference-default model (4x128, L=10, 2048 rays x 64) and 3 renders of a 400x400 image, nothing else.
   rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d <dir> -- python3 tools/step128_pmc_run.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")]
import torch
from tnerf import trainer
from data import make_synthetic_scene
import nerf as nerf_mod, train as train_mod, encoding as enc_mod
dev = torch.device("cuda:0"); torch.cuda.set_stream(torch.cuda.Stream(dev))
scene = make_synthetic_scene(seed=0)
images = torch.from_numpy(scene["images"]).to(dev); poses = torch.from_numpy(scene["poses"]).to(dev); focal = float(scene["focal"])
torch.manual_seed(0)
m = nerf_mod.TinyNeRF(63, 128, 4, 2).to(dev)
with torch.no_grad(): m.sigma[0].bias += 0.5
tr = trainer.DatasetTrainer(m, trainer.FlatAdam(m, lr=5e-4), images, poses, focal, 2048, 64, 2.0, 6.0, seed=1234, graph=False)
for _ in range(40): tr.step()
enc = enc_mod.PositionalEncoding(10).to(dev)
with torch.no_grad():
    for _ in range(3): train_mod.render_one(m, enc, 400, 400, focal * 4, poses[0], dev, 64, 2.0, 6.0, chunk=32768)
torch.cuda.synchronize()
