"""rocprofv3 target: 60 fused train steps of the reference's default model (4x128, L=10, 2048 rays)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src")]
import torch
from tnerf import trainer
import nerf as nerf_mod
dev = torch.device("cuda:0")
H = W = 100; focal = 138.88887889922103
pose = torch.eye(4, device=dev); pose[2, 3] = 4.0
pixels = torch.rand(H * W, 3, device=dev)
torch.manual_seed(0)
model = nerf_mod.TinyNeRF(63, 128, 4, 2).to(dev)
with torch.no_grad():
    model.sigma[0].bias += 0.5
opt = trainer.FlatAdam(model, lr=5e-4)
tr = trainer.FusedTrainer(model, opt, 2.0, 6.0, 64, precision=os.environ.get("PREC", "fp32"))
gen = torch.Generator(device=dev); gen.manual_seed(1)
for _ in range(60):
    inds = torch.randint(0, H * W, (2048,), device=dev, generator=gen)
    u = torch.rand(2048, 64, device=dev, generator=gen)
    tr.step_camera(pose, H, W, focal, inds, pixels, t_rand=u)
torch.cuda.synchronize()
