"""bf16 training gradients at a batch large enough for several passes per workgroup; table and camera ray sources."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"), os.path.join(ROOT, "tests")]
import torch
from conftest import load_golden, golden_params
from oracle import tnerf_oracle as O
import nerf
from tnerf import ops, lib as L, trainer as T

dev = torch.device("cuda:0")
cfg, _ = golden_params("4x128")
g = torch.Generator().manual_seed(3)
params = O.mlp_init(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"], g)
params[2 * cfg["depth"] + 1] = params[2 * cfg["depth"] + 1] + 0.5
m = nerf.TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"]).to(dev)
with torch.no_grad():
    for p, v in zip(m.parameters(), params):
        p.copy_(v.to(dev))
st = m._ensure_packed()
H = W = 64; focal = 80.0
pose = load_golden("render_4x128")["pose"]
ro, rd = O.pinhole_rays(H, W, focal, pose)
for R in (2500, 5000):
    S = 64
    inds = torch.randint(0, H * W, (R,), generator=torch.Generator().manual_seed(4))
    o, d = ro[inds].contiguous(), rd[inds].contiguous()
    pixels = torch.rand(H * W, 3, generator=torch.Generator().manual_seed(1))
    tgt = pixels[inds].contiguous()
    t = torch.rand(R, S, generator=torch.Generator().manual_seed(2))
    l16, _, g16 = O.loss_and_grads_bf16(params, cfg["skip_at"], cfg["L"], o, d, tgt, 2., 6., S, t)
    w16 = torch.cat([x.reshape(-1) for x in g16])
    b = st.repack_bf16(); bp = b.train_plan(R, S)
    ztab = ops.depth_table(2.0, 6.0, S, dev)
    comp = torch.empty(R, 3, device=dev); gws = torch.empty(R, 4, device=dev); loss = torch.zeros(1, device=dev)
    od, dd, tg, td = o.to(dev), d.to(dev), tgt.to(dev), t.to(dev)
    s_ = torch.cuda.current_stream(dev).cuda_stream
    st.grad.zero_()
    L.call("tnerf_train_step_fused_bf16", C.byref(st.desc), b.packed.data_ptr(), od.data_ptr(), dd.data_ptr(), tg.data_ptr(), R, S,
           ztab.data_ptr(), 1, td.data_ptr(), 0, 0, 1, float(3 * R), comp.data_ptr(), gws.data_ptr(), gws.numel(), loss.data_ptr(),
           bp.stash.data_ptr(), bp.jobs.data_ptr(), bp.n_jobs, bp.slabs.data_ptr(), bp.reduce.data_ptr(), st.grad.data_ptr(), s_)
    torch.cuda.synchronize()
    flat = st.grad.cpu().clone()
    print(f"tables R={R}: loss {float(loss):.6f} want {float(l16):.6f} |g| {float(flat.norm()):.4e} want {float(w16.norm()):.4e} rel {float((flat - w16).norm() / w16.norm()):.2e}", flush=True)
    cam, keep = ops.camera_struct(pose.to(dev), H, W, focal, inds.to(dev), 0)
    pix = pixels.to(dev)
    st.grad.zero_()
    L.call("tnerf_train_step_fused_cam_bf16", C.byref(st.desc), b.packed.data_ptr(), C.byref(cam), pix.data_ptr(), R, S,
           ztab.data_ptr(), 1, td.data_ptr(), 0, 0, 1, float(3 * R), comp.data_ptr(), gws.data_ptr(), gws.numel(), loss.data_ptr(),
           bp.stash.data_ptr(), bp.jobs.data_ptr(), bp.n_jobs, bp.slabs.data_ptr(), bp.reduce.data_ptr(), st.grad.data_ptr(), s_)
    torch.cuda.synchronize()
    flat = st.grad.cpu().clone()
    print(f"camera R={R}: loss {float(loss):.6f} want {float(l16):.6f} |g| {float(flat.norm()):.4e} want {float(w16.norm()):.4e} rel {float((flat - w16).norm() / w16.norm()):.2e}", flush=True)
    # through the trainer
    opt = T.FlatAdam(m, lr=5e-4)
    tr = T.FusedTrainer(m, opt, 2.0, 6.0, S, precision="bf16")
    before = st.flat.clone()
    lo, _ = tr.step_camera(pose.to(dev), H, W, focal, inds.to(dev), pix, t_rand=td)
    torch.cuda.synchronize()
    print(f"trainer: loss {float(lo):.6f} |grad| {float(st.grad.norm()):.4e} |dW| {float((st.flat - before).norm()):.4e}", flush=True)
    with torch.no_grad():
        st.flat.copy_(before)
