"""Checker script (uses the CPU oracle, hence under tests/; run by hand on the GPU box, not collected by pytest).
GPU probe of the bf16 training step: gradients against the CPU restatement, per-kernel timing, PSNR after N steps."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"), os.path.join(ROOT, "tests")]
import torch
from conftest import load_golden, golden_params
from oracle import tnerf_oracle as O
import nerf
from tnerf import ops, lib as L, trainer as T

dev = torch.device("cuda:0")


def lively(cfg, seed=3):
    g = torch.Generator().manual_seed(seed)
    params = O.mlp_init(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"], g)
    params[2 * cfg["depth"] + 1] = params[2 * cfg["depth"] + 1] + 0.5
    return params


def grads_bf16(model, st, o, d, tgt, t, S, denom=None):
    R = o.shape[0]
    b = st.repack_bf16()
    bp = b.train_plan(R, S)
    ztab = ops.depth_table(2.0, 6.0, S, dev)
    comp = torch.empty(R, 3, device=dev); gws = torch.empty(R, 4, device=dev); loss = torch.zeros(1, device=dev)
    st.grad.zero_()
    L.call("tnerf_train_step_fused_bf16", C.byref(st.desc), b.packed.data_ptr(), o.data_ptr(), d.data_ptr(), tgt.data_ptr(), R, S,
           ztab.data_ptr(), 1, t.data_ptr(), 0, 0, 1, float(denom or 3 * R), comp.data_ptr(), gws.data_ptr(), gws.numel(), loss.data_ptr(),
           bp.stash.data_ptr(), bp.jobs.data_ptr(), bp.n_jobs, bp.slabs.data_ptr(), bp.reduce.data_ptr(), st.grad.data_ptr(),
           torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    return float(loss), comp.cpu(), st.grad.cpu().clone()


for tag in ("4x128", "8x256"):
    cfg, trained = golden_params(tag)
    g = load_golden(f"render_{tag}")
    ro, rd = O.pinhole_rays(int(g["H"]), int(g["W"]), float(g["focal"]), g["pose"])
    for pname, params in (("trained", trained), ("lively", lively(cfg))):
        m = nerf.TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"]).to(dev)
        with torch.no_grad():
            for p, v in zip(m.parameters(), params):
                p.copy_(v.to(dev))
        st = m._ensure_packed()
        for (R, S) in ((64, 64), (100, 48), (37, 100)):
            idx = torch.arange(0, ro.shape[0], max(1, ro.shape[0] // R))[:R]
            o, d = ro[idx].contiguous(), rd[idx].contiguous()
            tgt = torch.rand(R, 3, generator=torch.Generator().manual_seed(1))
            t = torch.rand(R, S, generator=torch.Generator().manual_seed(2))
            l16, _, g16 = O.loss_and_grads_bf16(params, cfg["skip_at"], cfg["L"], o, d, tgt, 2., 6., S, t)
            l32, _, g32 = O.loss_and_grads(params, cfg["skip_at"], cfg["L"], o, d, tgt, 2., 6., S, t)
            loss, comp, flat = grads_bf16(m, st, o.to(dev), d.to(dev), tgt.to(dev), t.to(dev), S)
            w16 = torch.cat([x.reshape(-1) for x in g16]); w32 = torch.cat([x.reshape(-1) for x in g32])
            worst = 0.0; off = 0
            for x in g16:
                n = x.numel(); e = float((flat[off:off + n] - x.reshape(-1)).norm() / (x.norm() + 1e-20)); worst = max(worst, e); off += n
            if os.environ.get("PROBE_DETAIL") and R == 64:
                off = 0
                for i, x in enumerate(g16):
                    n = x.numel(); gg = flat[off:off + n].reshape(x.shape); off += n
                    print(f"    p{i:02d} {tuple(x.shape)}: rel {float((gg - x).norm() / (x.norm() + 1e-20)):.2e} |g| {float(gg.norm()):.3e} |want| {float(x.norm()):.3e}", flush=True)
            print(f"{tag} {pname} R={R} S={S}: loss hip {loss:.6f} cpu16 {float(l16):.6f} cpu32 {float(l32):.6f} | "
                  f"|g-g16|/|g16| {float((flat - w16).norm() / w16.norm()):.2e} (worst tensor {worst:.2e})  |g-g32|/|g32| {float((flat - w32).norm() / w32.norm()):.2e}",
                  flush=True)

if os.environ.get("PROBE_DETAIL"): sys.exit(0)
# ---- timing at BASELINE cfg 2 shapes
cfg = dict(in_dim=39, hidden=256, depth=8, skip_at=4, L=6)
params = lively(cfg, 0)
m = nerf.TinyNeRF(39, 256, 8, 4).to(dev)
with torch.no_grad():
    for p, v in zip(m.parameters(), params):
        p.copy_(v.to(dev))
st = m._ensure_packed()
R, S = 4096, 64
g = load_golden("render_8x256")
pose = g["pose"].to(dev)
ro, rd = O.pinhole_rays(100, 100, 138.88887889922103, g["pose"])
idx = torch.randint(0, 10000, (R,), generator=torch.Generator().manual_seed(0))
o, d = ro[idx].contiguous().to(dev), rd[idx].contiguous().to(dev)
tgt = torch.rand(R, 3, device=dev); t = torch.rand(R, S, device=dev)
b = st.repack_bf16(); bp = b.train_plan(R, S)
ztab = ops.depth_table(2.0, 6.0, S, dev)
comp = torch.empty(R, 3, device=dev); gws = torch.rand(R, 3, device=dev) * 1e-3; sws = torch.empty(4 * R, device=dev); loss = torch.zeros(1, device=dev)
s_ = torch.cuda.current_stream(dev).cuda_stream
calls = {
    "pack": lambda: L.call("tnerf_mlp_pack_bf16", C.byref(st.desc), st.flat.data_ptr(), b.table.data_ptr(), b.packed.data_ptr(), s_),
    "fwd": lambda: L.call("tnerf_train_fwd_fused_bf16", C.byref(st.desc), b.packed.data_ptr(), o.data_ptr(), d.data_ptr(), R, S, ztab.data_ptr(), 1, t.data_ptr(), 0, 0, 1, comp.data_ptr(), bp.stash.data_ptr(), s_),
    "dgrad": lambda: L.call("tnerf_train_dgrad_fused_bf16", C.byref(st.desc), b.packed.data_ptr(), o.data_ptr(), d.data_ptr(), R, S, ztab.data_ptr(), 1, t.data_ptr(), 0, 0, 1, gws.data_ptr(), bp.stash.data_ptr(), s_),
    "wgrad": lambda: L.call("tnerf_wgrad_bf16", C.byref(st.desc), bp.stash.data_ptr(), bp.n_tiles, bp.jobs.data_ptr(), bp.n_jobs, bp.slabs.data_ptr(), s_),
    "reduce": lambda: L.call("tnerf_wgrad_reduce", bp.slabs.data_ptr(), bp.reduce.data_ptr(), st.n_params, st.grad.data_ptr(), s_),
    "step": lambda: L.call("tnerf_train_step_fused_bf16", C.byref(st.desc), b.packed.data_ptr(), o.data_ptr(), d.data_ptr(), tgt.data_ptr(), R, S, ztab.data_ptr(), 1, t.data_ptr(), 0, 0, 1, float(3 * R), comp.data_ptr(), sws.data_ptr(), sws.numel(), loss.data_ptr(), bp.stash.data_ptr(), bp.jobs.data_ptr(), bp.n_jobs, bp.slabs.data_ptr(), bp.reduce.data_ptr(), st.grad.data_ptr(), s_),
}
for name, fn in calls.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20:.4f} ms", flush=True)
