#!/usr/bin/env python3
"""Per-kernel HIP-event times of one train step for any model / batch shape, both precisions, next to the whole step
(hipGraph replay of tnerf_train_step_dataset):   python tools/step_breakdown.py L hidden depth skip rays samples
e.g. the reference's hard-coded model (reference src/train.py:78-79):   python tools/step_breakdown.py 10 128 4 2 2048 64"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd")); sys.path.insert(0, os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"))
from tnerf import ops, trainer, lib
import nerf
from data import make_synthetic_scene
L, hidden, depth, skip, R, S = (int(v) for v in (sys.argv[1:7] if len(sys.argv) >= 7 else (10, 128, 4, 2, 2048, 64)))
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
sc = make_synthetic_scene(n_images=8, seed=0)
images, poses, focal = torch.from_numpy(sc["images"]).to(dev), torch.from_numpy(sc["poses"]).to(dev), float(sc["focal"])
in_dim = 6 * L + 3
f = in_dim * hidden + (depth - 1) * hidden * hidden + (in_dim * hidden if skip else 0) + 4 * hidden
dg = (depth - 1) * hidden * hidden + 4 * hidden
fl = {"render_fwd": 2 * f * R * S, "train_fwd": 2 * f * R * S, "dgrad": 2 * dg * R * S, "wgrad": 2 * f * R * S}


def ev(fn, reps=20):
    fn(); torch.cuda.synchronize()
    es = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in es:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in es]))


for prec, peak in (("fp32", 2516.6 / 6), ("bf16", 2516.6)):
    torch.manual_seed(0)
    m = nerf.TinyNeRF(in_dim, hidden, depth, skip).to(dev)
    with torch.no_grad(): m.sigma[0].bias += 0.5
    opt = trainer.FlatAdam(m, lr=5e-4)
    tr = trainer.DatasetTrainer(m, opt, images, poses, focal, R, S, 2.0, 6.0, seed=1, precision=prec)
    for _ in range(5): tr.step()
    step_ms = ev(tr.step, 50)
    st = m.hip_state()
    g = torch.Generator(device=dev); g.manual_seed(3)
    d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev, generator=g), dim=-1)
    o = (-4.0 * d).contiguous(); u = torch.rand(R, S, device=dev, generator=g)
    ztab = ops.depth_table(2.0, 6.0, S, dev)
    comp = torch.empty(R, 3, device=dev); gws = torch.full((R, 3), 1e-4, device=dev); dep = torch.empty(R, 1, device=dev); acc = torch.empty(R, 1, device=dev)
    sp = torch.cuda.current_stream(dev).cuda_stream
    if prec == "fp32":
        m._ensure_packed(); plan = st.plan(R * S)
        common = (C.byref(st.desc), st.packed.data_ptr(), o.data_ptr(), d.data_ptr(), R, S, ztab.data_ptr(), 1, u.data_ptr(), 0, 0, 1)
        calls = {"render_fwd": lambda: lib.call("tnerf_render_fused", *common, comp.data_ptr(), dep.data_ptr(), acc.data_ptr(), sp),
                 "train_fwd": lambda: lib.call("tnerf_train_fwd_fused", *common, comp.data_ptr(), plan.stash.data_ptr(), plan.Mp, sp),
                 "dgrad": lambda: lib.call("tnerf_train_dgrad_fused", *common, gws.data_ptr(), plan.stash.data_ptr(), plan.Mp, sp),
                 "wgrad": lambda: lib.call("tnerf_wgrad", C.byref(st.desc), plan.stash.data_ptr(), plan.Mp, R * S, plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr(), sp),
                 "reduce": lambda: lib.call("tnerf_wgrad_reduce", plan.slabs.data_ptr(), plan.reduce.data_ptr(), st.n_params, st.grad.data_ptr(), sp)}
        if st.x3_capable and not (st.desc.flags & lib.FLAG_FP32_MFMA):          # what the step launches: the split-bf16 chain
            x3 = st.repack_x3(("probe", 0)); cx = (C.byref(st.desc), x3.packed.data_ptr()) + common[2:]
            calls.update({"render_fwd": lambda: lib.call("tnerf_render_fused_x3", *cx, comp.data_ptr(), dep.data_ptr(), acc.data_ptr(), sp),
                          "train_fwd": lambda: lib.call("tnerf_train_fwd_fused_x3", *cx, comp.data_ptr(), plan.stash.data_ptr(), plan.Mp, sp),
                          "dgrad": lambda: lib.call("tnerf_train_dgrad_fused_x3", *cx, gws.data_ptr(), plan.stash.data_ptr(), plan.Mp, sp)})
    else:
        b = st.repack_bf16(); bp = b.train_plan(R, S)
        common = (C.byref(st.desc), b.packed.data_ptr(), o.data_ptr(), d.data_ptr(), R, S, ztab.data_ptr(), 1, u.data_ptr(), 0, 0, 1)
        calls = {"render_fwd": lambda: lib.call("tnerf_render_fused_bf16", *common, comp.data_ptr(), dep.data_ptr(), acc.data_ptr(), sp),
                 "train_fwd": lambda: lib.call("tnerf_train_fwd_fused_bf16", *common, comp.data_ptr(), bp.stash.data_ptr(), sp),
                 "dgrad": lambda: lib.call("tnerf_train_dgrad_fused_bf16", *common, gws.data_ptr(), bp.stash.data_ptr(), sp),
                 "wgrad": lambda: lib.call("tnerf_wgrad_bf16", C.byref(st.desc), bp.stash.data_ptr(), bp.n_tiles, bp.jobs.data_ptr(), bp.n_jobs, bp.slabs.data_ptr(), sp),
                 "reduce": lambda: lib.call("tnerf_wgrad_reduce", bp.slabs.data_ptr(), bp.reduce.data_ptr(), st.n_params, st.grad.data_ptr(), sp)}
    k = {n: ev(fn) for n, fn in calls.items()}
    tot = sum(fl[n] for n in ("train_fwd", "dgrad", "wgrad"))
    print(f"[{prec}] L={L} {depth}x{hidden} skip {skip}, {R} rays x {S}: step {step_ms * 1e3:.1f} us = {R / step_ms * 1e3:.0f} rays/s, "
          f"{tot / step_ms / 1e9 / peak * 100:.1f} % of the {prec} MFMA peak | sum of the three big kernels {sum(k[n] for n in ('train_fwd', 'dgrad', 'wgrad')) * 1e3:.1f} us")
    for n, ms in k.items():
        print(f"     {n:11s} {ms * 1e3:8.1f} us" + (f"  {fl[n] / ms / 1e9:8.1f} TFLOP/s  {fl[n] / ms / 1e9 / peak * 100:5.1f} %" if n in fl else ""))
