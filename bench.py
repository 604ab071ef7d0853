#!/usr/bin/env python3
"""bench.py — rays/s of the TinyNeRF train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the training hot path over one batch (reference src/train.py:108-128):
randint pixel indices -> gather rays/targets -> jitter draw -> fused forward (sample, encode, 8x256 MLP,
composite) -> MSE -> backward (dgrad chain, weight-gradient GEMMs, slab reduce) -> [RCCL all-reduce] -> Adam.
Workload (BASELINE.json configs[1]): 100x100 scene, 106 views, L=6 (39 inputs), 8x256 ReLU MLP skip 4,
64 samples/ray, 4096 rays per GPU per step, fp32.  Data: seeded synthetic scene (the dataset blob is not
available offline), resident in HBM before the timed region.  N>1: one process per GPU, each with its own
4096 rays (weak scaling), one all-reduce(SUM) of the 1.93 MB flat gradient per step.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     — the dominant kernel's algorithmic FLOP/s against the fp32 MFMA peak (157.3 TFLOP/s),
                 kernel time measured live with HIP events on the launch stream,
  cpu_baseline — the CPU oracle (a port of the reference's fp32 CPU path) timed on this box's host cores on a
                 bounded sample (rank 0, N=1 only),
  kernels / psnr — per-kernel times and the PSNR reached (context, not part of the contract),
  bf16         — BASELINE.json configs[3]: the same step with bf16 weights/activations on MFMA (fp32 accumulate,
                 fp32 compositing, fp32 master weights), same data stream: step time, per-kernel times and the
                 PSNR difference against the fp32 run above.  Context only: the headline stays fp32.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "tiny-nerf-pytorch_amd")
for p in (ROOT, PKG, os.path.join(PKG, "src")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np          # noqa: E402
import torch                # noqa: E402
import torch.distributed as dist   # noqa: E402

L_FREQS, HIDDEN, DEPTH, SKIP = 6, 256, 8, 4
RAYS, SAMPLES, NEAR, FAR, LR = 4096, 64, 2.0, 6.0, 5e-4
PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 FLOP/clk/CU
PEAK_BF16_MFMA_TFLOPS = 2516.6      # 16x the fp32 MFMA rate (v_mfma_f32_32x32x16_bf16: 32 cycles per 32768 FLOP per SIMD)
PEAK_HBM_GBS = 8000.0
IN_DIM = 6 * L_FREQS + 3
MACS_PER_SAMPLE = IN_DIM * HIDDEN + (DEPTH - 1) * HIDDEN * HIDDEN + IN_DIM * HIDDEN + 4 * HIDDEN   # 479,744
FLOPS_FWD_PER_RAY = 2 * MACS_PER_SAMPLE * SAMPLES                                                   # 61,407,232


def algorithmic_flops():
    """Per 4096-ray launch of each MFMA kernel."""
    m = RAYS * SAMPLES
    fwd = 2 * MACS_PER_SAMPLE * m
    dgrad = 2 * ((DEPTH - 1) * HIDDEN * HIDDEN + 4 * HIDDEN) * m       # no gradient flows into the encoder
    wgrad = 2 * MACS_PER_SAMPLE * m
    return {"render_fwd": fwd, "train_fwd": fwd, "dgrad": dgrad, "wgrad": wgrad}


def host_cores() -> int:
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("TNERF_CPU_THREADS", "32"))))


def cpu_baseline(scene, seconds_budget=20.0):
    """CPU oracle (port of the reference's fp32 CPU path) on the same workload, bounded sample."""
    from oracle import tnerf_oracle as O
    torch.set_num_threads(host_cores())
    g = torch.Generator().manual_seed(0)
    params = O.mlp_init(IN_DIM, HIDDEN, DEPTH, SKIP, g)
    params[2 * DEPTH + 1] += 0.5                       # sigma bias nudge, as on the GPU side
    adam = O.AdamState(params, lr=LR)
    images, poses, focal = torch.from_numpy(scene["images"]), torch.from_numpy(scene["poses"]), float(scene["focal"])
    N, H, W, _ = images.shape
    ro_all, rd_all = O.pinhole_rays(H, W, focal, poses[0])
    pix = images[0].reshape(-1, 3)
    times = []
    t_start = time.time()
    for it in range(6):
        inds = torch.randint(0, H * W, (RAYS,), generator=g)
        u = torch.rand(RAYS, SAMPLES, generator=g)
        t0 = time.time()
        _, _, grads = O.loss_and_grads(params, SKIP, L_FREQS, ro_all[inds], rd_all[inds], pix[inds], NEAR, FAR, SAMPLES, u)
        adam.step(params, grads)
        dt = time.time() - t0
        if it > 0:
            times.append(dt)
        if time.time() - t_start > seconds_budget and len(times) >= 2:
            break
    med = float(np.median(times))
    return {"value": RAYS / med, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(times)} train steps of {RAYS} rays x {SAMPLES} samples after 1 warm-up (median {med:.2f} s/step), "
                      f"oracle/tnerf_oracle.py in PyTorch CPU fp32, torch {torch.__version__}"}


def main():
    # The contract is ONE JSON line on stdout.  Libraries chat on fd 1 (RCCL prints a 5-line version banner at
    # communicator creation): point fd 1 at stderr for the duration of the run and restore it for the final print.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--psnr-steps", type=int, default=1000, help="extra untimed steps before reporting PSNR (rank 0 context)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--philox", action="store_true", help="draw the jitter in-kernel (Philox) instead of torch.rand")
    ap.add_argument("--no-bf16", action="store_true", help="skip the bf16-mode section (BASELINE.json configs[3])")
    ap.add_argument("--ray-tables", action="store_true",
                    help="gather rays/targets from precomputed (N,HW,3) tables like the reference loop (train.py:94-112) "
                         "instead of generating them in the kernel from pose + pixel index")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or bool(os.environ.get("TNERF_FORCE_DIST"))      # the env var exercises RCCL init with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from tnerf import ops, trainer, lib
    from data import make_synthetic_scene
    import nerf as nerf_mod
    import train as train_mod
    from encoding import PositionalEncoding
    from utils import mse2psnr

    scene = make_synthetic_scene(seed=0)
    images = torch.from_numpy(scene["images"]).to(dev)
    poses = torch.from_numpy(scene["poses"]).to(dev)
    focal = float(scene["focal"])
    N, H, W, _ = images.shape

    encoder = PositionalEncoding(L_FREQS, True).to(dev)

    def make_trainer(precision):
        torch.manual_seed(0)                               # identical initial weights on every rank (and in both modes)
        mdl = nerf_mod.TinyNeRF(encoder.out_dim, HIDDEN, DEPTH, SKIP).to(dev)
        with torch.no_grad():
            # nn.Linear's default init leaves the sigma head at exactly 0 after its ReLU for this 8x256 model
            # (SURVEY.md §7-7): every weight and every gradient would be a zero and the MFMA kernels would be
            # timed on zeros (which clock higher).  Nudge the bias so the network is alive, as the fixtures do.
            mdl.sigma[0].bias += 0.5
        op = trainer.FlatAdam(mdl, lr=LR)
        return mdl, op, trainer.FusedTrainer(mdl, op, NEAR, FAR, SAMPLES, precision=precision)

    model, opt, tr = make_trainer("fp32")

    import rays as rays_mod
    all_o, all_d = [], []
    for i in range(N):                                     # train.py:94-101
        ro, rd = rays_mod.get_rays(H, W, focal, poses[i])
        all_o.append(ro); all_d.append(rd)
    all_o, all_d = torch.stack(all_o), torch.stack(all_d)
    pixels = images.view(N, H * W, 3)
    gen = torch.Generator(device=dev); gen.manual_seed(1234)          # same draws on every rank; rank takes its shard
    state = {"step": 0}
    run = {"tr": tr}

    def one_step():
        tr = run["tr"]
        s = state["step"]; state["step"] += 1
        img_i = s % N
        inds = torch.randint(0, H * W, (world * RAYS,), device=dev, generator=gen)[rank * RAYS:(rank + 1) * RAYS]
        kw = dict(global_rays=world * RAYS)
        if args.philox:
            kw["philox"] = (1234, s * world * RAYS * SAMPLES + rank * RAYS * SAMPLES)
        else:
            kw["t_rand"] = torch.rand(world * RAYS, SAMPLES, device=dev, generator=gen)[rank * RAYS:(rank + 1) * RAYS]
        if args.ray_tables:
            return tr.step(all_o[img_i, inds], all_d[img_i, inds], pixels[img_i, inds], **kw)
        return tr.step_camera(poses[img_i], H, W, focal, inds, pixels[img_i], **kw)

    def fence():
        if use_dist:
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = one_step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * RAYS * args.steps / dt

    out = {"metric": "rays/s (train step)", "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "train step: 100x100 synthetic Lego stand-in, 106 views, L=6 posenc, 8x256 ReLU MLP (skip 4), "
                                  "64 samples/ray, 4096 rays per GPU per step, Adam, fp32 (BASELINE.json configs[1])",
                      "rays_per_gpu": RAYS, "samples_per_ray": SAMPLES, "jitter": "philox-in-kernel" if args.philox else "torch.rand", "rays": "precomputed tables + gather" if args.ray_tables else "generated in-kernel from pose + pixel index",
                      "parallelism": f"rays sharded x{world}, 1 all-reduce of {opt._st.n_params * 4} B per step" if world > 1 else "single GPU"}}

    if rank == 0:
        # ---- per-kernel times, live, HIP events on the launch stream (torch's current stream)
        st = model.hip_state(); plan = st.plan(RAYS * SAMPLES)
        import ctypes as C
        img_i = 0
        inds = torch.randint(0, H * W, (RAYS,), device=dev, generator=gen)
        ro, rd, tgt = all_o[img_i, inds].contiguous(), all_d[img_i, inds].contiguous(), pixels[img_i, inds].contiguous()
        u = torch.rand(RAYS, SAMPLES, device=dev, generator=gen)
        ztab = ops.depth_table(NEAR, FAR, SAMPLES, dev)
        comp = torch.empty(RAYS, 3, device=dev); gws = torch.full((RAYS, 3), 1e-4, device=dev)
        model._ensure_packed()
        sp = torch.cuda.current_stream(dev).cuda_stream
        common = (C.byref(st.desc), st.packed.data_ptr(), ro.data_ptr(), rd.data_ptr(), RAYS, SAMPLES, ztab.data_ptr(), 1, u.data_ptr(), 0, 0, 1)
        dep = torch.empty(RAYS, 1, device=dev); acc_ = torch.empty(RAYS, 1, device=dev)
        calls = {
            "render_fwd": lambda: lib.call("tnerf_render_fused", *common, comp.data_ptr(), dep.data_ptr(), acc_.data_ptr(), sp),
            "train_fwd": lambda: lib.call("tnerf_train_fwd_fused", *common, comp.data_ptr(), plan.stash.data_ptr(), plan.Mp, sp),
            "dgrad": lambda: lib.call("tnerf_train_dgrad_fused", *common, gws.data_ptr(), plan.stash.data_ptr(), plan.Mp, sp),
            "wgrad": lambda: lib.call("tnerf_wgrad", C.byref(st.desc), plan.stash.data_ptr(), plan.Mp, RAYS * SAMPLES, plan.jobs.data_ptr(), plan.n_jobs, plan.slabs.data_ptr(), sp),
            "reduce": lambda: lib.call("tnerf_wgrad_reduce", plan.slabs.data_ptr(), plan.reduce.data_ptr(), st.n_params, st.grad.data_ptr(), sp),
        }
        kern = {}
        reps = max(5, min(20, args.steps))
        for name, fn in calls.items():
            fn(); torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in evs:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            kern[name] = float(np.mean([a.elapsed_time(b) for a, b in evs]))          # ms
        fl = algorithmic_flops()
        step_kernels = ("train_fwd", "dgrad", "wgrad")
        dom = max(step_kernels, key=lambda k: kern[k])
        ach = fl[dom] / (kern[dom] * 1e-3) / 1e12
        traffic = None
        try:      # HBM bytes per launch from the committed PMC passes (profiles/r01_traffic.json; see its _note)
            traffic = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))[dom]["hbm_bytes"]
        except (OSError, KeyError, ValueError):
            pass
        out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                           "flops_per_launch": fl[dom], "ms_per_launch": kern[dom]}
        out["kernels"] = {k: {"ms": kern[k], "tflops": (fl[k] / (kern[k] * 1e-3) / 1e12) if k in fl else None} for k in kern}
        step_flops = sum(fl[k] for k in step_kernels)
        out["step_mfma_frac"] = step_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS

    # ---- PSNR context: keep training (untimed), then report minibatch PSNR and a full-image PSNR
    if args.psnr_steps > 0:
        losses = []
        for i in range(args.psnr_steps):
            loss, _ = one_step()
            if i >= args.psnr_steps - 20:
                losses.append(loss.clone())
        if rank == 0:
            mb = float(mse2psnr(torch.stack(losses).mean() * world))          # each rank's loss is 1/world of the global mean
            img = train_mod.render_one(model, encoder, H, W, focal, poses[N - 1], dev, n_samples=SAMPLES, near=NEAR, far=FAR)
            full = float(mse2psnr(torch.mean((img - images[N - 1]) ** 2)))
            out["psnr"] = {"train_minibatch_db": mb, "full_image_view105_db": full, "after_steps": state["step"]}

    # ---- bf16 mode (BASELINE.json configs[3]): same initial weights, same pixel / jitter stream, same step counts
    if not args.no_bf16:
        model16, opt16, tr16 = make_trainer("bf16")
        run["tr"] = tr16
        gen.manual_seed(1234); state["step"] = 0
        for _ in range(args.warmup):
            one_step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        fence()
        dt16 = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt16], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt16 = float(t.item())
        b16 = {"dtype": "bf16 operands, fp32 accumulate / compositing / master weights", "ms_per_step": dt16 / args.steps * 1e3,
               "value": world * RAYS * args.steps / dt16, "unit": "rays/s", "speedup_vs_fp32_step": dt / dt16}
        if rank == 0:
            import ctypes as C
            st = model16.hip_state(); b = st.repack_bf16(); bp = b.train_plan(RAYS, SAMPLES)
            inds = torch.randint(0, H * W, (RAYS,), device=dev, generator=gen)
            ro, rd = all_o[0, inds].contiguous(), all_d[0, inds].contiguous()
            u = torch.rand(RAYS, SAMPLES, device=dev, generator=gen)
            ztab = ops.depth_table(NEAR, FAR, SAMPLES, dev)
            comp = torch.empty(RAYS, 3, device=dev); gws = torch.full((RAYS, 3), 1e-4, device=dev)
            dep = torch.empty(RAYS, 1, device=dev); acc_ = torch.empty(RAYS, 1, device=dev)
            sp = torch.cuda.current_stream(dev).cuda_stream
            common = (C.byref(st.desc), b.packed.data_ptr(), ro.data_ptr(), rd.data_ptr(), RAYS, SAMPLES, ztab.data_ptr(), 1, u.data_ptr(), 0, 0, 1)
            calls = {
                "render_fwd": lambda: lib.call("tnerf_render_fused_bf16", *common, comp.data_ptr(), dep.data_ptr(), acc_.data_ptr(), sp),
                "train_fwd": lambda: lib.call("tnerf_train_fwd_fused_bf16", *common, comp.data_ptr(), bp.stash.data_ptr(), sp),
                "dgrad": lambda: lib.call("tnerf_train_dgrad_fused_bf16", *common, gws.data_ptr(), bp.stash.data_ptr(), sp),
                "wgrad": lambda: lib.call("tnerf_wgrad_bf16", C.byref(st.desc), bp.stash.data_ptr(), bp.n_tiles, bp.jobs.data_ptr(), bp.n_jobs, bp.slabs.data_ptr(), sp),
            }
            fl = algorithmic_flops()
            # algorithmic HBM bytes of the stash streams: 2 KB per (32-sample tile, 32-feature tile) of bf16
            # activations / activation gradients; the forward also writes the ReLU bits and the head outputs
            NT = HIDDEN // 32
            tiles = RAYS * ((SAMPLES + 31) // 32)
            wg_tiles = (NT + 2) + (DEPTH - 1) * 2 * NT + ((NT + 2) if SKIP else 0) + (1 + NT)     # sum over job classes of A + B feature tiles
            hbm = {"train_fwd": tiles * ((2 + DEPTH * NT) * 2048 + DEPTH * (HIDDEN // 64) * 256 + 512),
                   "dgrad": tiles * (DEPTH * NT + 1) * 2048,
                   "wgrad": tiles * wg_tiles * 2048}
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic_bf16.json")))
            except (OSError, ValueError):
                pmc = {}
            kern16 = {}
            for name, fn in calls.items():
                fn(); torch.cuda.synchronize()
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
                for a_, b_ in evs:
                    a_.record(); fn(); b_.record()
                torch.cuda.synchronize()
                ms = float(np.mean([a_.elapsed_time(b_) for a_, b_ in evs]))
                kern16[name] = {"ms": ms, "tflops": fl[name] / (ms * 1e-3) / 1e12,
                                "mfma_frac": fl[name] / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS}
                if name in hbm:
                    kern16[name]["algorithmic_hbm_bytes"] = hbm[name]
                    kern16[name]["hbm_gbs"] = hbm[name] / (ms * 1e-3) / 1e9
                    kern16[name]["hbm_frac"] = hbm[name] / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS
                    kern16[name]["traffic"] = pmc.get(name, {}).get("hbm_bytes")
            b16["kernels"] = kern16
            b16["roofline"] = {"bound": "hbm", "kernel": "wgrad", "achieved": kern16["wgrad"]["hbm_gbs"], "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": kern16["wgrad"]["hbm_frac"], "traffic": kern16["wgrad"]["traffic"]}
        if args.psnr_steps > 0:
            losses = []
            for i in range(args.psnr_steps):
                loss, _ = one_step()
                if i >= args.psnr_steps - 20:
                    losses.append(loss.clone())
            if rank == 0:
                mb = float(mse2psnr(torch.stack(losses).mean() * world))
                img = train_mod.render_one(model16, encoder, H, W, focal, poses[N - 1], dev, n_samples=SAMPLES, near=NEAR, far=FAR)
                full = float(mse2psnr(torch.mean((img - images[N - 1]) ** 2)))
                st16 = model16._ensure_packed()
                img16 = ops.render_camera_fused_bf16(st16, poses[N - 1], H, W, focal, 0, H * W, NEAR, FAR, SAMPLES)[0].reshape(H, W, 3).clamp(0, 1)
                b16["psnr"] = {"train_minibatch_db": mb, "full_image_view105_db": full, "after_steps": state["step"],
                               "full_image_rendered_in_bf16_db": float(mse2psnr(torch.mean((img16 - images[N - 1]) ** 2))),
                               "max_abs_rgb_bf16_vs_fp32_render": float((img16 - img).abs().max())}
                if "psnr" in out:
                    b16["psnr"]["delta_full_image_db_vs_fp32"] = full - out["psnr"]["full_image_view105_db"]
        out["bf16"] = b16
        run["tr"] = tr

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene)
    fence()
    if use_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
