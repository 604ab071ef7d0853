"""Drop-in for the reference's src/train.py: same Config fields / flags, same render_one signature, same
loop (round-robin image, randint pixels, gather, sample -> encode -> MLP -> composite, MSE, Adam,
preview / checkpoint cadence and checkpoint dict keys), with the hot path in libtnerf_hip.so.

Differences that are deliberate: fp32 end to end (the reference switches to fp16 autocast on a
GPU; the parity target is its fp32 path); `Config.fused` selects the single-call fused step
(tnerf_train_step_fused + flat Adam) instead of autograd over the per-function ops; tyro / imageio / tqdm
are optional (absent in the build image): argparse and a built-in PNG writer stand in.

Multi-GPU (new; the reference is single-device): `python src/train.py --gpus N` spawns N rank processes
(or start it under torch.distributed.run).  Every rank seeds identically and draws the SAME global `inds` / jitter,
takes its rows (dist.shard_bounds), normalises its loss by the global 3*n_rand; the flat gradient is all-reduced
(RCCL) before the identical Adam step; rank 0 logs, previews and checkpoints.
"""
import os
import struct
import sys
import time
import zlib
from dataclasses import dataclass, fields
from typing import Optional

import numpy as np
import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _hip import ops, trainer as _trainer, dist as _dist   # noqa: E402
from tnerf import launch as _launch                         # noqa: E402
from data import load_scene                                 # noqa: E402
from encoding import PositionalEncoding                     # noqa: E402
from nerf import TinyNeRF                                   # noqa: E402
from rays import get_rays                                   # noqa: E402
from sampling import stratified_samples                     # noqa: E402
from volume import volume_render                            # noqa: E402
from utils import mse2psnr                                  # noqa: E402


@dataclass
class Config:
    iters: int = 20000
    n_rand: int = 2048
    n_samples: int = 64
    lr: float = 5e-4
    near: float = 2.0
    far: float = 6.0
    log_every: int = 50
    preview_every: int = 500
    ckpt_every: int = 1000
    ckpt_path: str = "checkpoints/tinynerf_latest.pth"
    out_dir: str = "outputs"
    resume: bool = True
    preview_pose: Optional[int] = None
    # --- additions (not in the reference) ---
    fused: bool = True          # fused train step + flat Adam; False = autograd over the per-function HIP ops
    num_freqs: int = 10         # the reference hard-codes L=10, 4x128, skip 2 (train.py:78-79)
    hidden: int = 128
    depth: int = 4
    skip_at: int = 2
    data_path: str = "data/tiny_nerf_data.npz"   # "synthetic" = the seeded stand-in scene (explicit opt-in; a missing file raises)
    matrix_pipe: Optional[str] = None     # None / "x3": fp16 three-partial-product chain; "fp32_mfma": plain fp32 MFMA (nerf.TinyNeRF)
    gpus: int = 1               # > 1 from a plain shell: spawn that many rank processes (one per GPU, RCCL all-reduce)
    rng: str = "torch"          # "torch": inds / jitter drawn with torch.randint / torch.rand like the reference (parity path);
                                # "philox": drawn inside the kernels from a device-side step counter — the whole step is one
                                # hipGraph replay (speed path, fused only)
    precision: str = "fp32"     # "bf16": bf16 weights/activations on MFMA, fp32 accumulate/compositing/master weights (fused only)


def write_png(path: str, img_u8: np.ndarray) -> None:
    """Minimal RGB8 PNG writer (imageio is optional)."""
    h, w, _ = img_u8.shape
    raw = b"".join(b"\x00" + img_u8[r].tobytes() for r in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def _fusable(model, encoder) -> bool:
    return (isinstance(model, TinyNeRF) and isinstance(encoder, PositionalEncoding) and encoder.include_input
            and encoder.out_dim == model.in_dim and encoder.num_freqs <= 10 and model._chain_kernels_cover())


@torch.no_grad()
def render_one(model: nn.Module, encoder: nn.Module, H: int, W: int, focal: float, pose: torch.Tensor,
               device: torch.device, n_samples: int = 64, near: float = 2.0, far: float = 6.0,
               chunk: int = 8192, *, precision: str = "fp32") -> torch.Tensor:
    """Full image for one pose, chunked over rays, clamped to [0,1].   [reference src/train.py:36-59]
    Each chunk is ONE fused kernel (pose in, colours out) when model/encoder are this package's.
    precision="bf16" (keyword-only addition): the bf16-MFMA kernel of BASELINE cfg 4."""
    model.eval()
    parts = []
    fused = _fusable(model, encoder)
    if precision not in ("fp32", "bf16") or (precision == "bf16" and not fused):
        raise ValueError(f"render_one: precision={precision!r} needs this package's TinyNeRF + PositionalEncoding")
    if fused:      # rays are generated inside the fused kernel: no get_rays launch, no (HW,3) tables
        st, pose_d = model._ensure_packed(), pose.to(device)
        render = ops.render_camera_fused_bf16 if precision == "bf16" else ops.render_camera_fused
        key = tuple(p._version for p in model._param_list())
        for i in range(0, H * W, chunk):
            kw = dict(key=key) if precision == "bf16" else dict(x3_key=key)
            comp, _, _ = render(st, pose_d, H, W, focal, i, min(chunk, H * W - i), near, far, n_samples, **kw)
            parts.append(comp)
        return torch.cat(parts, dim=0).reshape(H, W, 3).clamp(0.0, 1.0)
    rays_o, rays_d = get_rays(H, W, focal, pose.to(device), device=device)
    for i in range(0, rays_o.shape[0], chunk):
        ro, rd = rays_o[i:i + chunk], rays_d[i:i + chunk]
        z_vals, pts = stratified_samples(near, far, n_samples, ro, rd, randomized=False)
        rgb, sigma = model(encoder(pts.reshape(-1, 3)))
        comp, _, _, _ = volume_render(rgb.reshape(pts.shape[0], n_samples, 3), sigma.reshape(pts.shape[0], n_samples, 1), z_vals, rd)
        parts.append(comp)
    return torch.cat(parts, dim=0).reshape(H, W, 3).clamp(0.0, 1.0)


@torch.no_grad()
def render_one_sharded(model: nn.Module, encoder: nn.Module, H: int, W: int, focal: float, pose: torch.Tensor,
                       device: torch.device, n_samples: int = 64, near: float = 2.0, far: float = 6.0,
                       chunk: int = 8192, *, precision: str = "fp32") -> torch.Tensor:
    """render_one with the image's pixels sharded over the ranks of the default process group (contiguous flat
    pixel ranges) and all-gathered; identical to render_one on one rank.  Every rank returns the
    full (H,W,3) image."""
    rank, world = _dist.world()
    if world == 1 or not _fusable(model, encoder):
        return render_one(model, encoder, H, W, focal, pose, device, n_samples, near, far, chunk, precision=precision)
    if precision not in ("fp32", "bf16"):
        raise ValueError(f"render_one_sharded: precision={precision!r}")
    model.eval()
    lo, hi = _dist.shard_bounds(H * W, rank, world)
    st, pose_d = model._ensure_packed(), pose.to(device)
    render = ops.render_camera_fused_bf16 if precision == "bf16" else ops.render_camera_fused
    vkey = tuple(p._version for p in model._param_list())
    kw = dict(key=vkey) if precision == "bf16" else dict(x3_key=vkey)
    parts = [render(st, pose_d, H, W, focal, i, min(chunk, hi - i), near, far, n_samples, **kw)[0]
             for i in range(lo, hi, chunk)]
    local = torch.cat(parts, dim=0) if parts else torch.zeros(0, 3, device=device)
    return _dist.all_gather_rows(local, H * W).reshape(H, W, 3).clamp(0.0, 1.0)


def _save_ckpt(cfg, model, optimizer, step, in_dim):
    torch.save({"model": model.state_dict(), "opt": optimizer.state_dict(), "step": step, "in_dim": in_dim,
                "cfg": dict(hidden=cfg.hidden, depth=cfg.depth, skip_at=cfg.skip_at)}, cfg.ckpt_path)


def main(cfg: Config):
    torch.manual_seed(0); np.random.seed(0)                                   # train.py:63 (every rank: identical init and draws)
    if not torch.cuda.is_available():
        raise RuntimeError("train.py (HIP): no ROCm GPU visible; this package has no CPU path")
    rank, local, world = _launch.read_env()
    if world > 1 and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        n_dev = torch.cuda.device_count()
        if local >= n_dev and not os.environ.get("TNERF_SHARE_GPU"):
            raise RuntimeError(f"rank {rank}: LOCAL_RANK={local} but only {n_dev} GPU(s) visible")
        torch.cuda.set_device(local % n_dev)
        backend = os.environ.get("TNERF_DIST_BACKEND", "nccl")                # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local % n_dev))
        else:
            torch.distributed.init_process_group(backend)
    rank, world = _dist.world()
    device = torch.device("cuda", torch.cuda.current_device())
    chief = rank == 0
    if chief:
        os.makedirs(cfg.out_dir, exist_ok=True)
        os.makedirs(os.path.dirname(cfg.ckpt_path) or ".", exist_ok=True)
        print(f"[device] {device} torch={torch.__version__}" + (f" ranks={world}" if world > 1 else ""))

    d = load_scene(cfg.data_path)                                             # train.py:70-75 (a missing file raises, as there)
    images = torch.from_numpy(d["images"]).to(device)
    poses = torch.from_numpy(d["poses"]).to(device)
    focal = float(d["focal"])
    N, H, W, _ = images.shape
    if chief:
        print(f"[data] N={N} H={H} W={W} focal={focal:.2f}" + (" (synthetic stand-in)" if d.get("synthetic") else ""))

    encoder = PositionalEncoding(num_freqs=cfg.num_freqs, include_input=True).to(device)
    model = TinyNeRF(in_dim=encoder.out_dim, hidden=cfg.hidden, depth=cfg.depth, skip_at=cfg.skip_at, matrix_pipe=cfg.matrix_pipe).to(device)
    if cfg.rng not in ("torch", "philox") or (cfg.rng == "philox" and not cfg.fused):
        raise SystemExit("--rng must be 'torch' or 'philox' (philox needs the fused step)")
    if cfg.fused:
        optimizer = _trainer.FlatAdam(model, lr=cfg.lr)
        step_fn = _trainer.FusedTrainer(model, optimizer, cfg.near, cfg.far, cfg.n_samples, precision=cfg.precision)
    else:
        if cfg.precision != "fp32":
            raise SystemExit("--precision bf16 needs the fused step (--fused)")
        optimizer = torch.optim.Adam(model.parameters(), lr=cfg.lr)

    start_step = 0
    if cfg.resume and os.path.exists(cfg.ckpt_path):                            # train.py:83-92
        ckpt = torch.load(cfg.ckpt_path, map_location=device)
        model.load_state_dict(ckpt["model"])
        if "opt" in ckpt:
            optimizer.load_state_dict(ckpt["opt"])
        if "step" in ckpt:
            start_step = int(ckpt["step"])
        if chief:
            print(f"[resume] loaded {cfg.ckpt_path} from step {start_step}")
    if world > 1:                                                               # one set of weights, whatever each rank loaded
        st0 = model.hip_state()
        _dist.broadcast_(st0.flat, src=0)
        t = torch.tensor([start_step], device=device); _dist.broadcast_(t, src=0); start_step = int(t.item())

    if not cfg.fused:                                                           # train.py:94-101 (the fused step makes its rays in-kernel)
        rays = [get_rays(H, W, focal, poses[i], device=device) for i in range(N)]
        all_rays_o = torch.stack([r[0] for r in rays], dim=0)
        all_rays_d = torch.stack([r[1] for r in rays], dim=0)
    pixels = images.view(N, H * W, 3)
    lo, hi = _dist.shard_bounds(cfg.n_rand, rank, world)                        # this rank's rows of the global batch
    graph_step = None
    if cfg.fused and cfg.rng == "philox":
        graph_step = _trainer.DatasetTrainer(model, optimizer, images, poses, focal, cfg.n_rand, cfg.n_samples, cfg.near, cfg.far,
                                             seed=0, precision=cfg.precision, start_step=start_step)

    pbar = range(start_step, cfg.iters)
    if chief:
        try:
            from tqdm import tqdm
            pbar = tqdm(pbar, desc="train")
        except ImportError:
            pass
    t0 = time.time()
    for step in pbar:
        model.train()
        img_i = step % N
        if graph_step is None:
            inds = torch.randint(0, H * W, (cfg.n_rand,), device=device)         # train.py:109 — the GLOBAL draw on every rank
        if graph_step is not None:
            loss, _ = graph_step.step()                                           # image index, pixel and jitter draws happen in the kernels
        elif cfg.fused:
            t_rand = torch.rand(cfg.n_rand, cfg.n_samples, device=device)        # the draw of sampling.py:24
            loss, _ = step_fn.step_camera(poses[img_i], H, W, focal, inds[lo:hi], pixels[img_i], t_rand=t_rand[lo:hi],
                                          global_rays=cfg.n_rand)
        else:
            t_rand = None
            ro, rd, target = all_rays_o[img_i, inds], all_rays_d[img_i, inds], pixels[img_i, inds]
            if world > 1:          # same generator position on every rank: draw the global jitter, use this rank's rows
                t_rand = torch.rand(cfg.n_rand, cfg.n_samples, device=device)[lo:hi]
                ro, rd, target = ro[lo:hi], rd[lo:hi], target[lo:hi]
                z_vals, pts, _ = ops.sample_along_rays(cfg.near, cfg.far, cfg.n_samples, ro, rd, True, t_rand=t_rand)
            else:
                z_vals, pts = stratified_samples(cfg.near, cfg.far, cfg.n_samples, ro, rd, randomized=True)
            rgb, sigma = model(encoder(pts.reshape(-1, 3)))
            comp_rgb, _, _, _ = volume_render(rgb.reshape(hi - lo, cfg.n_samples, 3),
                                              sigma.reshape(hi - lo, cfg.n_samples, 1), z_vals, rd)
            loss = torch.sum((comp_rgb - target) ** 2) / (3.0 * cfg.n_rand)       # == torch.mean(...) on one rank (train.py:122)
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            if world > 1:
                for p in model.parameters():
                    _dist.all_reduce_sum_(p.grad)
            optimizer.step()

        if (step + 1) % cfg.log_every == 0:
            lg = loss.detach().clone()
            if world > 1:
                _dist.all_reduce_sum_(lg)                                         # shard shares add up to the batch MSE
            if chief:
                msg = dict(loss=float(lg.item()), psnr=float(mse2psnr(lg).item()))
                pbar.set_postfix(**msg) if hasattr(pbar, "set_postfix") else print(f"[{step + 1}] {msg}")
        if (step + 1) % cfg.preview_every == 0:
            pose_idx = (img_i + 1 if cfg.preview_pose is None else cfg.preview_pose) % N
            img = render_one_sharded(model, encoder, H, W, focal, poses[pose_idx], device, n_samples=cfg.n_samples, near=cfg.near, far=cfg.far)
            if chief:
                write_png(f"{cfg.out_dir}/preview_{step + 1:06d}.png", (img.cpu().numpy() * 255).astype(np.uint8))
        if (step + 1) % cfg.ckpt_every == 0 and chief:
            _save_ckpt(cfg, model, optimizer, step + 1, encoder.out_dim)

    dt = time.time() - t0
    img = render_one_sharded(model, encoder, H, W, focal, poses[-1], device, n_samples=cfg.n_samples, near=cfg.near, far=cfg.far)
    if chief:
        _save_ckpt(cfg, model, optimizer, cfg.iters, encoder.out_dim)
        write_png(f"{cfg.out_dir}/final.png", (img.cpu().numpy() * 255).astype(np.uint8))
        print(f"[done] {cfg.iters} iters in {dt / 60:.2f} min | saved {cfg.ckpt_path} and {cfg.out_dir}/final.png")
    if world > 1 and os.environ.get("TNERF_SPAWNED"):
        torch.distributed.destroy_process_group()
    return model


def _parse_cli() -> Config:
    try:
        import tyro
        return tyro.cli(Config)
    except ImportError:
        import argparse
        ap = argparse.ArgumentParser()
        for f in fields(Config):
            flag = "--" + f.name.replace("_", "-")
            if f.type is bool:
                ap.add_argument(flag, action=argparse.BooleanOptionalAction, default=f.default)
            elif f.name == "preview_pose":
                ap.add_argument(flag, type=int, default=None)
            else:
                ap.add_argument(flag, type=type(f.default), default=f.default)
        return Config(**vars(ap.parse_args()))


if __name__ == "__main__":
    _cfg = _parse_cli()
    if _cfg.gpus > 1 and not _launch.under_launcher():
        # plain `python src/train.py --gpus N`: this parent never touches the GPU; N fresh rank processes do
        sys.exit(_launch.spawn_ranks(_cfg.gpus, [os.path.abspath(__file__), *sys.argv[1:]]))
    main(_cfg)
