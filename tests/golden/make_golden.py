#!/usr/bin/env python3
"""
Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's hot-path modules (src/rays.py, sampling.py, encoding.py, nerf.py,
volume.py, utils.py, camera.py) on CPU fp32, feeds them seeded inputs and stores inputs + outputs
as compressed .npz.  `train.py` itself cannot be imported here (imageio / tyro are not installed),
so the 12-line step body (train.py:106-128) and `render_one` (train.py:36-59) are driven from this
script around the *imported* reference functions; every arithmetic op still executes reference code.

The fixtures are data (inputs and expected outputs) — no reference source text is stored.
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("TNERF_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "src"))
sys.dont_write_bytecode = True

from rays import get_rays                      # noqa: E402
from sampling import stratified_samples        # noqa: E402
from encoding import PositionalEncoding        # noqa: E402
from nerf import TinyNeRF                      # noqa: E402
from volume import volume_render               # noqa: E402
from utils import mse2psnr                     # noqa: E402
import camera                                  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
FOCAL = 138.88887889922103
torch.set_num_threads(1)


def save(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KB")


def look_at_pose(theta_deg, phi_deg, radius):
    """c2w with camera on a sphere looking at the origin (-z forward, y up)."""
    th, ph = np.deg2rad(theta_deg), np.deg2rad(phi_deg)
    eye = np.array([radius * np.cos(ph) * np.cos(th), radius * np.cos(ph) * np.sin(th), radius * np.sin(ph)])
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0])); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up, -fwd, eye
    return torch.tensor(m, dtype=torch.float32)


def poses3():
    ident = torch.eye(4); ident[2, 3] = 4.0
    generic = look_at_pose(37.0, 28.0, 4.03)
    spiral = camera.spiral_poses(generic, n_frames=7, radius=0.3)[3]
    return torch.stack([ident, generic, spiral])


# ------------------------------------------------------------------ rays / linspace / sampling
def fx_rays():
    P = poses3()
    out = {"poses": P, "focal": np.float64(FOCAL)}
    for pi in range(3):
        ro, rd = get_rays(5, 7, FOCAL, P[pi])
        out[f"o_5x7_{pi}"], out[f"d_5x7_{pi}"] = ro.contiguous(), rd
        ro, rd = get_rays(100, 100, FOCAL, P[pi])
        idx = torch.arange(0, 10000, 19)[:512]
        out[f"idx_100_{pi}"], out[f"o_100_{pi}"], out[f"d_100_{pi}"] = idx, ro[idx].contiguous(), rd[idx]
    save("rays", **out)


def fx_sampling():
    g = torch.Generator().manual_seed(101)
    ro, rd = get_rays(100, 100, FOCAL, poses3()[1])
    pick = torch.randperm(10000, generator=g)[:96]
    ro, rd = ro[pick].contiguous(), rd[pick].contiguous()
    out = {"rays_o": ro, "rays_d": rd, "near": np.float64(2.0), "far": np.float64(6.0)}
    for S in (64, 128, 256):
        t = torch.linspace(0., 1., steps=S)
        out[f"linspace_{S}"] = t
        out[f"zbase_{S}"] = 2.0 * (1. - t) + 6.0 * t
        z, pts = stratified_samples(2.0, 6.0, S, ro, rd, randomized=False)
        out[f"z_det_{S}"], out[f"pts_det_{S}"] = z[:4].contiguous(), pts[:4].contiguous()
        # randomized=True draws torch.rand_like(z) from the default generator: reproduce the draw
        torch.manual_seed(1000 + S)
        z, pts = stratified_samples(2.0, 6.0, S, ro, rd, randomized=True)
        torch.manual_seed(1000 + S)
        u = torch.rand_like(z)
        out[f"u_{S}"], out[f"z_rand_{S}"] = u, z
        out[f"pts_rand_{S}"] = pts if S == 64 else pts[:32].contiguous()
    # odd near/far
    torch.manual_seed(77)
    z, pts = stratified_samples(0.5, 3.25, 64, ro, rd, randomized=True)
    torch.manual_seed(77)
    out["u_odd"], out["z_odd"], out["pts_odd"] = torch.rand_like(z), z, pts[:32].contiguous()
    save("sampling", **out)


def fx_encoding():
    g = torch.Generator().manual_seed(202)
    ro, rd = get_rays(100, 100, FOCAL, poses3()[1])
    pick = torch.randperm(10000, generator=g)[:64]
    torch.manual_seed(5)
    _, pts = stratified_samples(2.0, 6.0, 64, ro[pick].contiguous(), rd[pick].contiguous(), randomized=True)
    x = pts.reshape(-1, 3)                                     # 4096 points
    out = {"x": x}
    for L in (6, 10):
        for inc in (True, False):
            enc = PositionalEncoding(L, inc)
            y = enc(x)
            assert y.shape[-1] == enc.out_dim
            out[f"enc_L{L}_{int(inc)}"] = y if (L == 6 and inc) else y[:512].contiguous()
    save("encoding", **out)


# ------------------------------------------------------------------ synthetic mini scene for training
def mini_scene(N, H, W, seed):
    """Seeded toy dataset with the npz schema of tiny_nerf_data (images (N,H,W,3), poses (N,4,4), focal)."""
    rng = np.random.RandomState(seed)
    poses = torch.stack([look_at_pose(rng.uniform(0, 360), rng.uniform(15, 60), 4.03) for _ in range(N)])
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    imgs = []
    for i in range(N):
        blob = np.exp(-((xx - 0.2 * np.cos(i)) ** 2 + (yy - 0.2 * np.sin(i)) ** 2) / 0.18)
        rgb = 1.0 - blob[..., None] * (1.0 - np.array([0.8, 0.45 + 0.1 * np.sin(i), 0.1]))
        imgs.append(rgb)
    images = torch.tensor(np.stack(imgs), dtype=torch.float32)
    focal = FOCAL * W / 100.0
    return images, poses, focal


def train_steps(model, encoder, images, poses, focal, n_rand, n_samples, steps, lr, record):
    """The reference step body (train.py:94-128) on CPU: autocast/GradScaler are disabled there."""
    N, H, W, _ = images.shape
    all_o, all_d = [], []
    for i in range(N):
        ro, rd = get_rays(H, W, focal, poses[i])
        all_o.append(ro); all_d.append(rd)
    all_o, all_d = torch.stack(all_o), torch.stack(all_d)
    pixels = images.view(N, H * W, 3)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    for step in range(steps):
        img_i = step % N
        inds = torch.randint(0, H * W, (n_rand,))
        ro, rd, target = all_o[img_i, inds], all_d[img_i, inds], pixels[img_i, inds]
        state = torch.random.get_rng_state()
        z_vals, pts = stratified_samples(2.0, 6.0, n_samples, ro, rd, randomized=True)
        torch.random.set_rng_state(state)
        u = torch.rand_like(z_vals)                      # identical draw to the one inside stratified_samples
        xenc = encoder(pts.reshape(-1, 3))
        rgb, sigma = model(xenc)
        comp, _, _, _ = volume_render(rgb.reshape(n_rand, n_samples, 3), sigma.reshape(n_rand, n_samples, 1), z_vals, rd)
        loss = torch.mean((comp - target) ** 2)
        psnr = mse2psnr(loss)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if record is not None:
            record(step, inds, u, ro, rd, target, comp, loss, psnr, [p.grad.clone() for p in model.parameters()])
        opt.step()


def trained_model(L, hidden, depth, skip_at, seed):
    """A non-degenerate state: seed-0 init + sigma bias nudge, then 20 reference train steps."""
    torch.manual_seed(seed)
    enc = PositionalEncoding(L, True)
    model = TinyNeRF(enc.out_dim, hidden, depth, skip_at)
    with torch.no_grad():
        model.sigma[0].bias += 0.5                      # untrained 8x256 renders pure white otherwise
    images, poses, focal = mini_scene(4, 16, 16, seed=3)
    train_steps(model, enc, images, poses, focal, n_rand=128, n_samples=32, steps=20, lr=5e-4, record=None)
    return enc, model


def flat_params(model):
    return [p.detach().clone() for p in model.parameters()]


def fx_model(tag, L, hidden, depth, skip_at):
    enc, model = trained_model(L, hidden, depth, skip_at, seed=0)
    names = [n for n, _ in model.named_parameters()]
    weights = {f"p{i:02d}": p for i, p in enumerate(flat_params(model))}
    save(f"weights_{tag}", names=np.array(names), cfg=np.array([L, hidden, depth, skip_at]), **weights)

    # ---- mlp fwd/bwd on 2048 encoded inputs with a fixed upstream gradient
    g = torch.Generator().manual_seed(303)
    ro, rd = get_rays(100, 100, FOCAL, poses3()[1])
    pick = torch.randperm(10000, generator=g)[:32]
    torch.manual_seed(9)
    z, pts = stratified_samples(2.0, 6.0, 64, ro[pick].contiguous(), rd[pick].contiguous(), randomized=True)
    x = enc(pts.reshape(-1, 3))                                       # (2048, D)
    rgb, sigma = model(x)
    g_rgb = torch.randn(rgb.shape, generator=g) * 0.1
    g_sig = torch.randn(sigma.shape, generator=g) * 0.1
    model.zero_grad(set_to_none=True)
    (rgb * g_rgb).sum().add((sigma * g_sig).sum()).backward()
    grads = {f"g{i:02d}": p.grad for i, p in enumerate(model.parameters())}
    save(f"mlp_{tag}", x=x, rgb=rgb, sigma=sigma, g_rgb=g_rgb, g_sigma=g_sig, **grads)

    # ---- full-image render, two chunk sizes (chunk invariance), S=64
    pose = poses3()[1]
    H = W = 100 if tag == "8x256" else 40
    focal = FOCAL * W / 100.0
    imgs = {}
    with torch.no_grad():
        for chunk in (8192, 1000):
            rays_o, rays_d = get_rays(H, W, focal, pose)
            outs = []
            for i in range(0, H * W, chunk):
                zz, pp = stratified_samples(2.0, 6.0, 64, rays_o[i:i + chunk], rays_d[i:i + chunk], randomized=False)
                c_rgb, c_sig = model(enc(pp.reshape(-1, 3)))
                comp, _, _, _ = volume_render(c_rgb.reshape(pp.shape[0], 64, 3), c_sig.reshape(pp.shape[0], 64, 1), zz, rays_d[i:i + chunk])
                outs.append(comp)
            imgs[chunk] = torch.cat(outs, 0).reshape(H, W, 3).clamp(0., 1.)
    assert torch.equal(imgs[8192], imgs[1000]) or (imgs[8192] - imgs[1000]).abs().max() < 1e-6
    save(f"render_{tag}", pose=pose, H=np.int64(H), W=np.int64(W), focal=np.float64(focal),
         img=imgs[8192], img_chunk1000=imgs[1000])

    # ---- 10 train steps from these weights on the mini scene: RNG order + fwd/bwd/Adam
    images, poses, focal = mini_scene(4, 16, 16, seed=3)
    rec = {"inds": [], "u": [], "loss": [], "psnr": [], "comp0": None, "gnorm0": None, "grads0": None}

    def record(step, inds, u, ro, rd, target, comp, loss, psnr, grads):
        rec["inds"].append(inds.clone()); rec["u"].append(u.clone())
        rec["loss"].append(loss.item()); rec["psnr"].append(psnr.item())
        if step == 0:
            rec["comp0"] = comp.detach().clone()
            rec["gnorm0"] = torch.stack([g.norm() for g in grads])
            rec["grads0"] = grads

    torch.manual_seed(4242)
    train_steps(model, enc, images, poses, focal, n_rand=256, n_samples=32, steps=10, lr=5e-4, record=record)
    final = flat_params(model)
    save(f"step_{tag}", images=images, poses=poses, focal=np.float64(focal),
         inds=torch.stack(rec["inds"]), u=torch.stack(rec["u"]),
         loss=np.array(rec["loss"], dtype=np.float64), psnr=np.array(rec["psnr"], dtype=np.float64),
         comp0=rec["comp0"], gnorm0=rec["gnorm0"],
         g0_first=rec["grads0"][0], g0_last_w=rec["grads0"][-2], g0_last_b=rec["grads0"][-1],
         final_sum=torch.stack([p.double().sum() for p in final]),
         final_abs=torch.stack([p.double().abs().sum() for p in final]),
         final_head=torch.cat([p.reshape(-1)[:64] for p in final]))


# ------------------------------------------------------------------ composite
def fx_composite():
    g = torch.Generator().manual_seed(404)
    out = {}
    for S in (64, 128):
        R = 96
        rgb = torch.rand(R, S, 3, generator=g)
        sigma = torch.relu(torch.randn(R, S, 1, generator=g) * 2.0)
        z = torch.sort(2.0 + 4.0 * torch.rand(R, S, generator=g), dim=-1).values
        rd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
        rd[48:96] *= torch.rand(48, 1, generator=g) * 3.0 + 0.2            # non-unit directions
        sigma[0:8] = 0.0                                                      # empty rays
        sigma[8:16] = 1e4                                                     # opaque at first sample
        sigma[16:24, -1] = 0.0                                                # sigma_last == 0
        sigma[24:32, -1] = 3.0                                                # sigma_last > 0
        sigma[32:40, :-1] = 0.0; sigma[32:40, -1] = 1e-12                     # tiny sigma against delta=1e10
        for white in (True, False):
            rgb_l = rgb.clone().requires_grad_(True); sig_l = sigma.clone().requires_grad_(True)
            comp, depth, acc, w = volume_render(rgb_l, sig_l, z, rd, white_bkgd=white)
            gC = torch.randn(R, 3, generator=torch.Generator().manual_seed(S + int(white)))
            (comp * gC).sum().backward()
            tag = f"{S}_{int(white)}"
            out.update({f"comp_{tag}": comp, f"depth_{tag}": depth, f"acc_{tag}": acc, f"w_{tag}": w,
                        f"gC_{tag}": gC, f"drgb_{tag}": rgb_l.grad, f"dsigma_{tag}": sig_l.grad})
        # all four outputs receive gradient
        rgb_l = rgb.clone().requires_grad_(True); sig_l = sigma.clone().requires_grad_(True)
        comp, depth, acc, w = volume_render(rgb_l, sig_l, z, rd, white_bkgd=True)
        gg = torch.Generator().manual_seed(900 + S)
        gC, gD, gA, gW = (torch.randn(comp.shape, generator=gg), torch.randn(depth.shape, generator=gg),
                          torch.randn(acc.shape, generator=gg), torch.randn(w.shape, generator=gg))
        ((comp * gC).sum() + (depth * gD).sum() + (acc * gA).sum() + (w * gW).sum()).backward()
        out.update({f"all_gC_{S}": gC, f"all_gD_{S}": gD, f"all_gA_{S}": gA, f"all_gW_{S}": gW,
                    f"all_drgb_{S}": rgb_l.grad, f"all_dsigma_{S}": sig_l.grad})
        out.update({f"rgb_{S}": rgb, f"sigma_{S}": sigma, f"z_{S}": z, f"rd_{S}": rd})
    out["psnr_in"] = torch.tensor([1e-12, 1e-10, 3e-4, 0.02, 0.5, 1.0])
    out["psnr_out"] = mse2psnr(out["psnr_in"])
    save("composite", **out)


def fx_spiral():
    ref = poses3()[1]
    save("spiral", ref=ref, poses60=camera.spiral_poses(ref), poses7=camera.spiral_poses(ref, n_frames=7, radius=0.5))


if __name__ == "__main__":
    print("torch", torch.__version__, "reference at", REF)
    fx_rays(); fx_sampling(); fx_encoding(); fx_composite(); fx_spiral()
    fx_model("4x128", 10, 128, 4, 2)
    fx_model("8x256", 6, 256, 8, 4)
