"""Pins the CPU oracle (oracle/tnerf_oracle.py) to fixtures produced by the reference itself
(tests/golden/make_golden.py).  Runs without a GPU."""
import math

import pytest
import torch

from conftest import load_golden, golden_params
from oracle import tnerf_oracle as O

torch.set_num_threads(max(1, min(8, torch.get_num_threads())))


def test_rays_match_reference_bitwise():
    g = load_golden("rays")
    for pi in range(3):
        ro, rd = O.pinhole_rays(5, 7, g["focal"], g["poses"][pi])
        assert torch.equal(ro, g[f"o_5x7_{pi}"]) and torch.equal(rd, g[f"d_5x7_{pi}"])
        ro, rd = O.pinhole_rays(100, 100, g["focal"], g["poses"][pi])
        idx = g[f"idx_100_{pi}"]
        assert torch.equal(ro[idx], g[f"o_100_{pi}"]) and torch.equal(rd[idx], g[f"d_100_{pi}"])


@pytest.mark.parametrize("S", [64, 128, 256])
def test_sample_bins_bitwise(S):
    g = load_golden("sampling")
    assert torch.equal(O.depth_bins(2.0, 6.0, S), g[f"zbase_{S}"])
    z, pts = O.stratified(2.0, 6.0, S, g["rays_o"], g["rays_d"], None)
    assert torch.equal(z[:4], g[f"z_det_{S}"]) and torch.equal(pts[:4], g[f"pts_det_{S}"])
    z, pts = O.stratified(2.0, 6.0, S, g["rays_o"], g["rays_d"], g[f"u_{S}"])
    assert torch.equal(z, g[f"z_rand_{S}"])
    n = g[f"pts_rand_{S}"].shape[0]
    assert torch.equal(pts[:n], g[f"pts_rand_{S}"])


def test_sample_bins_odd_range():
    g = load_golden("sampling")
    z, pts = O.stratified(0.5, 3.25, 64, g["rays_o"], g["rays_d"], g["u_odd"])
    assert torch.equal(z, g["z_odd"]) and torch.equal(pts[:32], g["pts_odd"])


@pytest.mark.parametrize("L,inc", [(6, True), (6, False), (10, True), (10, False)])
def test_encoding(L, inc):
    g = load_golden("encoding")
    want = g[f"enc_L{L}_{int(inc)}"]
    got = O.posenc(g["x"], L, inc)[: want.shape[0]]
    assert got.shape[-1] == O.posenc_dim(L, inc)
    assert torch.equal(got, want)


def test_encoding_rejects_non_xyz():
    with pytest.raises(AssertionError):
        O.posenc(torch.zeros(4, 2), 6)


@pytest.mark.parametrize("tag", ["4x128", "8x256"])
def test_mlp_forward_backward(tag):
    cfg, params = golden_params(tag)
    g = load_golden(f"mlp_{tag}")
    assert [tuple(p.shape) for p in params] == O.mlp_shapes(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"])
    leaves = [p.clone().requires_grad_(True) for p in params]
    rgb, sigma = O.mlp_forward(leaves, g["x"], cfg["skip_at"])
    torch.testing.assert_close(rgb, g["rgb"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sigma, g["sigma"], rtol=1e-5, atol=1e-6)
    ((rgb * g["g_rgb"]).sum() + (sigma * g["g_sigma"]).sum()).backward()
    for i, p in enumerate(leaves):
        want = g[f"g{i:02d}"]
        torch.testing.assert_close(p.grad, want, rtol=1e-4, atol=1e-5 * float(want.abs().max()))


@pytest.mark.parametrize("S", [64, 128])
@pytest.mark.parametrize("white", [True, False])
def test_composite_fwd_bwd(S, white):
    g = load_golden("composite")
    rgb = g[f"rgb_{S}"].clone().requires_grad_(True)
    sig = g[f"sigma_{S}"].clone().requires_grad_(True)
    tag = f"{S}_{int(white)}"
    comp, depth, acc, w = O.composite(rgb, sig, g[f"z_{S}"], g[f"rd_{S}"], white)
    for got, key in ((comp, "comp"), (depth, "depth"), (acc, "acc"), (w, "w")):
        assert torch.equal(got, g[f"{key}_{tag}"]), key
    (comp * g[f"gC_{tag}"]).sum().backward()
    assert torch.equal(rgb.grad, g[f"drgb_{tag}"])
    assert torch.equal(sig.grad, g[f"dsigma_{tag}"])


def test_psnr():
    g = load_golden("composite")
    assert torch.equal(O.psnr_from_mse(g["psnr_in"]), g["psnr_out"])


@pytest.mark.parametrize("tag", ["4x128", "8x256"])
def test_render_image(tag):
    cfg, params = golden_params(tag)
    g = load_golden(f"render_{tag}")
    for chunk, key in ((8192, "img"), (777, "img")):
        img = O.render_image(params, cfg["skip_at"], cfg["L"], g["H"], g["W"], g["focal"], g["pose"], 64, 2.0, 6.0, chunk)
        assert float((img - g[key]).abs().max()) <= 1e-5
    mse = torch.mean((img - g["img"]) ** 2)
    assert float(O.psnr_from_mse(mse)) >= 99.0


@pytest.mark.parametrize("tag", ["4x128", "8x256"])
def test_train_steps_follow_reference(tag):
    """10 Adam steps on the mini scene with the recorded index / jitter draws: loss trajectory,
    first-step gradients and final weights follow the reference."""
    cfg, params = golden_params(tag)
    g = load_golden(f"step_{tag}")
    images, poses, focal = g["images"], g["poses"], g["focal"]
    N, H, W, _ = images.shape
    rays = [O.pinhole_rays(H, W, focal, poses[i]) for i in range(N)]
    all_o = torch.stack([r[0] for r in rays]); all_d = torch.stack([r[1] for r in rays])
    pixels = images.reshape(N, H * W, 3)
    params = [p.clone() for p in params]
    opt = O.AdamState(params, lr=5e-4)
    for step in range(10):
        i = step % N
        inds = g["inds"][step]
        loss, psnr, grads = O.loss_and_grads(params, cfg["skip_at"], cfg["L"], all_o[i, inds], all_d[i, inds],
                                             pixels[i, inds], 2.0, 6.0, g["u"].shape[-1], g["u"][step])
        assert math.isclose(float(loss), g["loss"][step], rel_tol=2e-4), step
        assert abs(float(psnr) - g["psnr"][step]) < 2e-3
        if step == 0:
            gn = torch.stack([x.norm() for x in grads])
            torch.testing.assert_close(gn, g["gnorm0"], rtol=1e-4, atol=1e-9)
            torch.testing.assert_close(grads[0], g["g0_first"], rtol=1e-3, atol=1e-6 * float(g["g0_first"].abs().max()) + 1e-9)
            torch.testing.assert_close(grads[-2], g["g0_last_w"], rtol=1e-3, atol=1e-7)
        opt.step(params, grads)
    head = torch.cat([p.reshape(-1)[:64] for p in params])
    torch.testing.assert_close(head, g["final_head"], rtol=0, atol=2e-5)
    sums = torch.stack([p.double().sum() for p in params])
    torch.testing.assert_close(sums, g["final_sum"], rtol=0, atol=5e-3)


# ---- round 2: per-ray near/far and the novel-view path (fixtures: tests/golden/make_golden_r2.py)
def test_sample_bins_per_ray_near_far_bitwise():
    g = load_golden("sampling_per_ray")
    for S in (64, 33):
        z, pts = O.stratified(g["near"], g["far"], S, g["rays_o"], g["rays_d"], None)
        assert torch.equal(z, g[f"z_det_{S}"]) and torch.equal(pts[:16], g[f"pts_det_{S}"])
        z, pts = O.stratified(g["near"], g["far"], S, g["rays_o"], g["rays_d"], g[f"u_{S}"])
        assert torch.equal(z, g[f"z_rand_{S}"]) and torch.equal(pts[:16], g[f"pts_rand_{S}"])
    z, pts = O.stratified(torch.tensor(g["near0"]), g["far0"], 64, g["rays_o"], g["rays_d"], g["u_mixed"])
    assert torch.equal(z, g["z_mixed"]) and torch.equal(pts[:16], g["pts_mixed"])


def test_spiral_poses_match_reference():
    g = load_golden("spiral")
    p60 = O.spiral_poses(g["ref"])
    assert p60.shape == (60, 4, 4) and torch.equal(p60, g["poses60"])
    assert torch.equal(O.spiral_poses(g["ref"], n_frames=7, radius=0.5), g["poses7"])
    # properties of the path: rotation untouched, translation offsets of length `radius` in the camera's xy plane,
    # closed loop (linspace includes 2 pi)
    ref = g["ref"]
    assert torch.equal(p60[:, :3, :3], ref[:3, :3].expand(60, 3, 3)) and torch.equal(p60[:, 3], ref[3].expand(60, 4))
    off = (p60[:, :3, 3] - ref[:3, 3]) @ ref[:3, :3]               # back into the camera frame
    assert float((off.norm(dim=-1) - 0.3).abs().max()) < 1e-6 and float(off[:, 2].abs().max()) < 1e-6
    assert float((p60[0] - p60[-1]).abs().max()) < 1e-6
    assert O.deal_frames(7, 1, 3) == [1, 4] and sorted(sum((O.deal_frames(60, r, 8) for r in range(8)), [])) == list(range(60))


def test_novel_view_frames_match_reference():
    g = load_golden("novel_views")
    cfg, params = golden_params("4x128")
    path = O.spiral_poses(g["ref"])
    assert torch.equal(path, g["path"])
    H, W = int(g["H"]), int(g["W"])
    for j, k in enumerate(g["frame_index"].tolist()):
        img = O.render_image(params, cfg["skip_at"], cfg["L"], H, W, g["focal"], path[k], 64, 2.0, 6.0)
        assert float((img - g["frames"][j]).abs().max()) <= 1e-6
        u8 = (img.numpy() * 255).astype("uint8")                       # make_gif.py:26
        assert int((torch.from_numpy(u8).int() - g["frames_u8"][j].int()).abs().max()) <= 1
