"""Parity of the HIP path (through the C ABI, via the drop-in modules) against the CPU oracle and the
reference-generated golden fixtures.  Needs a MI355X: `pytest -m gpu`.

Tolerances (BASELINE.json north_star): ray indices / sample bins bit-exact; rendered RGB and PSNR within
1e-4 (fp32).  Gradients: relative to the largest entry of each tensor (GEMM summation order differs)."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, golden_params
from oracle import tnerf_oracle as O

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def mods():
    import rays, sampling, encoding, nerf, volume, utils, train   # drop-in call surface (tiny-nerf-pytorch_amd/src)
    from tnerf import ops, trainer, lib
    lib.load()
    return dict(rays=rays, sampling=sampling, encoding=encoding, nerf=nerf, volume=volume, utils=utils, train=train,
                ops=ops, trainer=trainer, lib=lib)


def make_model(mods, cfg, params, dev):
    m = mods["nerf"].TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"]).to(dev)
    with torch.no_grad():
        for p, v in zip(m.parameters(), params):
            p.copy_(v.to(dev))
    return m


def relmax(a, b):
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)


# ------------------------------------------------------------------------------------- rays
def test_get_rays(mods, dev):
    g = load_golden("rays")
    exact = total = 0
    for pi in range(3):
        for (H, W, key) in ((5, 7, "5x7"), (100, 100, "100")):
            ro, rd = mods["rays"].get_rays(H, W, g["focal"], g["poses"][pi].to(dev))
            assert ro.shape == (H * W, 3) and rd.shape == (H * W, 3) and rd.dtype == torch.float32
            assert ro.stride(0) == 0                                      # expand view like the reference
            ro, rd = ro.cpu(), rd.cpu()
            if key == "100":
                idx = g[f"idx_100_{pi}"]; ro, rd = ro[idx], rd[idx]
            assert torch.equal(ro, g[f"o_{key}_{pi}"])
            want = g[f"d_{key}_{pi}"]
            assert float((rd - want).abs().max()) <= 1.2e-7
            exact += int((rd == want).sum()); total += want.numel()
    assert exact / total > 0.9, f"only {exact}/{total} direction components bit-identical"


# --------------------------------------------------------------------------------- sampling
@pytest.mark.parametrize("S", [64, 128, 256])
def test_sample_bins_bit_exact(mods, dev, S):
    g = load_golden("sampling")
    ro, rd = g["rays_o"].to(dev), g["rays_d"].to(dev)
    z, pts, _ = mods["ops"].sample_along_rays(2.0, 6.0, S, ro, rd, False)
    assert z.stride(0) == 0
    assert torch.equal(z[:4].cpu(), g[f"z_det_{S}"]) and torch.equal(pts[:4].cpu(), g[f"pts_det_{S}"])
    z, pts, _ = mods["ops"].sample_along_rays(2.0, 6.0, S, ro, rd, True, t_rand=g[f"u_{S}"].to(dev))
    assert torch.equal(z.cpu(), g[f"z_rand_{S}"])
    n = g[f"pts_rand_{S}"].shape[0]
    assert torch.equal(pts[:n].cpu(), g[f"pts_rand_{S}"])


def test_sample_bins_odd_range_and_dropin_signature(mods, dev):
    g = load_golden("sampling")
    ro, rd = g["rays_o"].to(dev), g["rays_d"].to(dev)
    z, pts, _ = mods["ops"].sample_along_rays(0.5, 3.25, 64, ro, rd, True, t_rand=g["u_odd"].to(dev))
    assert torch.equal(z.cpu(), g["z_odd"]) and torch.equal(pts[:32].cpu(), g["pts_odd"])
    # drop-in: draws torch.rand on the device like the reference's rand_like
    torch.manual_seed(3)
    z1, p1 = mods["sampling"].stratified_samples(2.0, 6.0, 64, ro, rd, randomized=True)
    torch.manual_seed(3)
    u = torch.rand(ro.shape[0], 64, device=dev)
    zo, po = O.stratified(2.0, 6.0, 64, ro.cpu(), rd.cpu(), u.cpu())
    assert torch.equal(z1.cpu(), zo) and torch.equal(p1.cpu(), po)
    # in-kernel Philox jitter stays inside each bin
    z2, _, _ = mods["ops"].sample_along_rays(2.0, 6.0, 64, ro, rd, True, philox=(123, 0))
    zt = O.depth_bins(2.0, 6.0, 64)
    lo = torch.cat([zt[:1], 0.5 * (zt[1:] + zt[:-1])]); hi = torch.cat([0.5 * (zt[1:] + zt[:-1]), zt[-1:]])
    z2 = z2.cpu()
    assert bool(((z2 >= lo) & (z2 <= hi)).all()) and float(z2.std()) > 0.5
    assert not torch.equal(z2[0], z2[1])


# --------------------------------------------------------------------------------- encoding
@pytest.mark.parametrize("L,inc", [(6, True), (6, False), (10, True), (10, False)])
def test_encoding(mods, dev, L, inc):
    g = load_golden("encoding")
    enc = mods["encoding"].PositionalEncoding(L, inc).to(dev)
    assert enc.out_dim == O.posenc_dim(L, inc) and tuple(enc.freq_bands.shape) == (L,)
    want = g[f"enc_L{L}_{int(inc)}"]
    got = enc(g["x"].to(dev))[: want.shape[0]].cpu()
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= 2.5e-7              # sin/cos within ~2 ulp of ATen's
    if inc:
        assert torch.equal(got[:, :3], want[:, :3])
    with pytest.raises(AssertionError):
        enc(torch.zeros(4, 2, device=dev))
    assert enc(g["x"][:10].to(dev).reshape(2, 5, 3)).shape == (2, 5, enc.out_dim)


# -------------------------------------------------------------------------------- composite
@pytest.mark.parametrize("S", [64, 128])
@pytest.mark.parametrize("white", [True, False])
def test_composite_fwd_bwd(mods, dev, S, white):
    g = load_golden("composite")
    tag = f"{S}_{int(white)}"
    rgb = g[f"rgb_{S}"].to(dev).requires_grad_(True)
    sig = g[f"sigma_{S}"].to(dev).requires_grad_(True)
    comp, depth, acc, w = mods["volume"].volume_render(rgb, sig, g[f"z_{S}"].to(dev), g[f"rd_{S}"].to(dev), white_bkgd=white)
    assert comp.shape == (96, 3) and depth.shape == (96, 1) and acc.shape == (96, 1) and w.shape == (96, S)
    for got, key, tol in ((comp, "comp", 2e-6), (acc, "acc", 2e-6), (w, "w", 1e-6), (depth, "depth", 2e-5)):
        assert float((got.detach().cpu() - g[f"{key}_{tag}"]).abs().max()) <= tol, key
    (comp * g[f"gC_{tag}"].to(dev)).sum().backward()
    assert relmax(rgb.grad.cpu(), g[f"drgb_{tag}"]) <= 2e-6
    ds, want = sig.grad.cpu(), g[f"dsigma_{tag}"]
    assert ds.shape == want.shape
    _check_dsigma(ds, want, g, S, white, g[f"gC_{tag}"])


def _check_dsigma(ds, want, g, S, white, gC, extra=None):
    """d sigma spans 20 orders of magnitude (the 1e10 tail sample) and is a difference of two nearly
    equal terms, so fp32 results are judged against an fp64 evaluation of the same formula: the HIP
    kernel must be as close to it as the reference's own fp32 result is (x4 + a per-ray floor)."""
    rgb64 = g[f"rgb_{S}"].double(); sig64 = g[f"sigma_{S}"].double().requires_grad_(True)
    comp, depth, acc, w = O.composite(rgb64, sig64, g[f"z_{S}"].double(), g[f"rd_{S}"].double(), white)
    obj = (comp * gC.double()).sum()
    if extra is not None:
        obj = obj + (depth * extra[0].double()).sum() + (acc * extra[1].double()).sum() + (w * extra[2].double()).sum()
    obj.backward()
    ref = sig64.grad
    scale = ref.abs()[:, :-1].amax(dim=1, keepdim=True)                            # per ray, tail sample excluded
    scale = scale.clamp_min(1e-3 * float(scale.median()) + 1e-30)                  # rays whose gradient is ~0 (opaque / empty)
    e_hip = (ds.double() - ref).abs(); e_ref = (want.double() - ref).abs()
    body_hip, body_ref = (e_hip / scale)[:, :-1], (e_ref / scale)[:, :-1]
    assert float(body_hip.max()) <= 4.0 * float(body_ref.max()) + 2e-6, (float(body_hip.max()), float(body_ref.max()))
    tail_hip = e_hip[:, -1] / (ref[:, -1].abs() + 1e-30 + scale[:, 0])
    tail_ref = e_ref[:, -1] / (ref[:, -1].abs() + 1e-30 + scale[:, 0])
    assert float(tail_hip.max()) <= 4.0 * float(tail_ref.max()) + 1e-5, (float(tail_hip.max()), float(tail_ref.max()))
    assert bool(((want == 0) == (ds == 0)).all())                                   # ReLU / sigma==0 masks identical


def test_composite_all_outputs_receive_grad(mods, dev):
    g = load_golden("composite")
    S = 64
    rgb = g[f"rgb_{S}"].to(dev).requires_grad_(True); sig = g[f"sigma_{S}"].to(dev).requires_grad_(True)
    comp, depth, acc, w = mods["volume"].volume_render(rgb, sig, g[f"z_{S}"].to(dev), g[f"rd_{S}"].to(dev))
    ((comp * g[f"all_gC_{S}"].to(dev)).sum() + (depth * g[f"all_gD_{S}"].to(dev)).sum()
     + (acc * g[f"all_gA_{S}"].to(dev)).sum() + (w * g[f"all_gW_{S}"].to(dev)).sum()).backward()
    assert relmax(rgb.grad.cpu(), g[f"all_drgb_{S}"]) <= 2e-6
    _check_dsigma(sig.grad.cpu(), g[f"all_dsigma_{S}"], g, S, True, g[f"all_gC_{S}"],
                  extra=(g[f"all_gD_{S}"], g[f"all_gA_{S}"], g[f"all_gW_{S}"]))


def test_psnr(mods, dev):
    g = load_golden("composite")
    assert torch.allclose(mods["utils"].mse2psnr(g["psnr_in"].to(dev)).cpu(), g["psnr_out"], rtol=0, atol=1e-5)


# -------------------------------------------------------------------------------------- MLP
@pytest.mark.parametrize("pipe", ["x3", "fp32_mfma"])
@pytest.mark.parametrize("tag", ["4x128", "8x256"])
def test_mlp_forward_backward(mods, dev, tag, pipe):
    """The same gates for both matrix pipes: the default x3 scheme (three fp16 partial products) and plain fp32 MFMA."""
    cfg, params = golden_params(tag)
    g = load_golden(f"mlp_{tag}")
    model = mods["nerf"].TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"], matrix_pipe=pipe).to(dev)
    with torch.no_grad():
        for p_, v_ in zip(model.parameters(), params):
            p_.copy_(v_.to(dev))
    assert list(model.state_dict().keys())[:2] == ["layers.0.weight", "layers.0.bias"]
    assert "sigma.0.weight" in model.state_dict() and "rgb.0.bias" in model.state_dict()
    rgb, sigma = model(g["x"].to(dev))
    assert rgb.shape == g["rgb"].shape and sigma.shape == g["sigma"].shape
    assert float((rgb.detach().cpu() - g["rgb"]).abs().max()) <= 2e-6
    assert float((sigma.detach().cpu() - g["sigma"]).abs().max()) <= 1e-5 * max(1.0, float(g["sigma"].abs().max()))
    ((rgb * g["g_rgb"].to(dev)).sum() + (sigma * g["g_sigma"].to(dev)).sum()).backward()
    # two fp32 evaluations (the reference's CPU autograd in the fixture, the HIP kernels) judged against an fp64 evaluation:
    # HIP must be as close to it as the reference is (a ReLU input within rounding of 0 lands on either side)
    leaves = [p.double().requires_grad_(True) for p in params]
    r64, s64 = O.mlp_forward(leaves, g["x"].double(), cfg["skip_at"])
    g64 = torch.autograd.grad((r64 * g["g_rgb"].double()).sum() + (s64 * g["g_sigma"].double()).sum(), leaves)
    # Whole gradient vector AND every tensor: as close to fp64 as the reference's own fp32 evaluation (x2), and every element within
    # 2e-5 of the reference's value (SURVEY 8c's mlp_* pin).  Round 2's split-bf16 chain needed 16x / 1e-4 here: its accumulators
    # were biased by the matrix pipe's floor cut; the fp16 chain with split accumulators is not.
    flat = lambda ts: torch.cat([t.reshape(-1).double() for t in ts])
    gh_all, gr_all, gd_all = flat([p.grad.cpu() for p in model.parameters()]), flat([g[f"g{i:02d}"] for i in range(len(params))]), flat(g64)
    l2_hip, l2_ref = float((gh_all - gd_all).norm() / gd_all.norm()), float((gr_all - gd_all).norm() / gd_all.norm())
    print(f"[{tag}] MLP grads vs fp64, whole vector L2: hip {l2_hip:.2e} reference-fp32 {l2_ref:.2e}")
    assert l2_hip <= 2.0 * l2_ref + 1e-7, (l2_hip, l2_ref)
    for i, p in enumerate(model.parameters()):
        assert p.grad is not None
        gh, gr, gd = p.grad.cpu().double(), g[f"g{i:02d}"].double(), g64[i]
        t_hip, t_ref = float((gh - gd).norm() / gd.norm()), float((gr - gd).norm() / gd.norm())
        assert t_hip <= 2.0 * t_ref + 2e-7, (i, t_hip, t_ref)
        # two fp32 evaluations cannot agree better than the reference itself agrees with fp64 (rgb.0.weight of the 8x256 fixture:
        # the reference's own value is 4e-3 from fp64)
        assert relmax(p.grad.cpu(), g[f"g{i:02d}"]) <= max(2e-5, 2.0 * relmax(gr, gd)), (i, relmax(p.grad.cpu(), g[f"g{i:02d}"]), relmax(gr, gd))
    # ragged row count (not a multiple of 32) and no-grad inference agree with the training forward
    with torch.no_grad():
        r2, s2 = model(g["x"][:1001].to(dev))
    assert torch.equal(r2, rgb[:1001].detach()) and torch.equal(s2, sigma[:1001].detach())


def test_mlp_rejects_what_it_cannot_do(mods, dev):
    m = mods["nerf"].TinyNeRF(39, 256, 8, 4)
    with pytest.raises(RuntimeError):
        m(torch.zeros(4, 39))                                        # CPU: no fallback
    m = m.to(dev)
    with pytest.raises(RuntimeError):
        m(torch.zeros(4, 40, device=dev))
    wide = mods["nerf"].TinyNeRF(39, 512, 8, 4).to(dev)
    assert wide(torch.zeros(4, 39, device=dev))[0].shape == (4, 3)     # widths above 256: the layer-by-layer path (test below) ...
    with pytest.raises(NotImplementedError):
        wide.hip_state()                                             # ... but no fused kernels, no fused trainer


@pytest.mark.parametrize("arch", [(39, 512, 4, 2), (129, 300, 3, 1), (63, 384, 3, 0), (39, 256, 8, 4), (63, 128, 4, 2)])
def test_any_width_and_input_gradient_layer_by_layer(mods, dev, arch):
    """TinyNeRF(in_dim, hidden, ...) beyond what the chain kernels cover (hidden > 256, in_dim > 64; reference src/nerf.py:10 takes
    any) and, for every shape, the gradient w.r.t. the encoded input (src/nerf.py:29-41 is differentiable in x): tnerf_mlp_fwd_generic
    / tnerf_mlp_bwd_generic against the oracle's autograd; deterministic."""
    in_dim, hidden, depth, skip = arch
    torch.manual_seed(3)
    model = mods["nerf"].TinyNeRF(in_dim, hidden, depth, skip).to(dev)
    with torch.no_grad():
        model.sigma[0].bias += 0.5
    params = [p.detach().cpu().clone() for p in model.parameters()]
    g = torch.Generator().manual_seed(4)
    M = 1111
    x = torch.randn(M, in_dim, generator=g)
    gr, gs = torch.randn(M, 3, generator=g) * 0.1, torch.randn(M, 1, generator=g) * 0.1
    leaves = [p.clone().requires_grad_(True) for p in params]
    xo = x.clone().requires_grad_(True)
    ro, so = O.mlp_forward(leaves, xo, skip)
    go = torch.autograd.grad((ro * gr).sum() + (so * gs).sum(), leaves + [xo])

    def run():
        for p in model.parameters():
            p.grad = None
        xd = x.to(dev).requires_grad_(True)
        rgb, sig = model(xd)
        ((rgb * gr.to(dev)).sum() + (sig * gs.to(dev)).sum()).backward()
        return rgb.detach(), sig.detach(), [p.grad.clone() for p in model.parameters()], xd.grad.clone()

    rgb, sig, pg, dx = run()
    assert float((rgb.cpu() - ro.detach()).abs().max()) <= 2e-6
    assert float((sig.cpu() - so.detach()).abs().max()) <= 1e-5 * max(1.0, float(so.abs().max()))
    # gradients: as close to an fp64 evaluation as the reference's own fp32 evaluation is (x2; deep random-init nets flip ReLUs of
    # samples that sit within rounding of 0, which moves single fp32 evaluations apart by far more than 2e-5)
    l64 = [p.double().requires_grad_(True) for p in params]
    x64 = x.double().requires_grad_(True)
    r64, s64 = O.mlp_forward(l64, x64, skip)
    g64 = torch.autograd.grad((r64 * gr.double()).sum() + (s64 * gs.double()).sum(), l64 + [x64])
    for i, (a, b, c) in enumerate(zip(pg + [dx], go, g64)):
        t_hip, t_ref = relmax(a.cpu().double(), c), relmax(b.double(), c)
        assert t_hip <= 2.0 * t_ref + 2e-5, (i, t_hip, t_ref)      # + the 2e-5 gate of the fixture tests: bias gradients are cancelling sums of M terms
    rgb2, sig2, pg2, dx2 = run()
    assert torch.equal(rgb, rgb2) and torch.equal(dx, dx2) and all(torch.equal(a, b) for a, b in zip(pg, pg2))
    if hidden <= 256 and in_dim <= 64:
        # the same model without an input gradient runs on the chain kernels: same function
        with torch.no_grad():
            r3, s3 = model(x.to(dev))
        assert float((r3 - rgb).abs().max()) <= 2e-6 and float((s3 - sig).abs().max()) <= 1e-5 * max(1.0, float(sig.abs().max()))
    else:
        if (in_dim - 3) % 6 == 0:
            # render_one takes such a model through the per-function kernels (rays, bins, encoding, layer-by-layer MLP, compositing)
            L = (in_dim - 3) // 6
            enc = mods["encoding"].PositionalEncoding(L, True).to(dev)
            pose = torch.eye(4); pose[2, 3] = 4.0
            img = mods["train"].render_one(model, enc, 12, 10, 15.0, pose, dev, n_samples=24, near=2.0, far=6.0, chunk=50)
            ro_, rd_ = O.pinhole_rays(12, 10, 15.0, pose)
            co = O.render_rays(params, skip, L, ro_, rd_, 2.0, 6.0, 24, None)[0]
            assert float((img.cpu().reshape(-1, 3) - co.clamp(0, 1)).abs().max()) <= RGB_TOL
        # a plain torch optimizer trains it (no flat buffer behind these parameters)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        tgt = torch.rand(M, 3, generator=g).to(dev)
        l0 = None
        for _ in range(20):
            opt.zero_grad()
            loss = ((model(x.to(dev))[0] - tgt) ** 2).mean()
            loss.backward(); opt.step()
            l0 = float(loss) if l0 is None else l0
        assert float(loss) < l0


# -------------------------------------------------------------------------- fused render / train
@pytest.mark.parametrize("tag", ["4x128", "8x256"])
def test_render_one_matches_reference_image(mods, dev, tag):
    cfg, params = golden_params(tag)
    g = load_golden(f"render_{tag}")
    model = make_model(mods, cfg, params, dev)
    enc = mods["encoding"].PositionalEncoding(cfg["L"], True).to(dev)
    imgs = []
    for chunk in (8192, 1000, 77):
        img = mods["train"].render_one(model, enc, g["H"], g["W"], g["focal"], g["pose"], dev, 64, 2.0, 6.0, chunk).cpu()
        assert img.shape == g["img"].shape
        assert float((img - g["img"]).abs().max()) <= RGB_TOL
        imgs.append(img)
    assert torch.equal(imgs[0], imgs[1]) and torch.equal(imgs[0], imgs[2])        # chunk invariance, bitwise
    psnr = float(O.psnr_from_mse(torch.mean((imgs[0] - g["img"]) ** 2)))
    assert psnr >= 80.0
    # unfused composition of the per-function ops gives the same picture
    class Plain(torch.nn.Module):                                                 # not a PositionalEncoding instance
        def __init__(s, e): super().__init__(); s.e = e
        def forward(s, x): return s.e(x)
    img_u = mods["train"].render_one(model, Plain(enc), g["H"], g["W"], g["focal"], g["pose"], dev, 64, 2.0, 6.0, 4096).cpu()
    assert float((img_u - imgs[0]).abs().max()) <= 2e-5


def _step_inputs(g, cfg, step):
    images, poses, focal = g["images"], g["poses"], g["focal"]
    N, H, W, _ = images.shape
    rays = [O.pinhole_rays(H, W, focal, poses[i]) for i in range(N)]
    i = step % N
    inds = g["inds"][step]
    return rays[i][0][inds].contiguous(), rays[i][1][inds].contiguous(), images.reshape(N, H * W, 3)[i, inds].contiguous()


@pytest.mark.parametrize("tag", ["4x128", "8x256"])
def test_train_gradients_match_oracle(mods, dev, tag):
    cfg, params = golden_params(tag)
    g = load_golden(f"step_{tag}")
    ro, rd, tgt = _step_inputs(g, cfg, 0)
    u = g["u"][0]
    S = u.shape[-1]
    loss_o, psnr_o, grads_o = O.loss_and_grads(params, cfg["skip_at"], cfg["L"], ro, rd, tgt, 2.0, 6.0, S, u)
    model = make_model(mods, cfg, params, dev)
    st, plist = model._ensure_packed(), model._param_list()
    comp, _, _ = mods["ops"].render_rays_fused(st, plist, ro.to(dev), rd.to(dev), 2.0, 6.0, S, True, t_rand=u.to(dev))
    assert float((comp.detach().cpu() - g["comp0"]).abs().max()) <= RGB_TOL
    loss = torch.mean((comp - tgt.to(dev)) ** 2)
    assert abs(float(loss.detach()) - g["loss"][0]) <= 1e-4 * g["loss"][0] + 1e-7
    assert abs(float(mods["utils"].mse2psnr(loss.detach())) - g["psnr"][0]) <= 1e-4 * abs(g["psnr"][0])
    loss.backward()
    # judge fp32 gradients against an fp64 evaluation: HIP must be as close to it as the fp32 oracle is
    p64 = [p.double() for p in params]
    _, _, g64 = O.loss_and_grads(p64, cfg["skip_at"], cfg["L"], ro.double(), rd.double(), tgt.double(), 2.0, 6.0, S, u.double())
    worst_hip = max(relmax(p.grad.cpu().double(), gg) for p, gg in zip(plist, g64))
    worst_cpu = max(relmax(go.double(), gg) for go, gg in zip(grads_o, g64))
    flat = lambda gs: torch.cat([x.reshape(-1).double() for x in gs])
    g_hip, g_cpu, g_ref = flat([p.grad.cpu() for p in plist]), flat(grads_o), flat(g64)
    l2_hip = float((g_hip - g_ref).norm() / g_ref.norm()); l2_cpu = float((g_cpu - g_ref).norm() / g_ref.norm())
    print(f"[{tag}] grad err vs fp64: L2 hip {l2_hip:.2e} cpu-fp32 {l2_cpu:.2e} | worst element hip {worst_hip:.2e} cpu-fp32 {worst_cpu:.2e}")
    assert l2_hip <= 2.0 * l2_cpu + 1e-6, (l2_hip, l2_cpu)
    assert worst_hip <= 2.0 * worst_cpu + 1e-5, (worst_hip, worst_cpu)
    # (no direct fp32-vs-fp32 bound: both evaluations sit ~2e-3 from fp64 in the worst element, the yardstick gates above are the
    #  statement; the fixture's recorded norms pin the ORACLE in tests/test_oracle_golden.py)
    for p, gg, go in zip(plist, g64, grads_o):                  # ... and per tensor, as for the MLP fixture
        t_hip, t_cpu = float((p.grad.cpu().double() - gg).norm() / gg.norm()), float((go.double() - gg).norm() / gg.norm())
        assert t_hip <= 2.0 * t_cpu + 2e-6, (t_hip, t_cpu)
    # unfused autograd path (per-function HIP ops) gives the same gradients
    m2 = make_model(mods, cfg, params, dev)
    enc = mods["encoding"].PositionalEncoding(cfg["L"], True).to(dev)
    z, pts, _ = mods["ops"].sample_along_rays(2.0, 6.0, S, ro.to(dev), rd.to(dev), True, t_rand=u.to(dev))
    rgb, sig = m2(enc(pts.reshape(-1, 3)))
    c2, _, _, _ = mods["volume"].volume_render(rgb.reshape(-1, S, 3), sig.reshape(-1, S, 1), z, rd.to(dev))
    assert float((c2 - comp).abs().max()) <= 2e-5
    torch.mean((c2 - tgt.to(dev)) ** 2).backward()
    # (both run the x3 chain kernels; the per-function path sums the exact L1 norm of the encoded input where the fused kernel
    # uses its bound 6L + 3 max|p|, so the per-sample scales — and with them roundings at the 1e-7 level — differ)
    g_unf = flat([q.grad.cpu() for q in m2.parameters()])
    l2_unf = float((g_unf - g_ref).norm() / g_ref.norm())
    assert l2_unf <= 2.0 * l2_cpu + 1e-6, (l2_unf, l2_cpu)
    assert float((g_unf - g_hip).norm() / g_ref.norm()) <= 2.0 * l2_cpu + 1e-6
    for p, q in zip(plist, m2.parameters()):
        assert relmax(q.grad, p.grad) <= 2e-4


@pytest.mark.parametrize("tag", ["4x128", "8x256"])
def test_ten_fused_steps_follow_reference_trajectory(mods, dev, tag):
    cfg, params = golden_params(tag)
    g = load_golden(f"step_{tag}")
    model = make_model(mods, cfg, params, dev)
    S = g["u"].shape[-1]
    opt = mods["trainer"].FlatAdam(model, lr=5e-4)
    tr = mods["trainer"].FusedTrainer(model, opt, 2.0, 6.0, S)
    for step in range(10):
        ro, rd, tgt = _step_inputs(g, cfg, step)
        loss, _ = tr.step(ro.to(dev), rd.to(dev), tgt.to(dev), t_rand=g["u"][step].to(dev))
        assert math.isclose(float(loss), g["loss"][step], rel_tol=3e-4), (step, float(loss), g["loss"][step])
    final = [p.detach().cpu() for p in model.parameters()]
    head = torch.cat([p.reshape(-1)[:64] for p in final])
    torch.testing.assert_close(head, g["final_head"], rtol=0, atol=5e-5)
    sums = torch.stack([p.double().sum() for p in final])
    torch.testing.assert_close(sums, g["final_sum"], rtol=0, atol=1e-2)
    sd = opt.state_dict()                                     # torch.optim.Adam-shaped state
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 10.0


def test_adam_kernel_matches_oracle(mods, dev):
    cfg, params = golden_params("4x128")
    model = make_model(mods, cfg, params, dev)
    opt = mods["trainer"].FlatAdam(model, lr=5e-4)
    ref = [p.clone() for p in params]
    ost = O.AdamState(ref, lr=5e-4)
    gen = torch.Generator().manual_seed(0)
    for _ in range(5):
        grads = [torch.randn(p.shape, generator=gen) * 1e-2 for p in ref]
        for p, gg in zip(model.parameters(), grads):
            p.grad = gg.to(dev)
        opt.step()
        ost.step(ref, grads)
    for p, r in zip(model.parameters(), ref):
        assert float((p.detach().cpu() - r).abs().max()) <= 1e-7


# ------------------------------------------------------ BASELINE-size properties (no oracle at this size)
def test_full_size_properties(mods, dev):
    """cfg 2 shapes: 4096 rays x 64 samples, L=6, 8x256.  Determinism (bitwise), shard additivity of the
    gradient (what the multi-GPU all-reduce relies on), linearity in dL/dcomp, Philox == explicit jitter."""
    torch.manual_seed(0)
    enc_L, R, S = 6, 4096, 64
    model = mods["nerf"].TinyNeRF(39, 256, 8, 4).to(dev)
    with torch.no_grad():
        model.sigma[0].bias += 0.5
    st, plist = model._ensure_packed(), model._param_list()
    gen = torch.Generator().manual_seed(1)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
    o = (-4.0 * d + 0.3 * torch.randn(R, 3, generator=gen)).to(dev); d = d.to(dev)
    tgt = torch.rand(R, 3, generator=gen).to(dev); u = torch.rand(R, S, generator=gen).to(dev)

    def grads(lo, hi, scale=1.0):
        comp, _, _ = mods["ops"].render_rays_fused(st, plist, o[lo:hi], d[lo:hi], 2.0, 6.0, S, True, t_rand=u[lo:hi])
        loss = scale * ((comp - tgt[lo:hi]) ** 2).sum() / (3.0 * R)
        gs = torch.autograd.grad(loss, plist)
        return comp.detach(), torch.cat([x.reshape(-1) for x in gs])

    c1, g1 = grads(0, R); c2, g2 = grads(0, R)
    assert torch.equal(c1, c2) and torch.equal(g1, g2)                           # deterministic, bit for bit
    assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0
    ca, ga = grads(0, R // 2); cb, gb = grads(R // 2, R)
    assert torch.equal(torch.cat([ca, cb]), c1)                                  # rays are independent
    assert float((ga + gb - g1).abs().max()) <= 2e-5 * float(g1.abs().max())    # shard gradients add up
    _, g3 = grads(0, R, scale=3.0)
    assert float((g3 - 3.0 * g1).abs().max()) <= 2e-5 * float(g3.abs().max())  # linear in the upstream gradient
    with torch.no_grad():
        cp, dp, ap = mods["ops"].render_rays_fused(st, plist, o, d, 2.0, 6.0, S, False)
        cq, _, _ = mods["ops"].render_rays_fused(st, plist, o, d, 2.0, 6.0, S, False, white_bkgd=False)
    assert float((cp - (cq + (1.0 - ap))).abs().max()) <= 1e-6                  # white background identity
    assert bool(((ap >= 0) & (ap <= 1.0 + 1e-5)).all()) and bool(((dp >= 0) & (dp <= 6.0 * 1.0001)).all())


# ------------------------------------------------------------- ragged / large-S / generic shapes
@pytest.mark.parametrize("tag,R,S", [("4x128", 37, 50), ("8x256", 37, 50), ("8x256", 130, 128), ("4x128", 66, 256), ("8x256", 5, 2), ("4x128", 3, 33)])
def test_fused_ragged_shapes_forward_and_gradients(mods, dev, tag, R, S):
    """Sample counts that are not multiples of 32 (padding lanes, dump block), ray counts that are not multiples
    of 4 (idle waves), several 64-sample segments per ray (transmittance carry): RGB/depth/acc and gradients."""
    cfg, params = golden_params(tag)
    gen = torch.Generator().manual_seed(R * 1000 + S)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=gen), dim=-1)
    o = -4.0 * d + 0.3 * torch.randn(R, 3, generator=gen)
    tgt, u = torch.rand(R, 3, generator=gen), torch.rand(R, S, generator=gen)
    model = make_model(mods, cfg, params, dev)
    st, plist = model._ensure_packed(), model._param_list()
    with torch.no_grad():
        c_inf, dep, acc = mods["ops"].render_rays_fused(st, plist, o.to(dev), d.to(dev), 2.0, 6.0, S, True, t_rand=u.to(dev))
    co, do_, ao, _ = O.render_rays(params, cfg["skip_at"], cfg["L"], o, d, 2.0, 6.0, S, u)
    assert float((c_inf.cpu() - co).abs().max()) <= RGB_TOL
    assert float((dep.cpu() - do_).abs().max()) <= 1e-3 and float((acc.cpu() - ao).abs().max()) <= RGB_TOL
    comp, _, _ = mods["ops"].render_rays_fused(st, plist, o.to(dev), d.to(dev), 2.0, 6.0, S, True, t_rand=u.to(dev))
    assert torch.equal(comp.detach(), c_inf)                       # training forward == inference forward, bitwise
    torch.mean((comp - tgt.to(dev)) ** 2).backward()
    _, _, g32 = O.loss_and_grads(params, cfg["skip_at"], cfg["L"], o, d, tgt, 2.0, 6.0, S, u)
    _, _, g64 = O.loss_and_grads([p.double() for p in params], cfg["skip_at"], cfg["L"], o.double(), d.double(), tgt.double(), 2.0, 6.0, S, u.double())
    # fp32 vs fp32, both judged against an fp64 evaluation.  The whole-vector L2 error must be within 2x of the CPU-fp32 oracle's,
    # and so must every tensor's worst element — except the sigma head's (sigma.0.weight / bias): the 1e10 tail sample makes
    # d sigma ill-conditioned (exp(-sigma 1e10) amplifies the forward's last-bit differences in a sigma near zero), so that
    # tensor's worst element is a draw from a wide distribution for ANY fp32 evaluation order (tests/probes/ragged_grad_probe.py on
    # MI355X: CPU 1.3e-3 / fp32-MFMA kernels 5.7e-3 / x3 kernels 9.1e-3 on one shape, 1.8e-2 / 5.6e-3 / 3.6e-2 on another, every
    # other tensor within 1.0-1.8x of the CPU's); it gets 8x.
    flat = lambda gs: torch.cat([x.reshape(-1).double() for x in gs])
    g_hip, g_cpu, g_ref = flat([p.grad.cpu() for p in plist]), flat(g32), flat(g64)
    l2_hip = float((g_hip - g_ref).norm() / g_ref.norm()); l2_cpu = float((g_cpu - g_ref).norm() / g_ref.norm())
    e_hip = [relmax(p.grad.cpu().double(), gg) for p, gg in zip(plist, g64)]
    e_cpu = [relmax(a.double(), gg) for a, gg in zip(g32, g64)]
    print(f"[{tag} R={R} S={S}] grad err vs fp64: L2 hip {l2_hip:.2e} cpu-fp32 {l2_cpu:.2e} | worst element hip {max(e_hip):.2e} cpu-fp32 {max(e_cpu):.2e}")
    assert l2_hip <= 2.0 * l2_cpu + 1e-6, (l2_hip, l2_cpu)
    sigma_head = (2 * cfg["depth"], 2 * cfg["depth"] + 1)
    for i, (eh, ec) in enumerate(zip(e_hip, e_cpu)):
        assert eh <= (8.0 if i in sigma_head else 2.0) * ec + 2e-5, (i, eh, ec)


def test_deterministic_render_of_large_sample_counts(mods, dev):
    """BASELINE cfg 3/5 shapes per ray (S=128, S=256) without jitter: fused == unfused per-function path."""
    cfg, params = golden_params("8x256")
    model = make_model(mods, cfg, params, dev)
    enc = mods["encoding"].PositionalEncoding(cfg["L"], True).to(dev)
    g = load_golden("render_8x256")
    ro, rd = mods["rays"].get_rays(40, 40, 55.0, g["pose"].to(dev))
    st, plist = model._ensure_packed(), model._param_list()
    for S in (128, 256):
        with torch.no_grad():
            cf, _, _ = mods["ops"].render_rays_fused(st, plist, ro, rd, 2.0, 6.0, S, False)
            z, pts = mods["sampling"].stratified_samples(2.0, 6.0, S, ro, rd, randomized=False)
            rgb, sig = model(enc(pts.reshape(-1, 3)))
            cu, _, _, _ = mods["volume"].volume_render(rgb.reshape(-1, S, 3), sig.reshape(-1, S, 1), z, rd)
        assert float((cf - cu).abs().max()) <= 2e-5
        co, _, _, _ = O.render_rays(params, cfg["skip_at"], cfg["L"], ro.cpu().contiguous(), rd.cpu(), 2.0, 6.0, S, None)
        assert float((cf.cpu() - co).abs().max()) <= RGB_TOL


@pytest.mark.parametrize("in_dim,hidden,depth,skip", [(10, 128, 2, 1), (40, 256, 3, 0), (64, 128, 3, 2), (3, 128, 1, 0)])
def test_mlp_generic_input_width(mods, dev, in_dim, hidden, depth, skip):
    """TinyNeRF on inputs that are not a positional encoding (generic column pairing), ragged row counts."""
    gen = torch.Generator().manual_seed(in_dim)
    params = O.mlp_init(in_dim, hidden, depth, skip, gen)
    params[2 * depth + 1] += 0.3
    cfg = dict(in_dim=in_dim, hidden=hidden, depth=depth, skip_at=skip)
    model = make_model(mods, cfg, params, dev)
    x = torch.randn(333, in_dim, generator=gen)
    rgb, sig = model(x.to(dev))
    ro, so = O.mlp_forward(params, x, skip)
    assert float((rgb.cpu() - ro).abs().max()) <= 2e-6 and float((sig.cpu() - so).abs().max()) <= 2e-5
    g1, g2 = torch.randn(333, 3, generator=gen), torch.randn(333, 1, generator=gen)
    ((rgb * g1.to(dev)).sum() + (sig * g2.to(dev)).sum()).backward()
    leaves = [p.clone().requires_grad_(True) for p in params]
    r2, s2 = O.mlp_forward(leaves, x, skip)
    ((r2 * g1).sum() + (s2 * g2).sum()).backward()
    for p, q in zip(model.parameters(), leaves):
        assert relmax(p.grad.cpu(), q.grad) <= 5e-5


def test_empty_batches(mods, dev):
    cfg, params = golden_params("4x128")
    model = make_model(mods, cfg, params, dev)
    st, plist = model._ensure_packed(), model._param_list()
    e3 = torch.zeros(0, 3, device=dev)
    with torch.no_grad():
        c, _, _ = mods["ops"].render_rays_fused(st, plist, e3, e3, 2.0, 6.0, 64, False)
        r, s = model(torch.zeros(0, cfg["in_dim"], device=dev))
    assert c.shape == (0, 3) and r.shape == (0, 3) and s.shape == (0, 1)
    z, pts, _ = mods["ops"].sample_along_rays(2.0, 6.0, 64, e3, e3, True, t_rand=torch.zeros(0, 64, device=dev))
    assert z.shape == (0, 64) and pts.shape == (0, 64, 3)


def test_checkpoint_roundtrip_with_torch_adam(mods, dev, tmp_path):
    """Checkpoint dict of the reference loop (train.py:143-148): our FlatAdam state loads into torch.optim.Adam
    (and back) and both continue identically."""
    cfg, params = golden_params("4x128")
    g = load_golden("step_4x128")
    S = g["u"].shape[-1]
    m1 = make_model(mods, cfg, params, dev)
    o1 = mods["trainer"].FlatAdam(m1, lr=5e-4)
    t1 = mods["trainer"].FusedTrainer(m1, o1, 2.0, 6.0, S)
    for step in range(3):
        ro, rd, tgt = _step_inputs(g, cfg, step)
        t1.step(ro.to(dev), rd.to(dev), tgt.to(dev), t_rand=g["u"][step].to(dev))
    path = str(tmp_path / "ck.pth")
    torch.save({"model": m1.state_dict(), "opt": o1.state_dict(), "step": 3, "in_dim": cfg["in_dim"],
                "cfg": dict(hidden=cfg["hidden"], depth=cfg["depth"], skip_at=cfg["skip_at"])}, path)
    ck = torch.load(path, map_location=dev)
    # (a) into a fresh HIP model + torch.optim.Adam, autograd path
    m2 = mods["nerf"].TinyNeRF(ck["in_dim"], **ck["cfg"]).to(dev)
    m2.load_state_dict(ck["model"])
    o2 = torch.optim.Adam(m2.parameters(), lr=5e-4)
    o2.load_state_dict(ck["opt"])
    # (b) into a fresh HIP model + FlatAdam
    m3 = mods["nerf"].TinyNeRF(ck["in_dim"], **ck["cfg"]).to(dev)
    m3.load_state_dict(ck["model"])
    o3 = mods["trainer"].FlatAdam(m3, lr=5e-4)
    o3.load_state_dict(ck["opt"])
    t3 = mods["trainer"].FusedTrainer(m3, o3, 2.0, 6.0, S)
    ro, rd, tgt = _step_inputs(g, cfg, 3)
    u = g["u"][3].to(dev)
    t1.step(ro.to(dev), rd.to(dev), tgt.to(dev), t_rand=u)
    t3.step(ro.to(dev), rd.to(dev), tgt.to(dev), t_rand=u)
    st2 = m2._ensure_packed()
    comp, _, _ = mods["ops"].render_rays_fused(st2, m2._param_list(), ro.to(dev), rd.to(dev), 2.0, 6.0, S, True, t_rand=u)
    o2.zero_grad(set_to_none=True)
    torch.mean((comp - tgt.to(dev)) ** 2).backward()
    o2.step()
    for a, b, c in zip(m1.parameters(), m2.parameters(), m3.parameters()):
        assert torch.equal(a, c)
        assert float((a - b).abs().max()) <= 1e-6


def test_camera_sourced_rays_equal_table_rays_bitwise(mods, dev):
    """rays generated inside the fused kernels from pose + pixel index give bit-identical
    images, losses and gradients to get_rays tables + gathers (the reference's plumbing, train.py:94-112)."""
    cfg, params = golden_params("8x256")
    g = load_golden("render_8x256")
    H, W, focal, pose = g["H"], g["W"], g["focal"], g["pose"].to(dev)
    model = make_model(mods, cfg, params, dev)
    st, plist = model._ensure_packed(), model._param_list()
    ro, rd = mods["rays"].get_rays(H, W, focal, pose)
    with torch.no_grad():
        a, da, aa = mods["ops"].render_rays_fused(st, plist, ro[1000:3000], rd[1000:3000], 2.0, 6.0, 64, False)
    b, db, ab = mods["ops"].render_camera_fused(st, pose, H, W, focal, 1000, 2000, 2.0, 6.0, 64)
    assert torch.equal(a, b) and torch.equal(da, db) and torch.equal(aa, ab)
    # train step: tables vs camera
    gen = torch.Generator().manual_seed(5)
    inds = torch.randint(0, H * W, (512,), generator=gen).to(dev)
    pix = torch.rand(H * W, 3, generator=gen).to(dev)
    u = torch.rand(512, 64, generator=gen).to(dev)
    res = []
    for mode in ("tables", "camera"):
        m = make_model(mods, cfg, params, dev)
        opt = mods["trainer"].FlatAdam(m, lr=5e-4)
        tr = mods["trainer"].FusedTrainer(m, opt, 2.0, 6.0, 64)
        if mode == "tables":
            loss, comp = tr.step(ro[inds], rd[inds], pix[inds], t_rand=u)
        else:
            loss, comp = tr.step_camera(pose, H, W, focal, inds, pix, t_rand=u)
        res.append((loss.clone(), comp.clone(), m.hip_state().grad.clone(), m.hip_state().flat.clone()))
    for x, y in zip(res[0], res[1]):
        assert torch.equal(x, y)
    with pytest.raises(ValueError):
        mods["ops"].render_camera_fused(st, pose, H, W, focal, H * W - 10, 11, 2.0, 6.0, 64)      # pixel range beyond the image


def test_render_one_sharded_single_rank_equals_render_one(mods, dev):
    cfg, params = golden_params("4x128")
    g = load_golden("render_4x128")
    model = make_model(mods, cfg, params, dev)
    enc = mods["encoding"].PositionalEncoding(cfg["L"], True).to(dev)
    a = mods["train"].render_one(model, enc, g["H"], g["W"], g["focal"], g["pose"], dev, 64, 2.0, 6.0, 500)
    b = mods["train"].render_one_sharded(model, enc, g["H"], g["W"], g["focal"], g["pose"], dev, 64, 2.0, 6.0, 500)
    assert torch.equal(a, b) and float((a.cpu() - g["img"]).abs().max()) <= RGB_TOL


# ------------------------------------------------------------------------------------- bf16 mode (BASELINE cfg 4)
def _lively_params(cfg, seed=3):
    """Random-init weights whose density head is alive (seed-0 style inits often start with sigma == 0 everywhere)."""
    g = torch.Generator().manual_seed(seed)
    params = O.mlp_init(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"], g)
    params[2 * cfg["depth"] + 1] = params[2 * cfg["depth"] + 1] + 0.5          # sigma.0.bias
    return params


@pytest.mark.parametrize("tag", ["4x128", "8x256"])
def test_bf16_render_matches_cpu_restatement_and_fp32(mods, dev, tag):
    """bf16 weights/activations on MFMA, fp32 accumulate + compositing: against the CPU restatement of the same
    numerics (tight) and against the fp32 path on identical weights (SURVEY.md 8d cfg 4: max |dRGB| <= 2e-2)."""
    ops = mods["ops"]
    cfg, trained = golden_params(tag)
    g = load_golden(f"render_{tag}")
    ro, rd = O.pinhole_rays(g["H"], g["W"], g["focal"], g["pose"])
    for params in (trained, _lively_params(cfg)):
        model = make_model(mods, cfg, params, dev)
        st = model._ensure_packed()
        for (R, S, white) in ((8, 64, True), (1000, 64, True), (333, 32, False), (200, 128, True), (64, 48, True), (17, 100, False), (3, 2, True)):
            o, d = ro[1234:1234 + R].contiguous(), rd[1234:1234 + R].contiguous()
            want16 = O.render_rays_bf16(params, cfg["skip_at"], cfg["L"], o, d, 2.0, 6.0, S, white_bkgd=white)
            want32 = O.render_rays(params, cfg["skip_at"], cfg["L"], o, d, 2.0, 6.0, S, white_bkgd=white)
            comp, depth, acc = ops.render_rays_fused_bf16(st, o.to(dev), d.to(dev), 2.0, 6.0, S, white_bkgd=white)
            fp32 = ops.render_rays_fused(st, list(model.parameters()), o.to(dev), d.to(dev), 2.0, 6.0, S, False, white)[0]
            # a 1-ulp difference of the fp32 accumulation can flip a bf16 rounding of one activation (2^-8 relative)
            assert float((comp.cpu() - want16[0]).abs().max()) <= 2e-3, (tag, R, S)
            assert float((acc.cpu() - want16[2]).abs().max()) <= 2e-3
            assert float((depth.cpu() - want16[1]).abs().max()) <= 2e-2
            assert float((comp.cpu() - want32[0]).abs().max()) <= 2e-2, (tag, R, S)
            assert float((comp - fp32.detach()).abs().max()) <= 2e-2


def test_bf16_render_camera_full_image_and_jitter(mods, dev):
    """Full 100x100 image through the camera entry point (odd chunk sizes, a partial last workgroup) equals the
    table-sourced bf16 render bitwise, stays within 2e-2 of the fp32 image; explicit jitter is honoured."""
    ops = mods["ops"]
    cfg, _ = golden_params("8x256")
    params = _lively_params(cfg)
    model = make_model(mods, cfg, params, dev)
    st = model._ensure_packed()
    g = load_golden("render_8x256")
    H, W, focal, pose = g["H"], g["W"], g["focal"], g["pose"].to(dev)
    rd = torch.empty(H * W, 3, device=dev); ro = pose[:3, 3].expand(H * W, 3).contiguous()
    ro_t, rd_t = mods["rays"].get_rays(H, W, focal, pose)
    full = ops.render_rays_fused_bf16(st, ro_t.contiguous(), rd_t.contiguous(), 2.0, 6.0, 64)[0]
    parts = []
    for first, n in ((0, 4093), (4093, 5000), (9093, 907)):
        parts.append(ops.render_camera_fused_bf16(st, pose, H, W, focal, first, n, 2.0, 6.0, 64)[0])
    assert torch.equal(torch.cat(parts), full)
    fp32 = ops.render_camera_fused(st, pose, H, W, focal, 0, H * W, 2.0, 6.0, 64)[0]
    assert float((full - fp32).abs().max()) <= 2e-2
    mse = float(torch.mean((full - fp32) ** 2))
    assert -10.0 * math.log10(max(mse, 1e-12)) >= 45.0                 # PSNR of the bf16 image against the fp32 image
    # explicit jitter (the reference's RNG stream) through the bf16 kernel
    R, S = 512, 64
    t = torch.rand(R, S, generator=torch.Generator().manual_seed(5))
    o, d = ro_t[:R].cpu().contiguous(), rd_t[:R].cpu().contiguous()
    want = O.render_rays_bf16(params, cfg["skip_at"], cfg["L"], o, d, 2.0, 6.0, S, t_rand=t)[0]
    got = ops.render_rays_fused_bf16(st, o.to(dev), d.to(dev), 2.0, 6.0, S, randomized=True, t_rand=t.to(dev))[0]
    assert float((got.cpu() - want).abs().max()) <= 2e-3


def test_bf16_pack_is_round_to_nearest_even_gather(mods, dev):
    """tnerf_mlp_pack_bf16 = gather through the pack table + RNE rounding (weights) / plain copy (fp32 biases)."""
    ops = mods["ops"]
    cfg, params = golden_params("4x128")
    model = make_model(mods, cfg, params, dev)
    st = model._ensure_packed()
    b = st.repack_bf16()
    torch.cuda.synchronize()
    tab = b.table.cpu().long()
    flat = st.flat.cpu()
    vals = torch.where(tab >= 0, flat[tab.clamp(min=0)], torch.zeros(()))
    nw = b.n_fragments * 512
    raw = b.packed.cpu()
    got_w = raw[:nw * 2].view(torch.bfloat16)
    assert torch.equal(got_w, vals[:nw].to(torch.bfloat16))
    assert torch.equal(raw[nw * 2:].view(torch.float32), vals[nw:])


def _bf16_step_grads(mods, dev, model, st, o, d, tgt, t, S, cam=None, pixels=None):
    """One bf16 train step through the C ABI (no optimizer): (loss, comp, flat gradient)."""
    import ctypes as C
    ops, lib = mods["ops"], mods["lib"]
    R = t.shape[0]
    b = st.repack_bf16(); bp = b.train_plan(R, S)
    ztab = ops.depth_table(2.0, 6.0, S, dev)
    comp = torch.empty(R, 3, device=dev); gws = torch.empty(R, 4, device=dev); loss = torch.zeros(1, device=dev)
    st.grad.zero_()
    s_ = torch.cuda.current_stream(dev).cuda_stream
    tail = (comp.data_ptr(), gws.data_ptr(), gws.numel(), loss.data_ptr(), bp.stash.data_ptr(), bp.jobs.data_ptr(), bp.n_jobs, bp.slabs.data_ptr(),
            bp.reduce.data_ptr(), st.grad.data_ptr(), s_)
    if cam is None:
        lib.call("tnerf_train_step_fused_bf16", C.byref(st.desc), b.packed.data_ptr(), o.data_ptr(), d.data_ptr(), tgt.data_ptr(), R, S,
                 ztab.data_ptr(), 1, t.data_ptr(), 0, 0, 1, float(3 * R), *tail)
    else:
        lib.call("tnerf_train_step_fused_cam_bf16", C.byref(st.desc), b.packed.data_ptr(), C.byref(cam), pixels.data_ptr(), R, S,
                 ztab.data_ptr(), 1, t.data_ptr(), 0, 0, 1, float(3 * R), *tail)
    torch.cuda.synchronize()
    return float(loss), comp.cpu(), st.grad.cpu().clone()


@pytest.mark.parametrize("tag", ["4x128", "8x256"])
def test_bf16_train_gradients_match_cpu_restatement(mods, dev, tag):
    """bf16 forward + dgrad + wgrad against the hand-written CPU restatement of the same numerics (every tensor), and
    against the fp32 autograd gradients (bf16 rounding of activations / activation gradients: ~1 %)."""
    cfg, trained = golden_params(tag)
    g = load_golden(f"render_{tag}")
    ro, rd = O.pinhole_rays(int(g["H"]), int(g["W"]), float(g["focal"]), g["pose"])
    for params in (trained, _lively_params(cfg)):
        model = make_model(mods, cfg, params, dev)
        st = model._ensure_packed()
        for (R, S) in ((64, 64), (100, 48), (37, 100), (9, 2)):
            idx = torch.arange(0, ro.shape[0], max(1, ro.shape[0] // R))[:R]
            o, d = ro[idx].contiguous(), rd[idx].contiguous()
            tgt = torch.rand(R, 3, generator=torch.Generator().manual_seed(1))
            t = torch.rand(R, S, generator=torch.Generator().manual_seed(2))
            l16, _, g16 = O.loss_and_grads_bf16(params, cfg["skip_at"], cfg["L"], o, d, tgt, 2., 6., S, t)
            _, _, g32 = O.loss_and_grads(params, cfg["skip_at"], cfg["L"], o, d, tgt, 2., 6., S, t)
            loss, comp, flat = _bf16_step_grads(mods, dev, model, st, o.to(dev), d.to(dev), tgt.to(dev), t.to(dev), S)
            assert abs(loss - float(l16)) <= 2e-3 * max(1e-3, float(l16))
            w16 = torch.cat([x.reshape(-1) for x in g16]); w32 = torch.cat([x.reshape(-1) for x in g32])
            # 1-ulp differences of an fp32 accumulation (and of v_sin) flip a few bf16 roundings: 2^-8 on those entries
            assert float((flat - w16).norm() / w16.norm()) <= 1e-2, (tag, R, S)
            off = 0
            for i, x in enumerate(g16):
                n = x.numel()
                assert float((flat[off:off + n] - x.reshape(-1)).norm()) <= 3e-2 * float(x.norm()) + 1e-3 * float(w16.norm()), (tag, R, S, i)
                off += n
            assert float((flat - w32).norm() / w32.norm()) <= 6e-2, (tag, R, S)
            cos = float(torch.nn.functional.cosine_similarity(flat, w32, dim=0))
            assert cos >= 0.998, (tag, R, S, cos)


def test_bf16_train_large_batch_camera_and_determinism(mods, dev):
    """More ray groups than workgroups (several passes per workgroup, a ragged last group), camera-sourced rays equal
    table-sourced rays, and two launches give bit-identical gradients (slab reduction, no atomics)."""
    ops = mods["ops"]
    cfg, _ = golden_params("4x128")
    params = _lively_params(cfg)
    model = make_model(mods, cfg, params, dev)
    st = model._ensure_packed()
    H = W = 64; focal = 80.0
    pose = load_golden("render_4x128")["pose"]
    ro, rd = O.pinhole_rays(H, W, focal, pose)
    R, S = 2500, 64
    inds = torch.randint(0, H * W, (R,), generator=torch.Generator().manual_seed(4))
    o, d = ro[inds].contiguous(), rd[inds].contiguous()
    pixels = torch.rand(H * W, 3, generator=torch.Generator().manual_seed(1))
    t = torch.rand(R, S, generator=torch.Generator().manual_seed(2))
    l16, _, g16 = O.loss_and_grads_bf16(params, cfg["skip_at"], cfg["L"], o, d, pixels[inds].contiguous(), 2., 6., S, t)
    w16 = torch.cat([x.reshape(-1) for x in g16])
    a = _bf16_step_grads(mods, dev, model, st, o.to(dev), d.to(dev), pixels[inds].contiguous().to(dev), t.to(dev), S)
    b = _bf16_step_grads(mods, dev, model, st, o.to(dev), d.to(dev), pixels[inds].contiguous().to(dev), t.to(dev), S)
    assert torch.equal(a[2], b[2]) and torch.equal(a[1], b[1])
    assert float((a[2] - w16).norm() / w16.norm()) <= 1e-2
    cam, keep = ops.camera_struct(pose.to(dev), H, W, focal, inds.to(dev), 0)
    c = _bf16_step_grads(mods, dev, model, st, None, None, None, t.to(dev), S, cam=cam, pixels=pixels.to(dev))
    assert abs(c[0] - a[0]) <= 1e-6 and float((c[2] - a[2]).norm() / a[2].norm()) <= 2e-3     # rays differ by fp32 ulps only


def test_bf16_trainer_tracks_fp32_training(mods, dev):
    """SURVEY.md 8d cfg 4's protocol: the BASELINE model (8x256, L=6, skip 4), 4096 rays x 64 samples, 2000 steps from the
    same init on the same pixel / jitter stream in fp32 and in bf16 mode; |dPSNR| <= 0.1 dB on held-out views (8 poses of the
    scene that are not in the training set) and max |dRGB| <= 2e-2 between the bf16 and the fp32 render of identical weights.
    The held-out PSNR of ONE checkpoint moves by ~0.1 dB from step to step and between two equally valid fp32 evaluation
    orders of the same run (chaotic divergence of the trajectories; printed below), so the statistic compared is the mean
    over the checkpoints at steps 1500, 1600, ..., 2000."""
    from data import make_synthetic_scene
    trainer, train_mod = mods["trainer"], mods["train"]
    scene = make_synthetic_scene(seed=0)
    held = make_synthetic_scene(n_images=8, seed=1)                  # other cameras on the same scene
    images = torch.from_numpy(scene["images"]).to(dev); poses = torch.from_numpy(scene["poses"]).to(dev); focal = float(scene["focal"])
    h_images = torch.from_numpy(held["images"]).to(dev); h_poses = torch.from_numpy(held["poses"]).to(dev)
    N, H, W, _ = images.shape
    pixels = images.view(N, H * W, 3)
    enc = mods["encoding"].PositionalEncoding(6, True).to(dev)
    out, last, worst_rgb = {}, {}, 0.0
    for prec in ("fp32", "bf16"):
        torch.manual_seed(0)
        model = mods["nerf"].TinyNeRF(39, 256, 8, 4).to(dev)
        with torch.no_grad():
            model.sigma[0].bias += 0.5
        opt = trainer.FlatAdam(model, lr=5e-4)
        tr = trainer.FusedTrainer(model, opt, 2.0, 6.0, 64, precision=prec)
        gen = torch.Generator(device=dev); gen.manual_seed(7)
        evals = []
        for s in range(2000):
            i = s % N
            inds = torch.randint(0, H * W, (4096,), device=dev, generator=gen)
            u = torch.rand(4096, 64, device=dev, generator=gen)
            tr.step_camera(poses[i], H, W, focal, inds, pixels[i], t_rand=u)
            if s + 1 >= 1500 and (s + 1) % 100 == 0:
                ps = []
                for k in range(h_poses.shape[0]):
                    img = train_mod.render_one(model, enc, H, W, focal, h_poses[k], dev, n_samples=64, near=2.0, far=6.0)
                    ps.append(float(mods["utils"].mse2psnr(torch.mean((img - h_images[k]) ** 2))))
                    if s + 1 == 2000:
                        img16 = train_mod.render_one(model, enc, H, W, focal, h_poses[k], dev, n_samples=64, near=2.0, far=6.0, chunk=3000, precision="bf16")
                        worst_rgb = max(worst_rgb, float((img16 - img).abs().max()))
                evals.append(sum(ps) / len(ps))
        out[prec], last[prec] = sum(evals) / len(evals), evals
    print(f"held-out PSNR (8 unseen views), checkpoints 1500..2000: fp32 {[round(v, 3) for v in last['fp32']]} mean {out['fp32']:.3f}; "
          f"bf16 {[round(v, 3) for v in last['bf16']]} mean {out['bf16']:.3f}; max |dRGB| bf16 vs fp32 render of identical weights {worst_rgb:.2e}")
    assert out["fp32"] >= 24.0, out
    assert abs(out["fp32"] - out["bf16"]) <= 0.1, out
    assert worst_rgb <= 2e-2, worst_rgb


@pytest.mark.parametrize("arch", [(15, 128, 1, 0), (39, 128, 2, 1), (63, 256, 3, 2), (27, 256, 5, 0)])
def test_bf16_other_architectures(mods, dev, arch):
    """Shallow / skip-less / L=2..10 networks through the bf16 render and train kernels (the weight stream is then only
    a few stages long and wraps inside the DMA look-ahead)."""
    in_dim, hidden, depth, skip = arch
    cfg = dict(in_dim=in_dim, hidden=hidden, depth=depth, skip_at=skip, L=(in_dim - 3) // 6)
    params = _lively_params(cfg, seed=5)
    model = make_model(mods, cfg, params, dev)
    st = model._ensure_packed()
    g = torch.Generator().manual_seed(9)
    R, S = 77, 40
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    o = -4.0 * d + 0.2 * torch.randn(R, 3, generator=g)
    tgt, t = torch.rand(R, 3, generator=g), torch.rand(R, S, generator=g)
    want = O.render_rays_bf16(params, skip, cfg["L"], o, d, 2.0, 6.0, S, t_rand=t)[0]
    got = mods["ops"].render_rays_fused_bf16(st, o.to(dev), d.to(dev), 2.0, 6.0, S, randomized=True, t_rand=t.to(dev))[0]
    assert float((got.cpu() - want).abs().max()) <= 2e-3
    l16, _, g16 = O.loss_and_grads_bf16(params, skip, cfg["L"], o, d, tgt, 2., 6., S, t)
    loss, comp, flat = _bf16_step_grads(mods, dev, model, st, o.to(dev), d.to(dev), tgt.to(dev), t.to(dev), S)
    w16 = torch.cat([x.reshape(-1) for x in g16])
    assert abs(loss - float(l16)) <= 2e-3 * float(l16)
    assert float((flat - w16).norm() / w16.norm()) <= 1e-2, arch
