"""Host-side logic of libtnerf_hip.so, checked without a GPU:
  * the C-ABI library loads and exports every symbol declared in include/tnerf.h,
  * depth tables are bit-exact with torch.linspace / the reference arithmetic,
  * the MFMA fragment packing table reproduces the MLP when the kernel's data flow is emulated in numpy,
  * the wgrad job table + slab reduce table reproduce autograd's parameter gradients.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, golden_params
from oracle import tnerf_oracle as O
from tnerf import lib as tl


def _desc(in_dim, hidden, depth, skip_at):
    return tl.MlpDesc(in_dim, hidden, depth, skip_at)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "tnerf.h")).read()
    declared = set(re.findall(r"\b(tnerf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"tnerf_mlp_desc", "tnerf_plan_sizes"}
    assert len(declared) >= 24
    lib = tl.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in tnerf.h but not exported"
        assert name in tl.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.tnerf_version() == tl.ABI_VERSION == 3
    assert int(re.search(r"#define TNERF_ABI_VERSION (\d+)", hdr).group(1)) == 3
    assert tl.last_error() == "ok"


def test_step_args_struct_layout_matches_the_header(tmp_path):
    """The ctypes mirror of tnerf_step_args / tnerf_camera / the size structs has the C compiler's layout for include/tnerf.h
    (a binding that drifts from the header corrupts every field after the drift)."""
    import subprocess
    fields = [n for n, _ in tl.StepArgs._fields_]
    src = tmp_path / "layout.c"
    body = "".join(f'    printf("{n} %zu\\n", offsetof(tnerf_step_args, {n}));\n' for n in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "tnerf.h"\nint main(void) {\n' + body +
                   '    printf("sizeof %zu\\n", sizeof(tnerf_step_args));\n'
                   '    printf("camera %zu\\n", sizeof(tnerf_camera));\n'
                   '    printf("plan %zu\\n", sizeof(tnerf_plan_sizes));\n'
                   '    printf("bf16 %zu %zu\\n", sizeof(tnerf_bf16_sizes), sizeof(tnerf_bf16_train_plan));\n    return 0;\n}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split(" ", 1) for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for n in fields:
        assert int(out[n]) == getattr(tl.StepArgs, n).offset, n
    assert int(out["sizeof"]) == C.sizeof(tl.StepArgs)
    assert int(out["camera"]) == C.sizeof(tl.Camera)
    assert int(out["plan"]) == C.sizeof(tl.PlanSizes)
    assert out["bf16"].split() == [str(C.sizeof(tl.Bf16Sizes)), str(C.sizeof(tl.Bf16TrainPlan))]


def test_undersized_workspaces_are_refused_before_any_launch():
    """ABI 3: the per-ray workspace and the stash carry their capacity; a buffer sized for another library version (ABI 1's
    3*R floats, a stash without the bound words) is TNERF_ESMALL — checked on the host before the first launch, which is why
    this runs without a GPU: were anything launched here, the call would fail with a HIP error instead."""
    lib = tl.load()
    assert lib.tnerf_train_ws_floats(4096) == 4 * 4096 and tl.train_ws_floats(1) == 4
    assert lib.tnerf_train_ws_floats(0) == tl.EINVAL and "n_rays=0" in tl.last_error()
    d = _desc(39, 256, 8, 4)
    R, S, fake = 64, 32, 0x10000                          # never dereferenced: every call below must fail in validation
    ws_ok = tl.train_ws_floats(R)
    sz = tl.PlanSizes(); assert lib.tnerf_plan_sizes_query(C.byref(d), R * S, 256, C.byref(sz)) == 0
    for name, cam in (("tnerf_train_step_fused", False), ("tnerf_train_step_fused_cam", True)):
        for ws in (3 * R, ws_ok - 1):
            head = (C.byref(d), fake) + ((C.byref(tl.Camera(fake, 8, 8, 10.0, fake, 0)), fake) if cam else (fake, fake, fake))
            rc = getattr(lib, name)(*head, R, S, fake, 1, None, 0, 0, 1, float(3 * R), fake, fake, ws, fake, fake, sz.stash_row_stride,
                                    fake, sz.n_jobs, fake, fake, fake, None, None)
            assert rc == tl.ESMALL, (name, ws, tl.last_error())
            assert "tnerf_train_ws_floats" in tl.last_error()
    for name, cam in (("tnerf_train_step_fused_bf16", False), ("tnerf_train_step_fused_cam_bf16", True)):
        head = (C.byref(d), fake) + ((C.byref(tl.Camera(fake, 8, 8, 10.0, fake, 0)), fake) if cam else (fake, fake, fake))
        rc = getattr(lib, name)(*head, R, S, fake, 1, None, 0, 0, 1, float(3 * R), fake, fake, 3 * R, fake, fake, fake, 4, fake, fake, fake, None)
        assert rc == tl.ESMALL, (name, tl.last_error())
    # the dataset step: per-ray workspace, fp32 stash, bf16 stash
    def args(precision):
        a = tl.StepArgs()
        a.desc, a.precision, a.phases = d, precision, tl.PHASE_GRADIENT
        a.poses = a.pixels = a.ztab = a.step = a.packed = a.comp_rgb = a.ray_ws = a.stash = a.job_table = a.slabs = fake
        a.n_images, a.H, a.W, a.focal = 2, 8, 8, 10.0
        a.n_rays, a.ray_first, a.n_rays_global, a.n_samples = R, 0, R, S
        a.loss_denominator, a.n_jobs = 3.0 * R, sz.n_jobs
        a.ray_ws_floats, a.stash_row_stride, a.stash_capacity = ws_ok, sz.stash_row_stride, sz.stash_floats
        return a
    a = args(0); a.ray_ws_floats = 3 * R
    assert lib.tnerf_train_step_dataset(C.byref(a), None) == tl.ESMALL and "per-ray workspace" in tl.last_error()
    a = args(0); a.stash_capacity = sz.stash_floats - 1
    assert lib.tnerf_train_step_dataset(C.byref(a), None) == tl.ESMALL and "stash_floats" in tl.last_error()
    a = args(0); a.stash_row_stride = R * S - 64
    assert lib.tnerf_train_step_dataset(C.byref(a), None) == tl.EINVAL and "stash_row_stride" in tl.last_error()
    bp = tl.Bf16TrainPlan(); assert lib.tnerf_bf16_train_sizes(C.byref(d), R, S, 256, C.byref(bp)) == 0
    a = args(1); a.stash_capacity = bp.stash_bytes - 1
    assert lib.tnerf_train_step_dataset(C.byref(a), None) == tl.ESMALL and "stash_bytes" in tl.last_error()


def test_error_reporting_no_throw():
    lib = tl.load()
    d = _desc(39, 300, 8, 4)
    assert lib.tnerf_param_count(C.byref(d)) == -1
    assert "hidden=300" in tl.last_error()
    with pytest.raises(NotImplementedError):
        tl.check(tl.EUNSUPPORTED, "x")
    d = _desc(39, 256, 8, 8)     # skip_at == depth: heads would see hidden+in_dim (reference raises too)
    assert lib.tnerf_param_count(C.byref(d)) == -1


@pytest.mark.parametrize("S", [1, 2, 3, 7, 32, 50, 63, 64, 65, 100, 128, 255, 256, 257])
@pytest.mark.parametrize("near,far", [(2.0, 6.0), (0.5, 3.25), (0.1, 7.3), (1e-3, 1e3)])
def test_sample_tables_bitexact_vs_torch(S, near, far):
    lib = tl.load()
    ztab = np.empty(3 * S, np.float32); t = np.empty(S, np.float32)
    assert lib.tnerf_sample_tables(near, far, S, _ptr(ztab), _ptr(t)) == 0
    tt = torch.linspace(0., 1., steps=S)
    assert np.array_equal(t, tt.numpy()), "linspace"
    z = near * (1. - tt) + far * tt
    assert np.array_equal(ztab[:S], z.numpy()), "z"
    zz = z.expand(2, S)
    if S > 1:
        mids = 0.5 * (zz[:, :-1] + zz[:, 1:])
        hi = torch.cat([mids, zz[:, -1:]], -1)[0]; lo = torch.cat([zz[:, :1], mids], -1)[0]
    else:
        hi = lo = zz[0]
    assert np.array_equal(ztab[S:2 * S], lo.numpy()) and np.array_equal(ztab[2 * S:], hi.numpy())


def test_sample_tables_golden():
    g = load_golden("sampling")
    lib = tl.load()
    for S in (64, 128, 256):
        ztab = np.empty(3 * S, np.float32); t = np.empty(S, np.float32)
        lib.tnerf_sample_tables(2.0, 6.0, S, _ptr(ztab), _ptr(t))
        assert np.array_equal(t, g[f"linspace_{S}"].numpy())
        assert np.array_equal(ztab[:S], g[f"zbase_{S}"].numpy())
        # the jittered depths of the fixture follow from the table + the recorded draws, bit for bit
        u = g[f"u_{S}"].numpy()
        z = (ztab[S:2 * S] + ((ztab[2 * S:] - ztab[S:2 * S]).astype(np.float32) * u).astype(np.float32)).astype(np.float32)
        assert np.array_equal(z, g[f"z_rand_{S}"].numpy())


def test_param_layout_matches_state_dict_order():
    lib = tl.load()
    for cfg in ((39, 256, 8, 4), (63, 128, 4, 2), (39, 128, 3, 0)):
        d = _desc(*cfg)
        shapes = O.mlp_shapes(*cfg)
        n = 2 * cfg[2] + 4
        off = np.zeros(n, np.int64); rows = np.zeros(n, np.int64); cols = np.zeros(n, np.int64)
        assert lib.tnerf_param_layout(C.byref(d), _ptr(off), _ptr(rows), _ptr(cols)) == 0
        run = 0
        for i, sh in enumerate(shapes):
            assert off[i] == run
            assert int(rows[i] * cols[i]) == int(np.prod(sh))
            run += int(np.prod(sh))
        assert lib.tnerf_param_count(C.byref(d)) == run
    assert lib.tnerf_param_count(C.byref(_desc(39, 256, 8, 4))) == 481796
    assert lib.tnerf_param_count(C.byref(_desc(63, 128, 4, 2))) == 66308


def _plan(cfg, M, n_cu=256):
    lib = tl.load()
    d = _desc(*cfg)
    sz = tl.PlanSizes()
    assert lib.tnerf_plan_sizes_query(C.byref(d), M, n_cu, C.byref(sz)) == 0, tl.last_error()
    pack = np.empty(sz.packed_floats, np.int32); jobs = np.empty(sz.job_ints, np.int32); red = np.empty(sz.reduce_ints, np.int32)
    assert lib.tnerf_plan_fill(C.byref(d), M, n_cu, _ptr(pack), _ptr(jobs), _ptr(red)) == 0, tl.last_error()
    emap = np.empty(64, np.int32); ne = C.c_int32()
    assert lib.tnerf_input_pairing(cfg[0], _ptr(emap), C.byref(ne)) == 0
    return sz, pack, jobs.reshape(-1, 16), red, emap.reshape(32, 2), ne.value


def _acc_row(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def _mfma(a, b, acc):
    """v_mfma_f32_32x32x2_f32: a, b: [64] lane values; acc: [16, 64].  D[i][j] += sum_k A[i][k] B[k][j]."""
    A = a.reshape(2, 32).T            # [i][k]
    B = b.reshape(2, 32)              # [k][j]
    D = A @ B                         # [32, 32]
    for r in range(16):
        for h in range(2):
            acc[r, 32 * h:32 * h + 32] += D[_acc_row(r, h), :]
    return acc


@pytest.mark.parametrize("cfg", [(39, 256, 8, 4), (63, 128, 4, 2), (10, 128, 2, 1)])
def test_pack_table_emulated_mfma_chain_equals_mlp(cfg):
    """Run the forward kernel's exact register data flow (fragments, accumulator->operand chaining) in numpy."""
    in_dim, hidden, depth, skip_at = cfg
    sz, pack, jobs, red, emap, NE = _plan(cfg, 64)
    g = torch.Generator().manual_seed(7)
    params = O.mlp_init(in_dim, hidden, depth, skip_at, g)
    flat = torch.cat([p.reshape(-1) for p in params]).double().numpy()
    packed = np.where(pack >= 0, flat[np.clip(pack, 0, None)], 0.0)
    x = torch.randn(32, in_dim, generator=g)
    NT = hidden // 32
    lane = np.arange(64); j = lane & 31; h = lane >> 5
    # network input registers: enc[st][lane]
    enc = np.zeros((NE, 64))
    for st in range(NE):
        for L_ in range(64):
            c = emap[st, h[L_]]
            enc[st, L_] = x[j[L_], c].item() if c >= 0 else 0.0
    # walk the packed buffer in the order tn_build_layout lays it out
    pos = 0

    def take(n):
        nonlocal pos
        out = packed[pos:pos + n]; pos += n
        return out

    hcur = None
    for l in range(depth):
        bias = take(NT * 32).reshape(NT, 2, 16)
        has_enc = (l == 0) or (skip_at > 0 and l == skip_at)
        We = take(NT * NE * 64).reshape(NT, NE // 4, 64, 4) if has_enc else None
        Wh = take(NT * NT * 1024).reshape(NT, NT * 4, 64, 4) if l > 0 else None
        hnext = np.zeros((hidden // 2, 64))
        for t in range(NT):
            acc = np.zeros((16, 64))
            for r in range(16):
                acc[r] = bias[t, h, r]
            if Wh is not None:
                for gq in range(NT * 4):
                    for p in range(4):
                        acc = _mfma(Wh[t, gq, :, p], hcur[gq * 4 + p], acc)
            if We is not None:
                for gq in range(NE // 4):
                    for p in range(4):
                        acc = _mfma(We[t, gq, :, p], enc[gq * 4 + p], acc)
            hnext[t * 16:(t + 1) * 16] = np.maximum(acc, 0.0)
        hcur = hnext
    hb = take(32).reshape(2, 16)
    Wd = take(NT * 1024).reshape(NT * 4, 64, 4)
    acc = np.zeros((16, 64))
    for r in range(4):
        acc[r] = hb[h, r]
    for gq in range(NT * 4):
        for p in range(4):
            acc = _mfma(Wd[gq, :, p], hcur[gq * 4 + p], acc)
    rgb = 1.0 / (1.0 + np.exp(-acc[:3, :32])); sigma = np.maximum(acc[3, :32], 0.0)
    want_rgb, want_sigma = O.mlp_forward([p.double() for p in params], x.double(), skip_at)
    np.testing.assert_allclose(rgb.T, want_rgb.numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(sigma, want_sigma.numpy()[:, 0], rtol=0, atol=1e-12)

    # transposed fragments: dH_{l-1} = W_l^T dZ_l for every hidden layer, and the heads
    dz = np.random.RandomState(0).randn(hidden // 2, 64)           # registers [t*16+r][lane] = dZ[32t+row(r,h)][j]
    dZ = np.zeros((hidden, 32))
    for t in range(NT):
        for r in range(16):
            for hh in range(2):
                dZ[32 * t + _acc_row(r, hh)] = dz[t * 16 + r, 32 * hh:32 * hh + 32]
    for l in range(1, depth):
        Wt = take(NT * NT * 1024).reshape(NT, NT * 4, 64, 4)
        W = params[2 * l].double().numpy()[:, :hidden]
        want = W.T @ dZ
        for t in range(NT):
            acc = np.zeros((16, 64))
            for gq in range(NT * 4):
                for p in range(4):
                    acc = _mfma(Wt[t, gq, :, p], dz[gq * 4 + p], acc)
            for r in range(16):
                for hh in range(2):
                    np.testing.assert_allclose(acc[r, 32 * hh:32 * hh + 32], want[32 * t + _acc_row(r, hh)], atol=1e-12)
    Wth = take(NT * 256).reshape(NT, 64, 4)
    dzh = np.random.RandomState(1).randn(4, 32)
    Whead = np.concatenate([params[2 * depth + 2].double().numpy(), params[2 * depth].double().numpy()], 0)   # r,g,b,sigma
    want = Whead.T @ dzh
    for t in range(NT):
        acc = np.zeros((16, 64))
        for p in range(4):
            b = np.concatenate([dzh[p], np.zeros(32)])
            acc = _mfma(Wth[t, :, p], b, acc)
        for r in range(16):
            for hh in range(2):
                np.testing.assert_allclose(acc[r, 32 * hh:32 * hh + 32], want[32 * t + _acc_row(r, hh)], atol=1e-12)
    assert pos == sz.packed_floats


@pytest.mark.parametrize("cfg,M,n_cu", [((39, 256, 8, 4), 2048, 256), ((63, 128, 4, 2), 1000, 256), ((39, 128, 3, 0), 96, 8)])
def test_wgrad_jobs_and_reduce_table_reproduce_autograd(cfg, M, n_cu):
    in_dim, hidden, depth, skip_at = cfg
    sz, pack, jobs, red, emap, NE = _plan(cfg, M, n_cu)
    Mp = sz.stash_row_stride
    assert Mp % 64 == 0 and Mp >= M and len(jobs) == sz.n_jobs
    g = torch.Generator().manual_seed(11)
    params = [p.double().requires_grad_(True) for p in O.mlp_init(in_dim, hidden, depth, skip_at, g)]
    x = torch.randn(M, in_dim, generator=g).double()
    # forward with the intermediate activations kept, as the stash would hold them
    hs, zs = [], []
    hcur = x
    for l in range(depth):
        z = torch.nn.functional.linear(hcur, params[2 * l], params[2 * l + 1]); z.retain_grad(); zs.append(z)
        hh = torch.relu(z); hs.append(hh)
        hcur = torch.cat([hh, x], -1) if l == skip_at - 1 else hh
    zr = torch.nn.functional.linear(hcur, params[2 * depth + 2], params[2 * depth + 3]); zr.retain_grad()
    zsg = torch.nn.functional.linear(hcur, params[2 * depth], params[2 * depth + 1]); zsg.retain_grad()
    loss = (torch.sigmoid(zr) * torch.randn(M, 3, generator=g).double()).sum() + (torch.relu(zsg) * torch.randn(M, 1, generator=g).double()).sum()
    loss.backward()
    rows = 2 * NE + 2 * depth * hidden + 8
    assert sz.stash_floats == rows * (Mp + 32) + depth * (Mp + 32) * (hidden // 32) + 64   # float rows + ReLU sign bits (+ dump block) + magnitude-bound words
    stash = np.full((rows, Mp), np.nan)
    r0 = 0
    for st in range(NE):
        for hh in range(2):
            c = emap[st, hh]
            stash[r0 + 2 * st + hh, :M] = x[:, c].numpy() if c >= 0 else 0.0
    r0 += 2 * NE
    for l in range(depth):
        stash[r0:r0 + hidden, :M] = hs[l].detach().numpy().T; r0 += hidden
    r0 += 4
    for l in range(depth):
        stash[r0:r0 + hidden, :M] = zs[l].grad.numpy().T; r0 += hidden
    stash[r0:r0 + 3, :M] = zr.grad.numpy().T; stash[r0 + 3, :M] = zsg.grad.numpy()[:, 0]
    # run every job in numpy
    slabs = np.zeros(sz.slab_floats)
    covered = {}
    for jb in jobs:
        a0, ar, b0, br, nat, nbt, wa, blk0, nblk, soff, cls, hasb = (int(v) for v in jb[:12])
        ta, tb = -(-nat // wa), -(-nbt // (8 // wa))          # tiles per wave: every wave is full or idle
        assert wa in (1, 2, 4, 8) and nat % ta == 0 and nbt % tb == 0 and (ta, tb) in ((2, 4), (1, 2), (2, 1), (1, 1))
        m0, m1 = blk0 * 32, min(M, (blk0 + nblk) * 32)
        assert nblk > 0 and m0 < M
        covered.setdefault(cls, []).append((m0, m1))
        A = np.zeros((nat * 32, m1 - m0)); B = np.zeros((nbt * 32, m1 - m0))
        A[:ar] = stash[a0:a0 + ar, m0:m1]; B[:br] = stash[b0:b0 + br, m0:m1]
        blk = A @ B.T
        slabs[soff:soff + blk.size] = blk.reshape(-1)
        slabs[soff + blk.size: soff + blk.size + nat * 32] = A.sum(1)
    for cls, spans in covered.items():            # every class covers [0, M) exactly once
        spans.sort()
        assert spans[0][0] == 0 and spans[-1][1] == M
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert sz.n_jobs <= max(2 * n_cu, 64)
    ncls = red[0]
    hdr = red[1:1 + 4 * ncls].reshape(ncls, 4)
    ent = red[260:].reshape(-1, 2)
    grads = np.zeros(sz.n_params)
    for i in range(sz.n_params):
        off, cls = ent[i]
        first, stride, nch = hdr[cls, 0], hdr[cls, 1], hdr[cls, 2]
        grads[i] = sum(slabs[first + off + c * stride] for c in range(nch))
    want = torch.cat([p.grad.reshape(-1) for p in params]).numpy()
    np.testing.assert_allclose(grads, want, rtol=1e-9, atol=1e-9)


def test_input_pairing_is_a_permutation():
    lib = tl.load()
    for in_dim in (3, 9, 10, 39, 40, 45, 63, 64):
        emap = np.empty(64, np.int32); ne = C.c_int32()
        assert lib.tnerf_input_pairing(in_dim, _ptr(emap), C.byref(ne)) == 0
        used = emap[emap >= 0]
        assert sorted(used.tolist()) == list(range(in_dim))
        assert ne.value in (20, 32) and (emap[2 * ne.value:] == -1).all()
    L = 6
    emap = np.empty(64, np.int32); ne = C.c_int32()
    lib.tnerf_input_pairing(39, _ptr(emap), C.byref(ne))
    e = emap.reshape(32, 2)
    for k in range(L):
        for c in range(3):
            assert e[3 * k + c, 0] == 3 + 6 * k + c and e[3 * k + c, 1] == 3 + 6 * k + 3 + c     # (sin, cos) pair
    assert e[18].tolist() == [0, 1] and e[19].tolist() == [2, -1]


# ------------------------------------------------------------------------------------- bf16 mode
def _mfma16(a, b, acc):
    """v_mfma_f32_32x32x16_bf16 in exact arithmetic: a, b: [64, 8] (lane, element) with k = 8*(lane>>5) + e;
    acc: [16, 64].  D[i][j] += sum_k A[i][k] B[k][j]."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for L_ in range(64):
        A[L_ & 31, 8 * (L_ >> 5):8 * (L_ >> 5) + 8] = a[L_]
        B[8 * (L_ >> 5):8 * (L_ >> 5) + 8, L_ & 31] = b[L_]
    D = A @ B
    for r in range(16):
        for h in range(2):
            acc[r, 32 * h:32 * h + 32] += D[_acc_row(r, h), :]
    return acc


@pytest.mark.parametrize("cfg", [(39, 256, 8, 4), (63, 128, 4, 2), (39, 128, 2, 1), (15, 128, 1, 0)])
def test_bf16_fragment_stream_emulated_chain_equals_mlp(cfg):
    """The bf16 kernels' register data flow (fragment stream order, accumulator -> packed B operand chaining, input
    k-slot map) emulated in exact arithmetic must reproduce TinyNeRF.forward for the same (unrounded) weights."""
    in_dim, hidden, depth, skip_at = cfg
    lib = tl.load()
    d = _desc(*cfg)
    sz = tl.Bf16Sizes()
    assert lib.tnerf_bf16_plan_sizes(C.byref(d), C.byref(sz)) == 0, tl.last_error()
    tab = np.empty(sz.pack_entries, np.int32)
    assert lib.tnerf_bf16_pack_table(C.byref(d), _ptr(tab)) == 0, tl.last_error()
    assert sz.n_fwd_fragments % 16 == 0 and sz.n_fragments % 16 == 0 and sz.bias_offset_bytes == sz.n_fragments * 1024
    assert sz.packed_bytes == sz.bias_offset_bytes + 4 * (depth * hidden + 4)
    NF = sz.n_fwd_fragments
    g = torch.Generator().manual_seed(11)
    params = O.mlp_init(in_dim, hidden, depth, skip_at, g)
    flat = torch.cat([p.reshape(-1) for p in params]).double().numpy()
    used = tab[:NF * 512][tab[:NF * 512] >= 0]
    assert len(np.unique(used)) == len(used) == flat.size - (depth * hidden + 4)   # every weight exactly once in the forward stream
    vals = np.where(tab >= 0, flat[np.clip(tab, 0, None)], 0.0)
    frags = vals[:sz.n_fragments * 512].reshape(sz.n_fragments, 64, 8)
    bias = vals[sz.n_fragments * 512:]
    Lf = (in_dim - 3) // 6
    x3 = torch.rand(32, 3, generator=g) * 4 - 2
    x = O.posenc(x3.double(), Lf, True)                               # [32, in_dim]
    NT, KH, KE = hidden // 32, hidden // 16, 4
    lane = np.arange(64); j = lane & 31; h = lane >> 5
    # input k-steps: slot a = 8u+e -> (sin, cos)[h] of 2^(a/3) x_(a%3); a = 3L: (x, y)[h]; a = 3L+1: (z, 0)[h]
    enc = np.zeros((KE, 64, 8))
    for u in range(KE):
        for e in range(8):
            a = 8 * u + e
            for L_ in range(64):
                if a < 3 * Lf:
                    col = 3 + 6 * (a // 3) + (a % 3) + 3 * h[L_]
                elif a == 3 * Lf:
                    col = h[L_]
                elif a == 3 * Lf + 1:
                    col = 2 if h[L_] == 0 else -1
                else:
                    col = -1
                enc[u, L_, e] = x[j[L_], col].item() if col >= 0 else 0.0
    f = 0
    cur = None                                                        # [KH, 64, 8] B operands of the hidden k-steps
    for l in range(depth):
        nxt = np.zeros((KH, 64, 8))
        for t in range(NT):
            acc = np.zeros((16, 64))
            if l > 0:
                for s_ in range(KH):
                    acc = _mfma16(frags[f], cur[s_], acc); f += 1
            if l == 0 or (skip_at > 0 and l == skip_at):
                for u in range(KE):
                    acc = _mfma16(frags[f], enc[u], acc); f += 1
            for r in range(16):
                for L_ in range(64):
                    v = max(acc[r, L_] + bias[l * hidden + 32 * t + _acc_row(r, h[L_])], 0.0)
                    nxt[2 * t + (r >> 3), L_, r & 7] = v            # registers 0..7 -> k-step 2t, 8..15 -> k-step 2t+1
        assert f % 16 == 0                                            # every layer is a whole number of stages
        cur = nxt
    acc = np.zeros((16, 64))
    for s_ in range(KH):
        acc = _mfma16(frags[f], cur[s_], acc); f += 1
    assert np.all(frags[f:NF] == 0.0) and NF - f == 16 - KH
    hb = bias[depth * hidden:depth * hidden + 4]
    rgb = 1.0 / (1.0 + np.exp(-(acc[:3, :32] + hb[:3, None]))); sigma = np.maximum(acc[3, :32] + hb[3], 0.0)
    want_rgb, want_sigma = O.mlp_forward([p.double() for p in params], x, skip_at)
    np.testing.assert_allclose(rgb.T, want_rgb.numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(sigma, want_sigma.numpy()[:, 0], rtol=0, atol=1e-12)

    # ---- backward stream: dH = W^T dZ with dZ in the same accumulator -> operand register layout
    def to_operand(M):                                               # M: [hidden, 32] (feature, sample) -> [KH, 64, 8]
        out = np.zeros((KH, 64, 8))
        for s_ in range(KH):
            for L_ in range(64):
                for e in range(8):
                    out[s_, L_, e] = M[32 * (s_ >> 1) + _acc_row(8 * (s_ & 1) + e, h[L_]), j[L_]]
        return out

    def from_acc(acc):                                               # [16, 64] -> [32 rows, 32 samples]
        out = np.zeros((32, 32))
        for r in range(16):
            for hh in range(2):
                out[_acc_row(r, hh)] = acc[r, 32 * hh:32 * hh + 32]
        return out

    rs = np.random.RandomState(5)
    f = NF
    dzh = rs.randn(4, 32)                                            # rows r,g,b,sigma
    zh = np.zeros((64, 8)); zh[:32, :4] = dzh.T                      # lane-half 0, elements 0..3
    Whead = np.concatenate([params[2 * depth + 2].double().numpy(), params[2 * depth].double().numpy()], 0)
    want = Whead.T @ dzh
    for t in range(NT):
        np.testing.assert_allclose(from_acc(_mfma16(frags[f + t], zh, np.zeros((16, 64)))), want[32 * t:32 * t + 32], atol=1e-12)
    assert np.all(frags[f + NT:f + 16] == 0.0)
    f += 16
    for l in range(depth - 1, 0, -1):
        dZ = rs.randn(hidden, 32)
        op = to_operand(dZ)
        want = params[2 * l].double().numpy()[:, :hidden].T @ dZ
        for t in range(NT):
            acc = np.zeros((16, 64))
            for s_ in range(KH):
                acc = _mfma16(frags[f], op[s_], acc); f += 1
            np.testing.assert_allclose(from_acc(acc), want[32 * t:32 * t + 32], atol=1e-12)
    assert f == sz.n_fragments

    # ---- the transposing MFMAs: operand pair x selectors -> feature on the lane, samples in the registers
    X = rs.randn(32, 32)                                             # (feature-in-tile, sample)
    op = to_operand(np.concatenate([X, np.zeros((hidden - 32, 32))], 0))[:2]
    sel = np.zeros((2, 64, 8))
    for u in range(2):
        for L_ in range(64):
            for e in range(8):
                sel[u, L_, e] = 1.0 if (L_ & 31) == _acc_row(8 * u + e, L_ >> 5) else 0.0
    acc = _mfma16(op[1], sel[1], _mfma16(op[0], sel[0], np.zeros((16, 64))))
    for r in range(16):
        for L_ in range(64):
            assert acc[r, L_] == X[L_ & 31, _acc_row(r, L_ >> 5)]    # lane c = feature, register r = sample slot


@pytest.mark.parametrize("cfg,R,S,n_cu", [((39, 256, 8, 4), 64, 64, 256), ((63, 128, 4, 2), 37, 100, 256), ((39, 128, 3, 0), 5, 33, 8)])
def test_bf16_train_plan_jobs_and_reduce_table_reproduce_autograd(cfg, R, S, n_cu):
    """Emulate the bf16 wgrad decomposition on the host: per job, dW = A^T B over its tiles with A, B read from a stash
    laid out as the kernels write it (feature tiles; input slots for the encoder tiles), then the reduce table must
    gather exactly autograd's gradients."""
    in_dim, hidden, depth, skip_at = cfg
    lib = tl.load()
    d = _desc(*cfg)
    pl = tl.Bf16TrainPlan()
    assert lib.tnerf_bf16_train_sizes(C.byref(d), R, S, n_cu, C.byref(pl)) == 0, tl.last_error()
    jobs = np.empty(pl.job_ints, np.int32); red = np.empty(pl.reduce_ints, np.int32)
    assert lib.tnerf_bf16_train_fill(C.byref(d), R, S, n_cu, _ptr(jobs), _ptr(red)) == 0, tl.last_error()
    jobs = jobs.reshape(-1, 16)
    TPR = (S + 31) // 32
    tiles = R * TPR
    assert pl.n_tiles == tiles and len(jobs) == pl.n_jobs <= max(n_cu, depth + 2 + (1 if skip_at else 0))
    NT = hidden // 32
    n_ft = 2 + 2 * NT * depth + 1
    ft_enc, ft_h, ft_dz, ft_dzh = 0, [2 + NT * l for l in range(depth)], [2 + NT * depth + NT * l for l in range(depth)], 2 + 2 * NT * depth
    assert pl.stash_bytes == (tiles + 1) * n_ft * 2048 + depth * (tiles + 1) * 64 * (hidden // 64) * 4 + (tiles + 1) * 512
    g = torch.Generator().manual_seed(3)
    params = [p.double().requires_grad_(True) for p in O.mlp_init(in_dim, hidden, depth, skip_at, g)]
    Lf = (in_dim - 3) // 6
    M = tiles * 32
    x = torch.randn(M, in_dim, generator=g, dtype=torch.float64)
    # forward with hooks on the pre-activations to obtain dZ_l from autograd
    hcur, zs, hs = x, [], []
    for i in range(depth):
        zz = torch.nn.functional.linear(hcur, params[2 * i], params[2 * i + 1]); zz.retain_grad(); zs.append(zz)
        hh = torch.relu(zz); hs.append(hh)
        hcur = torch.cat([hh, x], -1) if i == skip_at - 1 else hh
    zc = torch.nn.functional.linear(hcur, params[2 * depth + 2], params[2 * depth + 3]); zc.retain_grad()
    zsg = torch.nn.functional.linear(hcur, params[2 * depth], params[2 * depth + 1]); zsg.retain_grad()
    up = torch.randn(M, 4, generator=g, dtype=torch.float64)
    (torch.cat([zc, zsg], -1) * up).sum().backward()
    # the stash as [tile*32 + slot, n_ft*32] (feature tile ft, column c)
    stash = np.zeros((M, n_ft * 32))
    for col in range(in_dim):                                        # encoder tiles: input slot order
        for u in range(4):
            for hh_ in range(2):
                for e in range(8):
                    a = 8 * u + e
                    cc = 3 + 6 * (a // 3) + (a % 3) + 3 * hh_ if a < 3 * Lf else (hh_ if a == 3 * Lf else ((2 if hh_ == 0 else -1) if a == 3 * Lf + 1 else -1))
                    if cc == col:
                        stash[:, (ft_enc + (u >> 1)) * 32 + _acc_row(8 * (u & 1) + e, hh_)] = x[:, col].numpy()
    for l in range(depth):
        stash[:, ft_h[l] * 32:(ft_h[l] + NT) * 32] = hs[l].detach().numpy()
        stash[:, ft_dz[l] * 32:(ft_dz[l] + NT) * 32] = zs[l].grad.numpy()
    stash[:, ft_dzh * 32:ft_dzh * 32 + 3] = zc.grad.numpy(); stash[:, ft_dzh * 32 + 3] = zsg.grad.numpy()[:, 0]
    slabs = np.zeros(pl.slab_floats)
    covered = {}
    for jb in jobs:
        a0, nat, b0, nbt, wa, t0, ntl, off, cls, hb = jb[0], jb[4], jb[2], jb[5], jb[6], jb[7], jb[8], jb[9], jb[10], jb[11]
        assert ntl > 0 and wa in (1, 2, 4, 8)
        rows = slice(t0 * 32, (t0 + ntl) * 32)
        A = stash[rows, a0 * 32:(a0 + nat) * 32]; B = stash[rows, b0 * 32:(b0 + nbt) * 32]
        slabs[off:off + nat * 32 * nbt * 32] = (A.T @ B).reshape(-1)
        if hb:
            slabs[off + nat * 32 * nbt * 32:off + nat * 32 * nbt * 32 + nat * 32] = A.sum(0)
        covered.setdefault(cls, []).append((t0, ntl))
    for cls, spans in covered.items():                               # every class covers every tile exactly once
        spans.sort()
        assert spans[0][0] == 0 and all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
        assert spans[-1][0] + spans[-1][1] == tiles
    ncls = red[0]
    E = red[260:].reshape(-1, 2)
    got = np.zeros(len(E))
    for i, (elem, cls) in enumerate(E):
        s0, stride, n = red[1 + 4 * cls], red[1 + 4 * cls + 1], red[1 + 4 * cls + 2]
        got[i] = sum(slabs[s0 + elem + c * stride] for c in range(n))
    want = np.concatenate([p.grad.reshape(-1).numpy() for p in params])
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)


def test_bf16_mode_rejects_generic_input_width():
    lib = tl.load()
    sz = tl.Bf16Sizes()
    assert lib.tnerf_bf16_plan_sizes(C.byref(_desc(10, 128, 2, 1)), C.byref(sz)) == tl.EUNSUPPORTED
    assert b"6L+3" in lib.tnerf_last_error_string()


def test_oracle_bf16_forward_is_close_to_fp32():
    """Sanity of the CPU restatement of the bf16 mode: bf16 rounding of weights/activations moves the outputs by a
    few 1e-3, far less than SURVEY.md 8(d) cfg 4's 2e-2 bound on rendered RGB."""
    g = load_golden("mlp_8x256")
    cfg, params = golden_params("8x256")
    rgb, sigma = O.mlp_forward(params, g["x"], cfg["skip_at"])
    rgb16, sigma16 = O.mlp_forward_bf16(params, g["x"], cfg["skip_at"])
    assert float((rgb - rgb16).abs().max()) < 2e-2
    assert float((sigma - sigma16).abs().max()) <= 2e-2 * max(1.0, float(sigma.abs().max()))


# ------------------------------------------------------------------------------- round 2: Philox draws of the dataset step
def philox4x32_10(seed, index):
    """numpy statement of the counter-based generator the kernels use (csrc/dev_common.hpp tn_philox_u32): Philox4x32-10,
    counter = (index >> 2, 0, 0) as (c0 | c1 << 32), key = seed, output word index & 3.  Vectorised over `index` (uint64)."""
    index = np.asarray(index, dtype=np.uint64)
    blk = index >> np.uint64(2)
    c0 = (blk & np.uint64(0xFFFFFFFF)).astype(np.uint64); c1 = (blk >> np.uint64(32)).astype(np.uint64)
    c2 = np.zeros_like(c0); c3 = np.zeros_like(c0)
    k0 = np.uint64(seed & 0xFFFFFFFF); k1 = np.uint64((seed >> 32) & 0xFFFFFFFF)
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2                       # 32x32 -> 64 bit products
        h0, l0, h1, l1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (h1 ^ c1 ^ k0) & MASK, l1, (h0 ^ c3 ^ k1) & MASK, l0
        k0 = (k0 + np.uint64(0x9E3779B9)) & MASK; k1 = (k1 + np.uint64(0xBB67AE85)) & MASK
    w = (index & np.uint64(3)).astype(np.int64)
    return np.choose(w, [c0, c1, c2, c3]).astype(np.uint32)


def test_philox_emulation_known_answers():
    """Random123's known-answer vectors for Philox4x32-10 (kat_vectors): counter 0 / key 0, and the all-ones vector."""
    out = [int(philox4x32_10(0, i)) for i in range(4)]
    assert out == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8], [hex(x) for x in out]
    # counter = (ffffffff, ffffffff, 0, 0) is reachable through the index mapping only up to 2^62; check determinism / spread instead
    a = philox4x32_10(1234, np.arange(4096, dtype=np.uint64))
    b = philox4x32_10(1235, np.arange(4096, dtype=np.uint64))
    assert len(set(a.tolist())) > 4090 and not np.array_equal(a, b)
    u = (a & np.uint32(0xFFFFFF)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 0.02


def test_scatter_table_is_the_inverse_of_the_pack_table():
    """The finishing kernel writes an updated weight straight to its packed positions: the scatter table must list, for
    every parameter, exactly the positions the pack table reads it from (fp32 fragments and the bf16 stream)."""
    from tnerf import trainer
    for cfg in ((39, 256, 8, 4), (63, 128, 4, 2), (10, 128, 2, 1)):
        d = _desc(*cfg)
        lib = tl.load()
        n_params = lib.tnerf_param_count(C.byref(d))
        sz = tl.PlanSizes()
        tl.call("tnerf_plan_sizes_query", C.byref(d), 64, 256, C.byref(sz))
        tab = np.empty(sz.packed_floats, np.int32)
        tl.call("tnerf_plan_fill", C.byref(d), 64, 256, _ptr(tab), None, None)
        tabs = [tab]
        if (cfg[0] - 3) % 6 == 0 and cfg[0] >= 9:
            bs = tl.Bf16Sizes()
            tl.call("tnerf_bf16_plan_sizes", C.byref(d), C.byref(bs))
            t16 = np.empty(int(bs.pack_entries), np.int32)
            tl.call("tnerf_bf16_pack_table", C.byref(d), _ptr(t16))
            tabs.append(t16)
        for t in tabs:
            sc = trainer._scatter_table(t, n_params)
            assert sc.shape[0] == n_params and 1 <= sc.shape[1] <= 4
            rebuilt = np.full_like(t, -1)
            for k in range(sc.shape[1]):
                ok = sc[:, k] >= 0
                rebuilt[sc[ok, k]] = np.nonzero(ok)[0]
                if k + 1 < sc.shape[1]:
                    assert not (sc[~ok, k + 1] >= 0).any()          # -1 terminated rows
            assert np.array_equal(rebuilt, t)
            assert (sc[:, 0] >= 0).all()                            # every parameter is packed somewhere


# ------------------------------------------------------------------------------- round 2: any hidden width up to 256
def _embed_true_params(cfg_true, hk, params_true):
    """Independent statement of the zero padding: true-width parameters placed into the flat vector of the hk-wide model
    (weights at [:Ht, :Ht] and — skip layer — the input columns behind column hk; everything else 0), and for every true
    parameter its index in that vector."""
    in_dim, ht, depth, skip_at = cfg_true
    flat, index = [], []
    off = 0
    fan_t, fan_k = in_dim, in_dim
    for l in range(depth):
        Wt, bt = params_true[2 * l].numpy(), params_true[2 * l + 1].numpy()
        Wk = np.zeros((hk, fan_k)); idx = -np.ones((hk, fan_k), np.int64)
        grid = off + np.arange(hk * fan_k).reshape(hk, fan_k)
        if l == 0:
            Wk[:ht, :] = Wt; sel = grid[:ht, :]
        else:
            Wk[:ht, :ht] = Wt[:, :ht]
            sel = grid[:ht, :ht]
            if fan_t > ht:                                  # skip layer: [hidden | input]
                Wk[:ht, hk:] = Wt[:, ht:]
                sel = np.concatenate([grid[:ht, :ht], grid[:ht, hk:]], axis=1)
        flat.append(Wk.reshape(-1)); index.append(sel.reshape(-1)); off += hk * fan_k
        bk = np.zeros(hk); bk[:ht] = bt
        flat.append(bk); index.append(off + np.arange(ht)); off += hk
        fan_t = ht + in_dim if (skip_at > 0 and l == skip_at - 1) else ht
        fan_k = hk + in_dim if (skip_at > 0 and l == skip_at - 1) else hk
    ws, bsig, wc, bc = (p.numpy() for p in params_true[2 * depth:])
    wk = np.zeros((1, hk)); wk[:, :ht] = ws
    flat.append(wk.reshape(-1)); index.append(off + np.arange(ht)); off += hk
    flat.append(bsig); index.append(off + np.arange(1)); off += 1
    wk = np.zeros((3, hk)); wk[:, :ht] = wc
    flat.append(wk.reshape(-1)); index.append((off + np.arange(3 * hk).reshape(3, hk))[:, :ht].reshape(-1)); off += 3 * hk
    flat.append(bc); index.append(off + np.arange(3)); off += 3
    return np.concatenate(flat), np.concatenate(index)


@pytest.mark.parametrize("cfg", [(39, 200, 3, 2), (63, 64, 4, 2), (39, 100, 2, 0), (27, 31, 3, 1)])
def test_any_hidden_width_runs_on_the_padded_kernels(cfg):
    """hidden = 31 / 64 / 100 / 200 (reference src/nerf.py:10 takes any width): the tables handed to the 128- / 256-wide
    kernels must be exactly the wide model's tables with the true parameters embedded and zeros elsewhere — pack tables
    (fp32 fragments, bf16 streams) and the slab -> gradient reduce tables (fp32 and bf16 plans)."""
    in_dim, ht, depth, skip_at = cfg
    hk = 128 if ht <= 128 else 256
    lib = tl.load()
    g = torch.Generator().manual_seed(11)
    params = O.mlp_init(in_dim, ht, depth, skip_at, g)
    flat_true = torch.cat([p.reshape(-1) for p in params]).double().numpy()
    d_t, d_k = _desc(in_dim, ht, depth, skip_at), _desc(in_dim, hk, depth, skip_at)
    assert lib.tnerf_param_count(C.byref(d_t)) == flat_true.shape[0]
    k = 2 * depth + 4
    off = np.zeros(k, np.int64); rows = np.zeros(k, np.int64); cols = np.zeros(k, np.int64)
    tl.call("tnerf_param_layout", C.byref(d_t), _ptr(off), _ptr(rows), _ptr(cols))
    assert [tuple(int(v) for v in rc) for rc in zip(rows, cols)] == [tuple(p.shape) if p.dim() == 2 else (p.shape[0], 1) for p in params]
    assert off.tolist() == np.concatenate([[0], np.cumsum([p.numel() for p in params])[:-1]]).tolist()
    flat_k, index = _embed_true_params(cfg, hk, [p.double() for p in params])
    assert index.shape[0] == flat_true.shape[0] and np.array_equal(flat_k[index], flat_true)
    M, n_cu = 96, 16
    sz_t, pack_t, jobs_t, red_t, _, _ = _plan(cfg, M, n_cu)
    sz_k, pack_k, jobs_k, red_k, _, _ = _plan((in_dim, hk, depth, skip_at), M, n_cu)
    assert sz_t.packed_floats == sz_k.packed_floats and sz_t.stash_floats == sz_k.stash_floats and sz_t.slab_floats == sz_k.slab_floats
    assert sz_t.n_params == flat_true.shape[0] and np.array_equal(jobs_t, jobs_k)
    got = np.where(pack_t >= 0, flat_true[np.clip(pack_t, 0, None)], 0.0)
    want = np.where(pack_k >= 0, flat_k[np.clip(pack_k, 0, None)], 0.0)
    assert np.array_equal(got, want)                                        # what the kernels read is the embedded model, zeros included
    HDR = 260
    assert np.array_equal(red_t[:HDR], red_k[:HDR])
    assert np.array_equal(red_t[HDR:].reshape(-1, 2), red_k[HDR:].reshape(-1, 2)[index])
    if in_dim >= 9 and (in_dim - 3) % 6 == 0:                               # bf16 mode
        tabs = []
        for d in (d_t, d_k):
            bs = tl.Bf16Sizes(); tl.call("tnerf_bf16_plan_sizes", C.byref(d), C.byref(bs))
            t16 = np.empty(int(bs.pack_entries), np.int32); tl.call("tnerf_bf16_pack_table", C.byref(d), _ptr(t16))
            tp = tl.Bf16TrainPlan(); tl.call("tnerf_bf16_train_sizes", C.byref(d), 5, 33, n_cu, C.byref(tp))
            jb = np.empty(tp.job_ints, np.int32); rd = np.empty(tp.reduce_ints, np.int32)
            tl.call("tnerf_bf16_train_fill", C.byref(d), 5, 33, n_cu, _ptr(jb), _ptr(rd))
            tabs.append((t16, jb, rd, tp.stash_bytes))
        (p_t, j_t, r_t, s_t), (p_k, j_k, r_k, s_k) = tabs
        assert p_t.shape == p_k.shape and s_t == s_k and np.array_equal(j_t, j_k)
        assert np.array_equal(np.where(p_t >= 0, flat_true[np.clip(p_t, 0, None)], 0.0), np.where(p_k >= 0, flat_k[np.clip(p_k, 0, None)], 0.0))
        assert np.array_equal(r_t[:HDR], r_k[:HDR]) and np.array_equal(r_t[HDR:].reshape(-1, 2), r_k[HDR:].reshape(-1, 2)[index])
