"""The N>1 path on CPU: two gloo ranks shard the rays, all-reduce the flat gradient and must land on the
single-process result.  The per-rank compute engine here is the CPU oracle — the product's
engine is the HIP library; what is under test is the sharding / normalisation / all-reduce / gather logic
of tiny-nerf-pytorch_amd/tnerf/dist.py that both share."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_params, load_golden
from oracle import tnerf_oracle as O
from tnerf import dist as tdist


def test_shard_bounds_cover_rows_once():
    for n in (0, 1, 7, 4096, 10000):
        for ws in (1, 2, 3, 8):
            spans = [tdist.shard_bounds(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert tdist.world() == (0, 1)
    t = torch.ones(3)
    assert tdist.all_reduce_sum_(t) is t and tdist.all_gather_rows(t[:, None], 3).shape == (3, 1)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        cfg, params = golden_params("4x128")
        g = load_golden("step_4x128")
        images, poses, focal = g["images"], g["poses"], g["focal"]
        N, H, W, _ = images.shape
        ro_all, rd_all = O.pinhole_rays(H, W, focal, poses[0])
        inds, u = g["inds"][0], g["u"][0]                      # every rank holds the same global draw
        R, S = inds.shape[0], u.shape[-1]
        ro, rd, tgt = ro_all[inds], rd_all[inds], images.reshape(N, H * W, 3)[0, inds]
        lo, hi = tdist.shard_bounds(R, rank, world)
        loss_l, _, grads_l = O.loss_and_grads(params, cfg["skip_at"], cfg["L"], ro[lo:hi], rd[lo:hi], tgt[lo:hi],
                                              2.0, 6.0, S, u[lo:hi], loss_denominator=3 * R)
        flat = torch.cat([x.reshape(-1) for x in grads_l])
        tdist.all_reduce_sum_(flat)
        loss = loss_l.clone(); dist.all_reduce(loss)
        loss_f, _, grads_f = O.loss_and_grads(params, cfg["skip_at"], cfg["L"], ro, rd, tgt, 2.0, 6.0, S, u)
        full = torch.cat([x.reshape(-1) for x in grads_f])
        rel = float((flat - full).abs().max() / full.abs().max())
        # identical Adam on every rank
        ps = [p.clone() for p in params]
        views, o = [], 0
        for p in ps:
            views.append(flat[o:o + p.numel()].view(p.shape)); o += p.numel()
        O.AdamState(ps, lr=5e-4).step(ps, views)
        psum = torch.stack([p.double().sum() for p in ps])
        gathered = [torch.zeros_like(psum) for _ in range(world)]
        dist.all_gather(gathered, psum)
        # sharded render + all_gather_rows == full render
        comp_l, _, _, _ = O.render_rays(params, cfg["skip_at"], cfg["L"], ro[lo:hi], rd[lo:hi], 2.0, 6.0, S, None)
        comp = tdist.all_gather_rows(comp_l, R)
        comp_f, _, _, _ = O.render_rays(params, cfg["skip_at"], cfg["L"], ro, rd, 2.0, 6.0, S, None)
        tdist.broadcast_(flat, src=0)
        if rank == 0:
            torch.save(dict(rel=rel, loss=float(loss), loss_full=float(loss_f), same=bool(all(torch.equal(gathered[0], x) for x in gathered)),
                            render=float((comp - comp_f).abs().max()), rows=comp.shape[0], world=tdist.world()[1]), out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_reproduce_single_process(tmp_path):
    out = str(tmp_path / "r.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["world"] == 2 and r["rows"] == 256
    assert r["rel"] <= 1e-5, r                      # sum of shard gradients == full-batch gradient
    assert abs(r["loss"] - r["loss_full"]) <= 1e-6 * r["loss_full"]
    assert r["same"]                                # parameters stay identical across ranks
    assert r["render"] <= 1e-6
