#!/usr/bin/env python3
"""One rank of the 2-rank DatasetTrainer rehearsal (tests/test_a_two_ranks_gpu.py starts two of these as fresh processes; both use
GPU 0, gradients exchanged over gloo).  Runs the sharded step with graphs (two captured graphs around the all-reduce) and without
(six eager launches) from identical initial weights and checks, bit for bit: losses, weights, packed copies.  Rank 0 prints one
JSON line.  argv: <out.json> <precision>"""
import json, os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tiny-nerf-pytorch_amd"), os.path.join(ROOT, "tiny-nerf-pytorch_amd", "src"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist
from conftest import golden_params

def main():
    out_path, prec = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    import nerf, data
    from tnerf import trainer as T
    cfg, params = golden_params("4x128")
    sc = data.make_synthetic_scene(n_images=5, H=20, W=20, focal=138.88887889922103 * 0.2, seed=4)
    images, poses, focal = torch.from_numpy(sc["images"]).to(dev), torch.from_numpy(sc["poses"]).to(dev), float(sc["focal"])
    Rg, S, seed, steps = 90, 40, 5, 6

    def run(graph, ws=None):
        m = nerf.TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"]).to(dev)
        with torch.no_grad():
            for p, v in zip(m.parameters(), params):
                p.copy_(v.to(dev))
        kw = {} if ws is None else dict(rank=0, world=1)
        t = T.DatasetTrainer(m, T.FlatAdam(m, lr=5e-4), images, poses, focal, Rg, S, 2.0, 6.0, seed=seed, precision=prec, graph=graph, **kw)
        losses = []
        for _ in range(steps):
            l, _ = t.step()
            l = l.clone()
            if ws is None:
                dist.all_reduce(l)                 # each rank's loss is its share of the global mean
            losses.append(float(l))
        torch.cuda.synchronize()
        return m, t, losses

    mg, tg, lg = run(True)
    me, te, le = run(False)
    assert tg._graph is not None and tg._graph_update is not None and te._graph is None, "graph mode did not capture two graphs"
    same_w = all(torch.equal(a, b) for a, b in zip(mg.parameters(), me.parameters()))
    same_pack = torch.equal(tg._packed, te._packed) and (tg._x3_packed is None or torch.equal(tg._x3_packed.view(torch.uint8), te._x3_packed.view(torch.uint8)))
    # both ranks hold the same weights (same all-reduced gradient, same update)
    flat = mg.hip_state().flat.clone()
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    same_ranks = all(torch.equal(o, flat) for o in other)
    # ... and they follow the one-rank trainer on the whole batch (different summation order of the shards: not bitwise)
    m1, t1, l1 = run(True, ws=1)
    dev_w = max(float((a - b).abs().max()) for a, b in zip(mg.parameters(), m1.parameters()))
    # the parity path (torch-drawn pixels and jitter, FusedTrainer.step_camera): every rank draws the SAME global batch and takes its
    # rows; the all-reduced gradient must train both ranks identically and follow a one-rank trainer on the whole batch
    from tnerf import dist as tdist
    N, H, W, _ = images.shape
    pixels = images.view(N, H * W, 3)

    def run_torch_rng(sharded):
        m = nerf.TinyNeRF(cfg["in_dim"], cfg["hidden"], cfg["depth"], cfg["skip_at"]).to(dev)
        with torch.no_grad():
            for p, v in zip(m.parameters(), params):
                p.copy_(v.to(dev))
        t = T.FusedTrainer(m, T.FlatAdam(m, lr=5e-4), 2.0, 6.0, S, precision=prec)
        gen = torch.Generator(device=dev); gen.manual_seed(77)
        lo, hi = tdist.shard_bounds(Rg, rank, world) if sharded else (0, Rg)
        for s_ in range(4):
            inds = torch.randint(0, H * W, (Rg,), device=dev, generator=gen)
            u = torch.rand(Rg, S, device=dev, generator=gen)
            if sharded:
                t.step_camera(poses[s_ % N], H, W, focal, inds[lo:hi], pixels[s_ % N], t_rand=u[lo:hi], global_rays=Rg)
            else:
                saved = tdist.all_reduce_sum_
                tdist.all_reduce_sum_ = lambda x: x            # a one-rank reference inside a two-rank group: no exchange
                T._dist.all_reduce_sum_ = tdist.all_reduce_sum_
                try:
                    t.step_camera(poses[s_ % N], H, W, focal, inds, pixels[s_ % N], t_rand=u, global_rays=Rg)
                finally:
                    tdist.all_reduce_sum_ = saved; T._dist.all_reduce_sum_ = saved
        torch.cuda.synchronize()
        return m
    ms, m1r = run_torch_rng(True), run_torch_rng(False)
    fl = ms.hip_state().flat.clone()
    oth = [torch.empty_like(fl) for _ in range(world)]
    dist.all_gather(oth, fl)
    parity_same_ranks = all(torch.equal(o, fl) for o in oth)
    parity_dev = max(float((a - b).abs().max()) for a, b in zip(ms.parameters(), m1r.parameters()))
    res = dict(parity_same_across_ranks=parity_same_ranks, parity_max_dev_vs_one_rank=parity_dev,
               rank=rank, world=world, precision=prec, losses_graph=lg, losses_eager=le, losses_one_rank=l1, same_losses=lg == le, same_weights=same_w,
               same_packed=same_pack, same_across_ranks=same_ranks, max_dev_vs_one_rank=dev_w, rays_local=tg.R,
               crc=zlib.crc32(flat.cpu().numpy().tobytes()))
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(res, f)
        print(json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    ok = res["same_losses"] and same_w and same_pack and same_ranks and parity_same_ranks
    sys.exit(0 if ok else 3)

main()
