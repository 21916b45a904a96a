"""CPU, world_size 2, gloo: the N>1 host path (sample sharding, Monte-Carlo count
reduction, flat gradient all-reduce).  The compute inside each rank is the CPU oracle
(there is no GPU here); what is under test is the sharding logic that bench.py and the
trainers run with one process per GPU over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gnode_oracle as O
    from gnode import sharding as sh
    try:
        n, B, H, maxTime, dT = 60, 5, 8, 4, 0.5
        rp, ci, _ = O.er_graph(n, 200, seed=1)
        P = O.init_params(H, seed=0)
        x = O.make_samples(n, B, H, seed=3)
        # ---- forward: contiguous sample blocks, no data-path collective; gather only to compare
        lo, hi = sh.shard_range(B, rank, world)
        S, I, R = O.odeblock_forward_single(x[lo:hi], P, rp, ci, maxTime, dT)
        sizes = [(sh.shard_range(B, r, world)[1] - sh.shard_range(B, r, world)[0]) * n for r in range(world)]
        S_all = sh.gather_rows(torch.from_numpy(S[..., 0]), sizes).numpy()
        S_ref, _, _ = O.odeblock_forward_single(x, P, rp, ci, maxTime, dT)
        ok_fwd = np.array_equal(S_all, S_ref[..., 0])
        # the same through the configs[3] helper (unequal blocks: 3 + 2 samples), and with FEWER samples than ranks (rank 1 owns none)
        fn = lambda xb: tuple(torch.from_numpy(a[..., 0]) for a in O.odeblock_forward_single(xb.numpy(), P, rp, ci, maxTime, dT))
        got = sh.sharded_forward(fn, torch.from_numpy(x))
        ok_fwd = ok_fwd and all(np.array_equal(g_.numpy(), r_[..., 0]) for g_, r_ in zip(got, O.odeblock_forward_single(x, P, rp, ci, maxTime, dT)))
        got1 = sh.sharded_forward(fn, torch.from_numpy(x[:1]))
        ok_fwd = ok_fwd and np.array_equal(got1[0].numpy(), O.odeblock_forward_single(x[:1], P, rp, ci, maxTime, dT)[0][..., 0])
        # ---- Monte-Carlo: sims range sharded, counts summed, row 0 restored
        sims, T = 37, 6
        slo, shi = sh.shard_range(sims, rank, world)
        c = O.sir_philox(n, rp, ci, [2, 9], 0.4, 0.2, shi - slo, T, rng_seed=77, sim_offset=slo)
        ct = torch.from_numpy(c.astype(np.int64))
        sh.allreduce_counts(ct)
        whole = O.sir_philox(n, rp, ci, [2, 9], 0.4, 0.2, sims, T, rng_seed=77)
        ok_mc = np.array_equal(ct.numpy().astype(np.uint32), whole)
        # fewer trajectories than ranks: rank 1's shard is EMPTY and must not disturb the assigned row 0
        slo, shi = sh.shard_range(1, rank, world)
        c1 = O.sir_philox(n, rp, ci, [2, 9], 0.4, 0.2, shi - slo, T, rng_seed=5, sim_offset=slo) if shi > slo else np.zeros((3, T, n), np.uint32)
        ct1 = torch.from_numpy(c1.astype(np.int64))
        sh.allreduce_counts(ct1)
        ok_mc = ok_mc and np.array_equal(ct1.numpy().astype(np.uint32), O.sir_philox(n, rp, ci, [2, 9], 0.4, 0.2, 1, T, rng_seed=5))
        # ---- flat gradient all-reduce
        w = torch.nn.Linear(4, 3)
        for p_ in w.parameters():
            p_.grad = torch.full_like(p_, float(rank + 1))
        sh.allreduce_flat_grads(list(w.parameters()), scale=0.5)
        ok_g = all(torch.allclose(p_.grad, torch.full_like(p_, 0.5 * sum(range(1, world + 1)))) for p_ in w.parameters())
        q.put((rank, ok_fwd, ok_mc, ok_g))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions():
    from gnode.sharding import shard_range
    for total in (0, 1, 7, 8, 64, 10000):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(180)
def test_world2_gloo_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, ok_fwd, ok_mc, ok_g in res:
        assert ok_fwd, f"rank {rank}: sharded forward != whole"
        assert ok_mc, f"rank {rank}: sharded Monte-Carlo counts != whole"
        assert ok_g, f"rank {rank}: flat gradient all-reduce wrong"
