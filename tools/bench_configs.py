#!/usr/bin/env python3
"""Secondary measurements on the shapes of BASELINE.json configs[0..2] and [4] (the headline
metric lives in bench.py): forward latency and one full training step (forward + L1 loss +
adjoint backward + Adam) through the reference's call surface, next to the CPU port of the
reference's op sequence (forward only, bounded).  One JSON line per case."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np
import scipy.sparse as sp
import torch

import gnode_oracle as O
from gnode import ops
from gnode.ode_nn_ngraph_sim import ODEBlock, ODEfunc
from gnode import ode_nn_ngraphs as multi

dev = torch.device("cuda:0")
torch.set_num_threads(O.usable_cores())


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def single(name, n, m, B, H, maxTime, reps):
    rp, ci, _ = O.er_graph(n, m, seed=1)
    A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
    torch.manual_seed(0)
    model = ODEBlock(maxTime, 0.5, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
    x_np = O.make_samples(n, B, H, seed=2)
    x = torch.from_numpy(x_np).to(dev)
    y = torch.from_numpy(np.random.default_rng(0).dirichlet(np.ones(3), size=(B * n, maxTime))).to(dev)
    rows = ops.subsample_rows(maxTime, 0.5)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)

    def fwd():
        with torch.no_grad():
            model(x, out_rows=rows)

    def train():
        opt.zero_grad()
        S, I, R = model(x, out_rows=rows)
        pred = torch.cat((S, I, R), -1).transpose(0, 1)[:, 1:, :]
        (pred.double() - y[:, 1:, :]).abs().mean().backward()
        opt.step()

    tf, tt = timed(fwd, reps), timed(train, max(2, reps // 2))
    # the trainer's epoch loop (batch of B per optimiser step), eager launches vs HIP-graph replay
    from gnode.trainer import Runner
    xs = [x.cpu()[b] for b in range(B)] * 4
    ys = [y.cpu()[b * n:(b + 1) * n] for b in range(B)] * 4
    loop = {}
    for mode in (False, True):
        run = Runner(model, 1e-3, maxTime, 0.5, dev, stack=True, use_graphs=mode)
        xs, ys = run.place(xs, ys)
        run.train_epoch(xs, ys, B, 0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ep in range(3):
            run.train_epoch(xs, ys, B, ep)
        torch.cuda.synchronize()
        loop[mode] = (time.perf_counter() - t0) / (3 * 4)
    P = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if "ln." not in k}
    steps = len(ops.time_grid(maxTime, 0.5)) - 1
    cs = min(steps, 8)
    _, _, _, secs, done = O.torch_port_forward(x_np, P, rp, ci, maxTime, 0.5, n_steps=cs, threads=O.usable_cores())
    print(json.dumps({"case": name, "n": n, "nnz": int(ci.shape[0]), "B": B, "H": H, "euler_steps": steps,
                      "forward_ms": tf * 1e3, "train_step_ms": tt * 1e3, "trainer_step_eager_ms": loop[False] * 1e3,
                      "trainer_step_graph_ms": loop[True] * 1e3, "fwd_node_timesteps_per_s": B * n * steps / tf,
                      "cpu_port_fwd_node_timesteps_per_s": B * n * done / secs, "cpu_threads": O.usable_cores()}))


def multigraph(reps):
    sizes = [(62, 159), (620, 2102), (1893, 13835), (2905, 15645), (7066, 100736)]
    csr = [O.er_graph(n, m, seed=n)[:2] for n, m in sizes]
    A_list = [sp.csr_matrix((np.ones(c.shape[0]), c, r), shape=(len(r) - 1, len(r) - 1)) for r, c in csr]
    H, maxTime = 8, 20
    torch.manual_seed(0)
    model = multi.ODEBlock(maxTime, 0.5, H, multi.ODEfunc(A_list, H, dev), dev).to(dev)
    picks = [0, 1, 2, 3, 4, 2, 1, 4]                       # batch_size 8 (monitorer-ngraphs.py:10)
    xs = []
    for j, p in enumerate(picks):
        xi = O.make_samples(sizes[p][0], 1, H, seed=j)[0]
        xi[0, 5] = p + 1
        xs.append(xi)
    x = torch.from_numpy(np.concatenate(xs, 0)).to(dev)
    tot = x.shape[0]
    y = torch.from_numpy(np.random.default_rng(0).dirichlet(np.ones(3), size=(tot, maxTime))).to(dev)
    rows = ops.subsample_rows(maxTime, 0.5)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)

    def fwd():
        with torch.no_grad():
            model(x, out_rows=rows)

    def train():
        opt.zero_grad()
        S, I, R = model(x, out_rows=rows)
        pred = torch.cat((S, I, R), -1).transpose(0, 1)[:, 1:, :]
        (pred.double() - y[:, 1:, :]).abs().mean().backward()
        opt.step()

    tf, tt = timed(fwd, reps), timed(train, max(2, reps // 2))
    print(json.dumps({"case": "config5-like multi-graph batch of 8 (dolphins..wiki-vote sizes), H=8", "sum_nodes": tot, "H": H,
                      "euler_steps": 39, "forward_ms": tf * 1e3, "train_step_ms": tt * 1e3,
                      "fwd_node_timesteps_per_s": tot * 39 / tf}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "karate":
        single("config0-like karate-size, B=1 (monitorer-sim batch_size)", 34, 78, 1, 64, 20, 40)
        single("dolphins-size, B=4", 62, 159, 4, 64, 20, 40)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mid":
        single("config1-like fb-social-size, B=1", 1893, 13835, 1, 64, 30, 20)
        single("config2-like wiki-vote-size, B=1", 7066, 100736, 1, 64, 30, 10)
        sys.exit(0)
    single("config0-like karate-size, B=1 (monitorer-sim batch_size)", 34, 78, 1, 64, 20, 20)
    single("config1-like fb-social-size, B=1", 1893, 13835, 1, 64, 30, 10)
    single("config1-like fb-social-size, B=8", 1893, 13835, 8, 64, 30, 10)
    single("config2-like wiki-vote-size, B=1", 7066, 100736, 1, 64, 30, 6)
    single("epinions-size, B=1 (train step needs the 4.6 GB trajectory)", 75000, 500000, 1, 64, 30, 3)
    multigraph(10)
