#!/usr/bin/env python3
"""Profile target: the drop-in trainer's HIP-graph step on a small graph (argv: n m B H maxTime; default karate size)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gn-ode-sir_amd"))
import numpy as np, scipy.sparse as sp, torch
from gnode import synth
from gnode.ode_nn_ngraph_sim import ODEBlock, ODEfunc
from gnode.trainer import Runner

n, m, B, H, maxTime = [int(v) for v in (sys.argv[1:6] + ["34", "78", "1", "64", "20"][len(sys.argv) - 1:])]
dev = torch.device("cuda:0")
rp, ci = synth.er_csr(n, m, seed=1)
A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
model = ODEBlock(maxTime, 0.5, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
xs = [torch.from_numpy(synth.samples(n, 1, H, seed=2 + j))[0] for j in range(8 * B)]
ys = [torch.from_numpy(np.random.default_rng(j).dirichlet(np.ones(3), size=(n, maxTime))) for j in range(8 * B)]
run = Runner(model, 1e-3, maxTime, 0.5, dev, stack=True, use_graphs=True)
xp, yp = run.place(xs, ys)
for ep in range(3):
    run.train_epoch(xp, yp, B, ep)
torch.cuda.synchronize()
