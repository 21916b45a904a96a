#!/usr/bin/env python3
"""Profile target: training steps (forward with trajectory + adjoint backward) on one shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, scipy.sparse as sp, torch
import gnode_oracle as O
from gnode import ops
from gnode.autograd import l1_loss_sum
from gnode.ode_nn_ngraph_sim import ODEBlock, ODEfunc

n, m, B, H, maxTime = [int(v) for v in (sys.argv[1:6] + ["75000", "500000", "1", "64", "30"][len(sys.argv) - 1:])]
dev = torch.device("cuda:0")
rp, ci, _ = O.er_graph(n, m, seed=1)
A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
model = ODEBlock(maxTime, 0.5, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
x = torch.from_numpy(O.make_samples(n, B, H, seed=2)).to(dev)
y = torch.from_numpy(np.random.default_rng(0).dirichlet(np.ones(3), size=(B * n, maxTime))).to(dev)
rows = ops.subsample_rows(maxTime, 0.5)
for _ in range(3):
    S, I, R = model(x, out_rows=rows)
    (l1_loss_sum(S, I, R, y, 1) / (B * n * (maxTime - 1) * 3)).backward()
torch.cuda.synchronize()
