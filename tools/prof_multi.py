#!/usr/bin/env python3
"""Profile target: training steps on a multi-graph batch of 8 (BASELINE configs[4] shape, H=8)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, scipy.sparse as sp, torch
import gnode_oracle as O
from gnode import ops
from gnode import ode_nn_ngraphs as multi

dev = torch.device("cuda:0")
sizes = [(62, 159), (620, 2102), (1893, 13835), (2905, 15645), (7066, 100736)]
csr = [O.er_graph(n, m, seed=n)[:2] for n, m in sizes]
A_list = [sp.csr_matrix((np.ones(c.shape[0]), c, r), shape=(len(r) - 1, len(r) - 1)) for r, c in csr]
H, maxTime = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 20
torch.manual_seed(0)
model = multi.ODEBlock(maxTime, 0.5, H, multi.ODEfunc(A_list, H, dev), dev).to(dev)
picks = [0, 1, 2, 3, 4, 2, 1, 4]
xs = []
for j, p in enumerate(picks):
    xi = O.make_samples(sizes[p][0], 1, H, seed=j)[0]
    xi[0, 5] = p + 1
    xs.append(xi)
x = torch.from_numpy(np.concatenate(xs, 0)).to(dev)
y = torch.from_numpy(np.random.default_rng(0).dirichlet(np.ones(3), size=(x.shape[0], maxTime))).to(dev)
rows = ops.subsample_rows(maxTime, 0.5)
for _ in range(3):
    S, I, R = model(x, out_rows=rows)
    pred = torch.cat((S, I, R), -1).transpose(0, 1)[:, 1:, :]
    (pred.double() - y[:, 1:, :]).abs().mean().backward()
torch.cuda.synchronize()
