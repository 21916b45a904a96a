#!/usr/bin/env python3
"""Profile target: repeated forward on a karate-size graph (host vs GPU time split)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import gnode_oracle as O
from gnode import ops
from gnode.graph import DeviceGraph
n, m, B, H = 34, 78, int(sys.argv[1]) if len(sys.argv) > 1 else 1, 64
dev = torch.device("cuda:0")
rp, ci, _ = O.er_graph(n, m, seed=1)
g = DeviceGraph(rp, ci)
P = {k: torch.from_numpy(v).to(dev) for k, v in O.init_params(H, seed=0).items()}
x = torch.from_numpy(O.make_samples(n, B, H, seed=1)).to(dev).reshape(B * n, 3 + H)
dts = ops.step_sizes(ops.time_grid(20, 0.5))
for _ in range(3): ops.forward(g, x, P, dts)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): ops.forward(g, x, P, dts)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue per call %.1f us, total per call %.1f us" % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
