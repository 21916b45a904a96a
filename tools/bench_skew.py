#!/usr/bin/env python3
"""Degree-skew check: forward time on a Chung-Lu (power-law-like) graph vs an Erdos-Renyi graph with the
same node and edge counts.  One JSON line per case."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import gnode_oracle as O
from gnode import ops
from gnode.graph import DeviceGraph
dev = torch.device("cuda:0")
for n, m, B in ((7066, 100736, 8), (75000, 500000, 4)):
    for kind in ("er", "chung-lu"):
        rp, ci, _ = (O.er_graph if kind == "er" else O.chung_lu_graph)(n, m, seed=0)
        deg = np.diff(rp)
        g = DeviceGraph(rp, ci)
        for H in (64, 8):
            P = {k: torch.from_numpy(v).to(dev) for k, v in O.init_params(H, seed=0).items()}
            x = torch.from_numpy(O.make_samples(n, B, H, seed=1)).to(dev).reshape(B * n, 3 + H)
            dts = ops.step_sizes(ops.time_grid(30, 0.5))
            ops.forward(g, x, P, dts); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                ops.forward(g, x, P, dts)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            print(json.dumps({"graph": kind, "n": n, "nnz": int(ci.shape[0]), "max_degree": int(deg.max()), "B": B, "H": H,
                              "forward_ms": dt * 1e3, "us_per_step": dt * 1e6 / 59, "node_timesteps_per_s": B * n * 59 / dt}))
