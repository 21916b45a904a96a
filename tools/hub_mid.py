#!/usr/bin/env python3
"""Forward / training-forward time per Euler step on mid-size heavy-tailed graphs whose longest hub is a few hundred to
two thousand edges (the regime of the reference's real datasets): the one-launch hub sums vs -DGN_HUB_FUSED_SEGS=0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import gnode_oracle as O
from gnode import ops
from gnode.graph import DeviceGraph
dev = torch.device("cuda:0")
for n, m, ex in ((1893, 13835, 0.8), (7066, 100736, 0.6)):
    rp, ci, _ = O.chung_lu_graph(n, m, exponent=ex, seed=0)
    deg = np.diff(rp)
    g = DeviceGraph(rp, ci)
    P = {k: torch.from_numpy(v).to(dev) for k, v in O.init_params(64, seed=0).items()}
    dts = ops.step_sizes(ops.time_grid(30, 0.5))
    for B in (1, 8):
        x = torch.from_numpy(O.make_samples(n, B, 64, seed=1)).to(dev).reshape(B * n, 67)
        res = []
        for want_sol in (False, True):
            for _ in range(3): ops.forward(g, x, P, dts, "euler", None, want_sol)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): ops.forward(g, x, P, dts, "euler", None, want_sol)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 10 / 59 * 1e6)
        print(f"n={n} max_degree={int(deg.max())} hubs={(deg > 96).sum()} B={B}: inference {res[0]:.1f} us/step, training forward {res[1]:.1f} us/step")
