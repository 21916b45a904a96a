#!/usr/bin/env python3
"""Forward latency of mid-size graphs: eager launches vs one HIP-graph replay of the same 60 launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import gnode_oracle as O
from gnode import ops
from gnode.graph import DeviceGraph
dev = torch.device("cuda:0")
for name, n, m, B, T in [("karate", 34, 78, 1, 20), ("fb", 1893, 13835, 1, 30), ("fb x8", 1893, 13835, 8, 30), ("wiki", 7066, 100736, 1, 30)]:
    rp, ci, _ = O.er_graph(n, m, seed=1)
    g = DeviceGraph(rp, ci)
    P = {k: torch.from_numpy(v).to(dev) for k, v in O.init_params(64, seed=0).items()}
    x = torch.from_numpy(O.make_samples(n, B, 64, seed=1)).to(dev).reshape(B * n, 67)
    dts = ops.step_sizes(ops.time_grid(T, 0.5))
    rows = ops.subsample_rows(T, 0.5)
    for _ in range(3): ops.forward(g, x, P, dts, "euler", rows)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): ops.forward(g, x, P, dts, "euler", rows)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            out = ops.forward(g, x, P, dts, "euler", rows)
    torch.cuda.synchronize()
    gr.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): gr.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / 50
    print(f"{name}: eager {eager*1e6:.0f} us, graph replay {rep*1e6:.0f} us")
