#!/usr/bin/env python3
"""Forward throughput vs hidden size on the 75k-node benchmark graph (generic path for H != 64)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import gnode_oracle as O
from gnode import ops
from gnode.graph import DeviceGraph
dev = torch.device("cuda:0")
n, m, B = 75000, 500000, 8
rp, ci, _ = O.er_graph(n, m, seed=0)
g = DeviceGraph(rp, ci)
for H in ([int(v) for v in sys.argv[1:]] or (8, 16, 32, 64, 128)):
    P = {k: torch.from_numpy(v).to(dev) for k, v in O.init_params(H, seed=0).items()}
    x = torch.from_numpy(O.make_samples(n, B, H, seed=1)).to(dev).reshape(B * n, 3 + H)
    dts = ops.step_sizes(ops.time_grid(30, 0.5))
    ops.forward(g, x, P, dts); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ops.forward(g, x, P, dts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    alg = (ci.shape[0] * 4 + ci.shape[0] * H * 4 + 8 * n * H * 4) * B * 59
    print(json.dumps({"H": H, "forward_ms": dt * 1e3, "us_per_step": dt * 1e6 / 59, "node_timesteps_per_s": B * n * 59 / dt,
                      "algorithmic_TBps": alg / dt / 1e12}))
