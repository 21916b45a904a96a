#!/usr/bin/env python3
"""Adjoint sweep of tiny graphs (one launch per sample batch): with the forward's kept activations vs recomputing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, torch
import gnode_oracle as O
from gnode import ops
from gnode.graph import DeviceGraph
dev = torch.device("cuda:0")
for name, n, m, B, T in [("karate B=1", 34, 78, 1, 20), ("karate B=32", 34, 78, 32, 20), ("dolphins B=4", 62, 159, 4, 20)]:
    rp, ci, _ = O.er_graph(n, m, seed=1)
    g = DeviceGraph(rp, ci)
    P = {k: torch.from_numpy(v).to(dev) for k, v in O.init_params(64, seed=0).items()}
    x = torch.from_numpy(O.make_samples(n, B, 64, seed=1)).to(dev).reshape(B * n, 67)
    dts = ops.step_sizes(ops.time_grid(T, 0.5))
    rows = ops.subsample_rows(T, 0.5)
    S, I, R, sol = ops.forward(g, x, P, dts, "euler", rows, want_sol=True)
    gs = [torch.randn_like(S) for _ in range(3)]
    out = {}
    for label, keep in (("kept", "auto"), ("recomputed", None)):
        for _ in range(3): ops.backward(g, x, P, dts, "euler", rows, sol, *gs, keep=keep)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.backward(g, x, P, dts, "euler", rows, sol, *gs, keep=keep)
        e1.record(); torch.cuda.synchronize()
        out[label] = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name}: backward {out['kept']:.0f} us over kept activations, {out['recomputed']:.0f} us recomputing (keep buffer: {sol.gnode_keep is not None})")
