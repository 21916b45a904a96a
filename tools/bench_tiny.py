#!/usr/bin/env python3
"""Tiny graphs (configs[0]: karate, 34 nodes, batch 1, maxTime 20): forward, training forward, adjoint backward and the
trainer's HIP-graph step.  One JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gn-ode-sir_amd"))
import numpy as np, scipy.sparse as sp, torch
from gnode import ops, synth
from gnode.graph import DeviceGraph
dev = torch.device("cuda:0")

def ev(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for n, m, B in ((34, 78, 1), (62, 159, 4), (34, 78, 8)):
    rp, ci = synth.er_csr(n, m, seed=1)
    g = DeviceGraph(rp, ci)
    P = {k: torch.from_numpy(v).to(dev) for k, v in synth.linear_params(64, seed=0).items()}
    x = torch.from_numpy(synth.samples(n, B, 64, seed=1)).to(dev).reshape(B * n, 67)
    dts = ops.step_sizes(ops.time_grid(20, 0.5))
    rows_out = ops.subsample_rows(20, 0.5)
    out = {"n": n, "B": B, "steps": len(dts), "train_path": ops.forward_path(g, B * n, 64, len(dts), len(rows_out), want_sol=True)}
    out["fwd_ms"] = round(ev(lambda: ops.forward(g, x, P, dts, "euler", None)), 4)
    out["train_fwd_ms"] = round(ev(lambda: ops.forward(g, x, P, dts, "euler", rows_out, want_sol=True)), 4)
    gs = [torch.randn(len(rows_out), B * n, device=dev) for _ in range(3)]
    S, I, R, sol = ops.forward(g, x, P, dts, "euler", rows_out, want_sol=True)
    out["bwd_ms"] = round(ev(lambda: ops.backward(g, x, P, dts, "euler", rows_out, sol, *gs)), 4)
    print(json.dumps(out), flush=True)
