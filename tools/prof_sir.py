#!/usr/bin/env python3
"""rocprof target: the Monte-Carlo kernel on configs[2]'s shape (wiki-vote size, 10 000 sims x T = 20).  argv: beta gamma"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gn-ode-sir_amd"))
import torch
from gnode import synth
from gnode.graph import DeviceGraph
from gnode.ode_nn import sir_counts
beta, gamma = (float(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else (0.3, 0.2)
n, m = 7066, 100736
rp, ci = synth.er_csr(n, m, seed=0)
g = DeviceGraph(rp, ci)
sir_counts(g, [1, n // 2], beta, gamma, 64, 20, rng_seed=1)
torch.cuda.synchronize()
for _ in range(3):
    sir_counts(g, [1, n // 2], beta, gamma, 10000, 20, rng_seed=2)
torch.cuda.synchronize()
