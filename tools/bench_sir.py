#!/usr/bin/env python3
"""Monte-Carlo SIR label generator (A8) throughput: trajectory-steps/s = sims*(T-1)/s.
Secondary measurement (bench.py carries the headline metric).  Prints one JSON line per case."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np
import torch

import gnode_oracle as O
import oracle_c as OC
from gnode.graph import DeviceGraph
from gnode.ode_nn import sir_counts, sir_counts_counted


def main():
    cases = [("wiki-vote-sized", 7066, 100736, 10000, 20), ("fb-social-sized", 1893, 13835, 10000, 20),
             ("epinions-sized", 75000, 500000, 2000, 30)]
    points = [(0.3, 0.2), (0.05, 0.1)]                     # (beta, gamma): a fast burn-through, and a long-lived frontier
    for (name, n, m, sims, T), (beta, gamma) in [(c, p) for c in cases for p in points]:
        rp, ci, _ = O.er_graph(n, m, seed=0)
        g = DeviceGraph(rp, ci)
        seeds = [1, n // 2]
        sir_counts(g, seeds, beta, gamma, 64, T, rng_seed=1)            # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cnt = sir_counts(g, seeds, beta, gamma, sims, T, rng_seed=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        csims = max(16, sims // 50)
        t1 = time.perf_counter()
        want = OC.sir_philox(n, rp, ci, seeds, beta, gamma, csims, T, rng_seed=2)
        cdt = time.perf_counter() - t1
        ok = bool(np.array_equal(sir_counts(g, seeds, beta, gamma, csims, T, rng_seed=2).cpu().numpy().astype(np.uint32), want))
        infected_frac = float(1.0 - cnt[0, T - 1].float().mean().item() / sims)
        _, st = sir_counts_counted(g, seeds, beta, gamma, sims, T, rng_seed=2)
        print(json.dumps({"case": name, "beta": beta, "gamma": gamma, "counted": st, "n": n, "nnz": int(ci.shape[0]), "sims": sims, "T": T, "gpu_s": dt,
                          "gpu_traj_steps_per_s": sims * (T - 1) / dt, "gpu_edge_visits_per_s": sims * (T - 1) * ci.shape[0] / dt,
                          "cpu_port_traj_steps_per_s": csims * (T - 1) / cdt, "cpu_threads": O.usable_cores(),
                          "bit_exact_vs_oracle": ok, "final_attack_rate": infected_frac}))


if __name__ == "__main__":
    main()
