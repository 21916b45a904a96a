#!/usr/bin/env python3
"""Timings of the two comparison baselines next to their CPU oracles."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, scipy.sparse as sp, torch
import gnode_oracle as O
from gnode import ode_nn
from gnode.dmp import DMP_SIR


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for name, n, m, T in (("karate-size", 34, 78, 20), ("wiki-vote-size", 7066, 100736, 30), ("epinions-size", 75000, 500000, 30)):
    rp, ci, _ = O.er_graph(n, m, seed=1)
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n, n))
    beta, gamma = 1.0 / max(1.0, len(ci) / n) * 0.5, 0.3
    model = DMP_SIR(A * beta, [gamma] * n)
    t_dmp = timed(lambda: model.run([0, 1], T))
    t_mf = timed(lambda: ode_nn.runge_kutta_order4(ode_nn.sir, A, n, [0, 1], beta, gamma, 0.5, T), reps=1)
    t0 = time.perf_counter(); O.meanfield_rk(rp, ci, [0, 1], beta, gamma, 0.5, T); t_mf_cpu = time.perf_counter() - t0
    row = {"case": name, "n": n, "nnz": int(len(ci)), "T": T, "dmp_gpu_ms": t_dmp * 1e3, "meanfield_gpu_ms": t_mf * 1e3,
           "meanfield_scipy_lsoda_sparse_ms": t_mf_cpu * 1e3}
    if n <= 8000:
        t0 = time.perf_counter(); O.dmp_sir(rp, ci, np.full(len(ci), beta, np.float32), np.full(n, gamma, np.float32), [0, 1], T)
        row["dmp_numpy_oracle_ms"] = (time.perf_counter() - t0) * 1e3
    print(json.dumps(row))
