#!/usr/bin/env python3
"""The drop-in trainer's optimiser step (gnode/trainer.py Runner: forward + L1 + adjoint sweep + Adam), eager launches vs
HIP-graph replay, on one shape.  argv: n m B [H maxTime]   (default: configs[1]'s bench shape 1893 13835 8)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gn-ode-sir_amd"))
import numpy as np, scipy.sparse as sp, torch
from gnode import synth
from gnode.ode_nn_ngraph_sim import ODEBlock, ODEfunc
from gnode.trainer import Runner

n, m, B, H, maxTime = [int(v) for v in (sys.argv[1:6] + ["1893", "13835", "8", "64", "30"][len(sys.argv) - 1:])]
tail = float(os.environ.get("TAIL", "0"))
dev = torch.device("cuda:0")
rp, ci = synth.heavy_tail_csr(n, m, tail, seed=1) if tail else synth.er_csr(n, m, seed=1)
A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
out = {"n": n, "nnz": int(ci.shape[0]), "B": B, "H": H, "maxTime": maxTime, "longest_row": int(np.diff(rp).max())}
for mode in (True, False):
    model = ODEBlock(maxTime, 0.5, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
    xs = [torch.from_numpy(synth.samples(n, 1, H, seed=2 + j))[0] for j in range(4 * B)]
    ys = [torch.from_numpy(np.random.default_rng(j).dirichlet(np.ones(3), size=(n, maxTime))) for j in range(4 * B)]
    run = Runner(model, 1e-3, maxTime, 0.5, dev, stack=True, use_graphs=mode)
    xp, yp = run.place(xs, ys)
    run.train_epoch(xp, yp, B, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for ep in range(reps):
        run.train_epoch(xp, yp, B, ep)
    torch.cuda.synchronize()
    out["trainer_step_%s_ms" % ("hip_graph" if mode else "eager")] = round((time.perf_counter() - t0) / (reps * 4) * 1e3, 4)
    del run, model
print(json.dumps(out))
