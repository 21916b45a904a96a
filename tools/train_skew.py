#!/usr/bin/env python3
"""Training step on a heavy-tailed (Chung-Lu) graph vs Erdos-Renyi with the same node / edge counts (75k / 1M, 4 samples):
the hub path of the backward over kept activations at scale."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, scipy.sparse as sp, torch
import gnode_oracle as O
from gnode import ops, _lib
from gnode.autograd import l1_loss_sum
from gnode.ode_nn_ngraph_sim import ODEBlock, ODEfunc
dev = torch.device("cuda:0")
n, m, B, H, T = 75000, 500000, 4, 64, 30
lib = _lib.load()
for kind in ("er", "chung-lu"):
    rp, ci, _ = (O.er_graph if kind == "er" else O.chung_lu_graph)(n, m, seed=0)
    A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
    model = ODEBlock(T, 0.5, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
    x = torch.from_numpy(O.make_samples(n, B, H, seed=2)).to(dev)
    y = torch.from_numpy(np.random.default_rng(0).dirichlet(np.ones(3), size=(B * n, T))).to(dev)
    rows = ops.subsample_rows(T, 0.5)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    def step():
        opt.zero_grad()
        S, I, R = model(x, out_rows=rows)
        loss = l1_loss_sum(S, I, R, y, 1) / (B * n * (T - 1) * 3)
        loss.backward(); opt.step()
        return loss
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(json.dumps({"graph": kind, "max_degree": int(np.diff(rp).max()), "train_step_ms": dt * 1e3, "loss_finite": bool(torch.isfinite(loss).item())}))
    del model, x, y
    torch.cuda.empty_cache()
