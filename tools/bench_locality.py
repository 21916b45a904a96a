#!/usr/bin/env python3
"""Does node order matter?  A community-structured graph (stochastic block model, 300 blocks of 250 nodes,
80 % of the edges inside blocks) with its nodes (a) in block order, (b) randomly relabelled, (c) relabelled and
then re-ordered with scipy's reverse Cuthill-McKee.  Same graph, same work; only the L2 locality of the gather
changes."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np, scipy.sparse as sp, torch
from scipy.sparse.csgraph import reverse_cuthill_mckee
import gnode_oracle as O
from gnode import ops
from gnode.graph import DeviceGraph
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
nb, bs = 300, 250
n = nb * bs
m_in, m_out = 400000, 100000
blk = rng.integers(0, nb, m_in)
u = blk * bs + rng.integers(0, bs, m_in); v = blk * bs + rng.integers(0, bs, m_in)
uo = rng.integers(0, n, m_out); vo = rng.integers(0, n, m_out)
e = np.stack([np.concatenate([u, uo]), np.concatenate([v, vo])], 1)
e = e[e[:, 0] != e[:, 1]]
def csr(edges): return O.csr_from_edges(n, edges)
perm = rng.permutation(n)
variants = {"block order": e, "random labels": perm[e]}
rp_s, ci_s = csr(perm[e])
A = sp.csr_matrix((np.ones(ci_s.shape[0]), ci_s, rp_s), shape=(n, n))
rcm = reverse_cuthill_mckee(A, symmetric_mode=True)
inv = np.empty(n, dtype=np.int64); inv[rcm] = np.arange(n)
variants["random labels + RCM"] = inv[perm[e]]
B, H = 8, 64
P = {k: torch.from_numpy(w).to(dev) for k, w in O.init_params(H, seed=0).items()}
x = torch.from_numpy(O.make_samples(n, B, H, seed=1)).to(dev).reshape(B * n, 3 + H)
dts = ops.step_sizes(ops.time_grid(30, 0.5))
for name, ed in variants.items():
    rp, ci = csr(ed)
    g = DeviceGraph(rp, ci)
    ops.forward(g, x, P, dts); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): ops.forward(g, x, P, dts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(json.dumps({"order": name, "nnz": int(ci.shape[0]), "max_degree": int(np.diff(rp).max()), "us_per_step": dt * 1e6 / 59,
                      "node_timesteps_per_s": B * n * 59 / dt}))
