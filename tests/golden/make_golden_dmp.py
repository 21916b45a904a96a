#!/usr/bin/env python3
"""Golden vectors for the DMP baseline, produced by running the REFERENCE class
`DMP_SIR` (dmp.py:74-170) unchanged in the build container.

`torch_scatter` and `torch_geometric` are absent from this image and stay absent.  dmp.py calls
`torch_scatter.scatter(src, index, reduce='mul', dim_size=...)` (dmp.py:93-99, 124, 141): the callable handed to it
here multiplies `src[e]` into `out[index[e]]` in ascending e, which is that library's documented CPU behaviour.
The vectors therefore pin the reference's own arithmetic around that call (edge list construction, cavity index,
update order, float32 rounding) but NOT the scatter itself: **parity unpinned** at that boundary.
No pickle of the reference is loaded: graphs come from networkx generators / numpy RNG.
"""
import os
import sys
import types

import numpy as np
import networkx as nx
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (import shims for ndlib / torchdiffeq)

REF = "/root/reference"


def main():
    MG._install_import_shims()
    ts = types.ModuleType("torch_scatter")

    def scatter(src, index, dim=-1, reduce="sum", dim_size=None):
        assert reduce == "mul"
        out = torch.ones(dim_size, dtype=src.dtype)
        for e in range(src.shape[0]):
            out[index[e]] = out[index[e]] * src[e]
        return out

    ts.scatter = scatter
    sys.modules["torch_scatter"] = ts
    tg, tgu = types.ModuleType("torch_geometric"), types.ModuleType("torch_geometric.utils")
    tgu.degree = None
    sys.modules["torch_geometric"], sys.modules["torch_geometric.utils"] = tg, tgu
    sys.path.insert(0, REF)
    import dmp as REFDMP
    torch.set_default_dtype(torch.float32)

    rng = np.random.default_rng(7)
    er = nx.gnm_random_graph(120, 500, seed=3)
    loops = nx.gnm_random_graph(40, 90, seed=5)
    loops.add_edges_from([(3, 3), (17, 17)])
    cases = [("karate", nx.karate_club_graph(), [0, 33], 0.3, 0.2, 20),
             ("er120", er, [5], 0.15, 0.35, 30),
             ("loops40", loops, [1, 2, 30], 0.5, 0.1, 12)]
    for name, G, seeds, beta, gamma, T in cases:
        A = sp.csr_matrix(nx.adjacency_matrix(G, nodelist=sorted(G.nodes()))).astype(np.float64)
        A.data[:] = 1.0
        A.sort_indices()
        gam = rng.uniform(0.5, 1.5, size=A.shape[0]) * gamma if name == "er120" else np.full(A.shape[0], gamma)
        m = REFDMP.DMP_SIR(A * beta, list(gam))
        out = m.run(list(seeds), T).numpy()
        np.savez_compressed(os.path.join(HERE, f"dmp_{name}.npz"), rowptr=A.indptr.astype(np.int32),
                            col=A.indices.astype(np.int32), weights=(A * beta).data.astype(np.float32),
                            gamma=np.asarray(gam, np.float32), seeds=np.asarray(seeds, np.int32), maxTime=np.int32(T),
                            out=out.astype(np.float32))
        print(name, out.shape, float(out.sum()))


if __name__ == "__main__":
    main()
