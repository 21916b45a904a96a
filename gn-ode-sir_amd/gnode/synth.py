"""Synthetic inputs of the benchmark shapes: graph, weights and samples for bench.py and the tools.
Independent of `oracle/` (which only tests, smoke() and bench.py's cpu_baseline leg may touch)."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def er_csr(n: int, m: int, seed: int = 0):
    """Erdos-Renyi G(n, m): m undirected edges, no self-loops, no duplicates, `numpy.random.default_rng(seed)`;
    symmetrised CSR int32 with sorted columns -> (rowptr, col)."""
    rng = np.random.default_rng(seed)
    keys = np.empty(0, dtype=np.int64)
    while keys.shape[0] < m:
        k = int((m - keys.shape[0]) * 1.2) + 16
        u, v = rng.integers(0, n, k, dtype=np.int64), rng.integers(0, n, k, dtype=np.int64)
        keep = u != v
        keys = np.unique(np.concatenate([keys, np.minimum(u, v)[keep] * n + np.maximum(u, v)[keep]]))
        if keys.shape[0] > m:
            keys = np.sort(rng.permutation(keys)[:m])
    a, b = keys // n, keys % n
    A = sp.coo_matrix((np.ones(2 * m, dtype=np.int8), (np.concatenate([a, b]), np.concatenate([b, a]))), shape=(n, n)).tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32)


def heavy_tail_csr(n: int, m: int, exponent: float = 0.5, seed: int = 0):
    """A graph with the degree tail of the reference's real datasets (fb-social's longest row ~700 of 1 893 nodes,
    wiki-vote's ~1 065 of 7 066: SURVEY section 7): m distinct undirected edges whose endpoints are drawn with probability
    ~ (rank + 1)^-exponent (Chung-Lu), no self-loops; the big nodes come first, as in many real node numberings.
    Symmetrised CSR int32 with sorted columns -> (rowptr, col)."""
    rng = np.random.default_rng(seed)
    w = (np.arange(n) + 1.0) ** (-exponent)
    p = w / w.sum()
    keys = np.empty(0, dtype=np.int64)
    while keys.shape[0] < m:
        k = int((m - keys.shape[0]) * 1.5) + 64
        u, v = rng.choice(n, size=k, p=p), rng.choice(n, size=k, p=p)
        keep = u != v
        keys = np.unique(np.concatenate([keys, np.minimum(u, v)[keep].astype(np.int64) * n + np.maximum(u, v)[keep]]))
        if keys.shape[0] > m:
            keys = np.sort(rng.permutation(keys)[:m])
    a, b = keys // n, keys % n
    A = sp.coo_matrix((np.ones(2 * m, dtype=np.int8), (np.concatenate([a, b]), np.concatenate([b, a]))), shape=(n, n)).tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32)


def linear_params(H: int, seed: int = 0):
    """The eight trained tensors with `nn.Linear`'s default init U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (the reference
    never calls its init_weights, ode_nn_ngraph_sim.py:52,137), from a numpy generator."""
    rng = np.random.default_rng(seed)
    shapes = {"odefunc.linear.weight": (H, H), "odefunc.linear.bias": (H,), "linearS1.weight": (H, 1), "linearS1.bias": (H,),
              "linear3.weight": (4, H), "linear3.bias": (4,), "linearS2.weight": (1, 4), "linearS2.bias": (1,)}
    out = {}
    for name, shp in shapes.items():
        fan_in = shp[1] if len(shp) == 2 else shapes[name.replace("bias", "weight")][1]
        out[name] = rng.uniform(-1.0 / np.sqrt(fan_in), 1.0 / np.sqrt(fan_in), size=shp).astype(np.float32)
    return out


def samples(n: int, B: int, H: int, seed: int = 0, n_seeds: int = 2):
    """x [B, n, 3+H] as ode_nn_ngraph_sim.py:382-390 builds it: S0 | I0 | R0, then an H-wide slab whose column 0 is
    beta and column 1 gamma, both U(0.1, 0.5) (monitorer-sim.py:116-119)."""
    rng = np.random.default_rng(seed)
    x = np.zeros((B, n, 3 + H), dtype=np.float32)
    for b in range(B):
        s = rng.choice(n, size=min(n_seeds, n), replace=False)
        x[b, :, 0] = 1.0
        x[b, s, 0] = 0.0
        x[b, s, 1] = 1.0
        x[b, :, 3] = rng.uniform(0.1, 0.5)
        x[b, :, 4] = rng.uniform(0.1, 0.5)
    return x
