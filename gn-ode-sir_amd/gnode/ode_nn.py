"""Drop-in for the helpers of reference ode_nn.py that sit on the hot path:

    sir_torch(G, seed_set, beta, gamma, sims=10000, T=20)        reference :30-88
    get_sir_t_nodes_torch(x_rk, maxTime, deltaT, count=True)     reference :249-261
    create_graph(n_nodes, graph_label='none')                    reference :394-414
    sir(x, y, A, beta, gamma), runge_kutta_order4(sir, A, ...)   reference :214-233 (mean-field comparison column)
"""
from __future__ import annotations

import ctypes as C
import os
import pickle

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib
from .graph import DeviceGraph


def _edge_arrays(G):
    if hasattr(G, "edge_array"):                      # CsrGraph (restored from the CSR cache file)
        return G.edge_array
    e = np.asarray(list(G.edges()), dtype=np.int64).reshape(-1, 2)
    return e


def _csr_from_edges(n, e):
    r = np.concatenate([e[:, 0], e[:, 1]])
    c = np.concatenate([e[:, 1], e[:, 0]])
    a = sp.coo_matrix((np.ones(r.shape[0], dtype=np.int8), (r, c)), shape=(n, n)).tocsr()
    a.sum_duplicates()
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32)


_GRAPH_CACHE: dict = {}


def _device_graph_for(G):
    """The label loop calls sir_torch once per (seed set, beta, gamma) on the SAME networkx graph
    (ode_nn_ngraph_sim.py:360-368): build and upload its CSR once, not 200 times."""
    import weakref
    key = (id(G), G.number_of_nodes(), G.number_of_edges())
    hit = _GRAPH_CACHE.get(key)
    if hit is not None and hit[0]() is G:            # same live object, not a recycled id
        return hit[1]
    if len(_GRAPH_CACHE) >= 4:
        _GRAPH_CACHE.pop(next(iter(_GRAPH_CACHE)))
    dg = DeviceGraph(*_csr_from_edges(G.number_of_nodes(), _edge_arrays(G)))
    try:
        _GRAPH_CACHE[key] = (weakref.ref(G), dg)
    except TypeError:                                # not weak-referenceable: do not cache
        pass
    return dg


def sir_counts(graph: DeviceGraph, seed_set, beta, gamma, sims, T, rng_seed, sim_offset=0, device="cuda",
               counts: torch.Tensor | None = None, edge_scan: bool = False) -> torch.Tensor:
    """Production Monte-Carlo on the GPU: uint32 (stored as int32 tensor) counts [3, T, n].

    `counts` may be passed to accumulate several shards of the sims range into one array.  edge_scan=True runs the
    edge-parallel statement of the same model (`gnode_sir_mc_philox_scan`: identical counts, O(nnz) per step)."""
    lib = _lib.load()
    seeds = np.ascontiguousarray(list(seed_set), dtype=np.int32)
    if counts is None:
        counts = torch.zeros((3, T, graph.n), dtype=torch.int32, device=device)
    ws_bytes = lib.gnode_sir_workspace_bytes(graph.handle, T)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=counts.device)
    fn = lib.gnode_sir_mc_philox_scan if edge_scan else lib.gnode_sir_mc_philox
    _lib.check(fn(graph.handle, _lib.host_ptr(seeds), int(seeds.shape[0]), float(beta), float(gamma),
                  int(sims), int(sim_offset), int(T), C.c_uint64(int(rng_seed) & (2**64 - 1)),
                  _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
    return counts


def sir_counts_counted(graph: DeviceGraph, seed_set, beta, gamma, sims, T, rng_seed, sim_offset=0, device="cuda"):
    """`sir_counts` through the kernel's profiling instantiation: (counts, stats) with stats = what the launch did --
    Philox blocks computed, infection coins drawn, recovery coins drawn, CSR entries read (bench.py's `sir` roofline)."""
    lib = _lib.load()
    seeds = np.ascontiguousarray(list(seed_set), dtype=np.int32)
    counts = torch.zeros((3, T, graph.n), dtype=torch.int32, device=device)
    ws = torch.empty(lib.gnode_sir_workspace_bytes(graph.handle, T), dtype=torch.uint8, device=counts.device)
    st = (C.c_uint64 * 4)()
    _lib.check(lib.gnode_sir_mc_philox_counted(graph.handle, _lib.host_ptr(seeds), int(seeds.shape[0]), float(beta), float(gamma),
                                               int(sims), int(sim_offset), int(T), C.c_uint64(int(rng_seed) & (2**64 - 1)),
                                               _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(), st))
    return counts, {"philox_blocks": int(st[0]), "infection_coins": int(st[1]), "recovery_coins": int(st[2]), "csr_entries_read": int(st[3])}


def sir_counts_coins(n, table: np.ndarray, seed_set, beta, gamma, sims, T, coins: np.ndarray, device="cuda"):
    """Parity mode: consume a recorded torch.rand stream exactly like the reference.
    Returns (counts int32 [3,T,n] on the GPU, coins consumed)."""
    lib = _lib.load()
    seeds = np.ascontiguousarray(list(seed_set), dtype=np.int32)
    tsrc = torch.from_numpy(np.ascontiguousarray(table[:, 0], dtype=np.int32)).to(device)
    tdst = torch.from_numpy(np.ascontiguousarray(table[:, 1], dtype=np.int32)).to(device)
    cz = torch.from_numpy(np.ascontiguousarray(coins, dtype=np.float64)).to(device)
    counts = torch.zeros((3, T, n), dtype=torch.int32, device=device)
    ws = torch.empty(lib.gnode_sir_coins_workspace_bytes(), dtype=torch.uint8, device=device)
    used = C.c_int64(0)
    _lib.check(lib.gnode_sir_mc_coins(_lib.ptr(tsrc), _lib.ptr(tdst), int(tsrc.numel()), int(n), _lib.host_ptr(seeds),
                                      int(seeds.shape[0]), float(beta), float(gamma), int(sims), int(T),
                                      _lib.ptr(cz) if cz.numel() else None, int(cz.numel()), _lib.ptr(counts),
                                      C.byref(used), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
    return counts, int(used.value)


def sir_torch(G, seed_set, beta, gamma, sims=10000, T=20, rng_seed=None, coins=None, normalize_t0=False):
    """Monte-Carlo SIR label generator, reference ode_nn.py:30-88.

    Returns (S, I, R) numpy float64 COUNTS of shape [1, T, n] exactly like the
    reference (callers divide by `sims`, ode_nn_ngraph_sim.py:199); row 0 of S and
    I holds the initial state once (the reference assigns it, :55-56).

    Coins: by default counter-based Philox keyed by (edge|node, step, sim), seeded
    from torch's CPU generator so `torch.manual_seed` governs reproducibility as it
    does for the reference's `torch.rand` (:65,:70).  `coins=` (a recorded stream)
    switches to the bit-exact parity mode.

    normalize_t0 (extension): the reference ASSIGNS row 0 in every trajectory
    (:55-56) instead of accumulating it, so after the caller's `/sims` that row reads 1/sims, not 1 (the loss
    skips t = 0, so it is invisible there).  False (default) reproduces the reference's counts exactly;
    True scales row 0 by `sims` so that counts/sims is the initial state itself.
    """
    n = G.number_of_nodes()
    if coins is not None:
        e = _edge_arrays(G)
        table = np.empty((2 * e.shape[0], 2), dtype=np.int64)     # reference :32-38
        table[0::2, 0], table[0::2, 1] = e[:, 0], e[:, 1]
        table[1::2, 0], table[1::2, 1] = e[:, 1], e[:, 0]
        counts, _ = sir_counts_coins(n, table, seed_set, beta, gamma, sims, T, np.asarray(coins))
    else:
        if rng_seed is None:
            rng_seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
        counts = sir_counts(_device_graph_for(G), seed_set, beta, gamma, sims, T, rng_seed)
    c = counts.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    c = c.astype(np.float64)
    if normalize_t0:
        c[:, 0] *= float(sims)
    return c[0][None], c[1][None], c[2][None]


def get_sir_t_nodes_torch(x_rk, maxTime, deltaT, count=True):
    """Row subsample out[i] = x[int(i/deltaT)], reference ode_nn.py:249-261.
    One device-side index_select instead of maxTime D2H row copies; the result stays
    on x_rk's device (the reference builds it on the CPU and callers move it back,
    ode_nn_ngraph_sim.py:234)."""
    idx = torch.as_tensor([int(i / deltaT) for i in range(int(maxTime))], device=x_rk.device)
    if count:
        return torch.sum(x_rk, axis=1).index_select(0, idx)
    return x_rk.index_select(0, idx)


def sir(x, y, A, beta, gamma):
    """Mean-field RHS, reference ode_nn.py:214-220 (numpy; kept for callers that pass it to `runge_kutta_order4`)."""
    n = np.shape(A)[0]
    S, I = x[:n], x[n:2 * n]
    AI = np.asarray(A @ I).reshape(-1)
    dS = -beta * AI * S
    return np.hstack([dS, -dS - gamma * I, gamma * I])


def runge_kutta_order4(sir, A, n_nodes, indices, beta_factor, gamma_factor, deltaT=1, maxTime=70, rtol=1e-10, atol=1e-12):
    """Mean-field baseline, reference ode_nn.py:222-233 (despite its name the reference runs scipy's LSODA):
    returns (I_sampled_t, S_sampled_t, R_sampled_t), float64 [maxTime, n] at the times int(i/deltaT)*deltaT.
    `sir` (the RHS callable) is accepted for signature compatibility; the integration runs in libgnode_hip.so
    (adaptive Dormand-Prince 5(4), sparse A I) -- SURVEY 8f rank 4, not on the `ode_nn` path."""
    lib = _lib.load()
    Ac = sp.csr_matrix(A)
    Ac.sort_indices()
    g = DeviceGraph(Ac.indptr.astype(np.int32), Ac.indices.astype(np.int32))
    n = Ac.shape[0]
    grid = np.arange(0, maxTime, deltaT)
    t_out = np.ascontiguousarray([grid[int(i / deltaT)] for i in range(int(maxTime))], dtype=np.float64)
    seeds = np.ascontiguousarray(list(indices), dtype=np.int32)
    dev = torch.device("cuda", torch.cuda.current_device())
    gam = torch.full((n,), float(gamma_factor), dtype=torch.float64, device=dev)
    out = torch.empty((3, len(t_out), n), dtype=torch.float64, device=dev)
    ws = torch.empty(lib.gnode_meanfield_workspace_bytes(g.handle), dtype=torch.uint8, device=dev)
    steps = C.c_int64(0)
    _lib.check(lib.gnode_meanfield_f64(g.handle, _lib.host_ptr(seeds), int(seeds.shape[0]), float(beta_factor), _lib.ptr(gam),
                                       _lib.host_ptr(t_out), int(len(t_out)), float(rtol), float(atol), _lib.ptr(out[0]),
                                       _lib.ptr(out[1]), _lib.ptr(out[2]), C.byref(steps), _lib.ptr(ws), ws.numel(),
                                       _lib.stream_ptr()))
    o = out.cpu().numpy()
    return o[0], o[1], o[2]


class CsrGraph:
    """What the path needs from the reference's networkx graph, restored from the CSR cache file: node count,
    the edge list in `G.edges()` iteration order (sir_torch's table order, ode_nn.py:32-38) and the adjacency."""

    def __init__(self, n, edges):
        self._n, self.edge_array = int(n), np.ascontiguousarray(edges, dtype=np.int64).reshape(-1, 2)

    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        return int(self.edge_array.shape[0])

    def edges(self):
        return [tuple(e) for e in self.edge_array.tolist()]

    def nodes(self):
        return range(self._n)


CACHE_SUFFIX = ".gnode-csr.npz"


def create_graph(n_nodes, graph_label="none", cache=False):
    """reference ode_nn.py:394-414: pickled networkx graph -> undirected -> largest
    connected component -> scipy adjacency.  Returns (G, A, 0).

    cache=True (extension): keep `<graph_label>.gnode-csr.npz` next to the pickle -- node count,
    edge list in `G.edges()` order, CSR of the adjacency -- and, when it is at least as new as the pickle, ingest
    from it without unpickling / re-deriving the component and the adjacency (G is then a `CsrGraph`)."""
    import networkx as nx
    if graph_label != "none":
        src, cfile = graph_label + ".pkl", graph_label + CACHE_SUFFIX
        if cache and os.path.exists(cfile) and os.path.getmtime(cfile) >= os.path.getmtime(src):
            z = np.load(cfile, allow_pickle=False)
            n = int(z["n"])
            A = sp.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=(n, n))
            return CsrGraph(n, z["edges"]), A, 0
        with open(src, "rb") as fh:
            G = pickle.load(fh)
        G = G.to_undirected()
        largest_cc = max(nx.connected_components(G), key=len)
        G = G.subgraph(largest_cc)
    else:
        G = nx.fast_gnp_random_graph(n_nodes, 0.2)
    A = nx.adjacency_matrix(G)
    if cache and graph_label != "none":
        # positions, not labels: the reference indexes tensors with node ids and all its graphs are labelled 0..n-1
        # in node order; anything else is not cacheable in this form
        nodes = list(G.nodes())
        if nodes == list(range(len(nodes))):
            Ac = sp.csr_matrix(A)
            np.savez(graph_label + CACHE_SUFFIX, n=np.int64(len(nodes)), edges=_edge_arrays(G), data=Ac.data,
                     indices=Ac.indices, indptr=Ac.indptr)
    return G, A, 0
