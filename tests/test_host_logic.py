"""CPU: host-side logic of the drop-in layer that needs no GPU."""
import os

import numpy as np
import pytest
import torch


def test_time_grid_and_subsample_rows():
    from gnode import ops
    g = ops.time_grid(20, 0.5)
    assert g.dtype == np.float64 and g.shape == (40,) and g[-1] == 19.5          # ode_nn_ngraph_sim.py:110
    dts = ops.step_sizes(g)
    assert dts.dtype == np.float32 and dts.shape == (39,) and np.all(dts == np.float32(0.5))
    assert ops.subsample_rows(20, 0.5).tolist() == list(range(0, 40, 2))         # ode_nn.py:257-259
    assert ops.subsample_rows(3, 0.3).tolist() == [int(i / 0.3) for i in range(3)]
    # non-dyadic step: fp32(dt_k) of the float64 grid differences, as torch would apply them
    d = ops.step_sizes(ops.time_grid(1, 0.1))
    assert d.shape == (9,) and np.all(np.abs(d - 0.1) < 1e-7)


def test_sample_tensor_and_splits():
    from gnode.trainer import sample_tensor, split_indices
    x = sample_tensor(10, 8, [2, 7], 0.3, 0.15)
    assert x.shape == (10, 11) and x.dtype == torch.float32
    assert x[:, 0].sum() == 8 and x[[2, 7], 1].tolist() == [1.0, 1.0] and x[:, 2].sum() == 0
    assert torch.all(x[:, 3] == np.float32(0.3)) and torch.all(x[:, 4] == np.float32(0.15)) and x[:, 5:].abs().sum() == 0
    xm = sample_tensor(5, 8, [0], 0.2, 0.1, marker=3)
    assert xm[0, 5] == 3 and xm[1:, 5].abs().sum() == 0                         # ode_nn_ngraphs.py:333
    tr, va, te = split_indices(200, [0.6, 0.2, 0.2])
    assert (len(tr), len(va), len(te)) == (120, 40, 40) and tr[0] == 0 and te[-1] == 199
    tr, va, te = split_indices(6, None, {"train": [0, 2], "val": [5], "test": [1, 3, 4]})
    assert (tr, va, te) == ([0, 2], [5], [1, 3, 4])


def test_label_paths_and_csv(tmp_path):
    from gnode.trainer import label_paths, csv_trials
    ps = label_paths("./real_graphs/karate", "./multi-graph-1/Experiments-seed2-karate", [25, 18])
    assert ps[0] == "./multi-graph-1/Experiments-seed2-karate/karate-S-25-18.pkl"   # ode_nn_ngraph_sim.py:191
    f = str(tmp_path / "Metrics")
    csv_trials(f, ["a", "b"], [1, 2]); csv_trials(f, ["a", "b"], [3, 4])
    assert open(f).read().splitlines() == ["a,b", "1,2", "3,4"]


def test_concat_csr_matches_block_diag():
    import scipy.sparse as sp
    import gnode_oracle as O
    from gnode.graph import concat_csr, csr_arrays
    gs = [O.er_graph(n, m, seed=n)[:2] for n, m in ((7, 9), (12, 20), (5, 4))]
    rp, ci = concat_csr([gs[1], gs[0], gs[1], gs[2]])
    mats = [sp.csr_matrix((np.ones(c.shape[0]), c, r), shape=(len(r) - 1,) * 2) for r, c in gs]
    bd = sp.block_diag([mats[1], mats[0], mats[1], mats[2]]).tocsr()
    bd.sort_indices()
    assert np.array_equal(rp, bd.indptr) and np.array_equal(ci, bd.indices)
    r2, c2 = csr_arrays(bd)
    assert np.array_equal(r2, rp) and np.array_equal(c2, ci)
    orp, oci, _ = O.concat_csr(gs, [1, 0, 1, 2])
    assert np.array_equal(orp, rp) and np.array_equal(oci, ci)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from gnode import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.GnodeError, match="no CPU fallback"):
        _lib.load()


def test_create_graph_csr_cache_roundtrip(tmp_path):
    """graph ingest through the CSR cache file gives the same adjacency, node count and
    `G.edges()` order (sir_torch's edge-table order) as the pickled networkx graph."""
    import pickle
    import networkx as nx
    from gnode import ode_nn
    G0 = nx.karate_club_graph()
    base = str(tmp_path / "karate")
    with open(base + ".pkl", "wb") as fh:
        pickle.dump(G0, fh)
    G1, A1, _ = ode_nn.create_graph(0, base, cache=True)
    assert os.path.exists(base + ode_nn.CACHE_SUFFIX) and not isinstance(G1, ode_nn.CsrGraph)
    G2, A2, _ = ode_nn.create_graph(0, base, cache=True)
    assert isinstance(G2, ode_nn.CsrGraph)
    assert G2.number_of_nodes() == G1.number_of_nodes() and G2.number_of_edges() == G1.number_of_edges()
    assert [tuple(e) for e in G1.edges()] == G2.edges()
    assert (abs(A1 - A2)).nnz == 0
    assert np.array_equal(ode_nn._edge_arrays(G1), ode_nn._edge_arrays(G2))
    G3, _, _ = ode_nn.create_graph(0, base)                      # default: the reference's own route
    assert not isinstance(G3, ode_nn.CsrGraph)


def test_synth_inputs_match_oracle_generators():
    """bench.py builds its inputs with gnode.synth (the oracle is only touched by its cpu_baseline leg); the two
    generators must stay identical so that the CPU baseline and the parity tests see the same workload."""
    import gnode_oracle as O
    from gnode import synth
    rp, ci = synth.er_csr(3000, 17000, seed=3)
    rp2, ci2, _ = O.er_graph(3000, 17000, seed=3)
    assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
    P, P2 = synth.linear_params(64, seed=5), O.init_params(64, seed=5)
    assert set(P) == set(P2) and all(np.array_equal(P[k], P2[k]) for k in P)
    assert np.array_equal(synth.samples(77, 3, 16, seed=2), O.make_samples(77, 3, 16, seed=2))
