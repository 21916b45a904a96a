"""CPU: the numpy oracle against the golden vectors the REFERENCE produced
(tests/golden/make_golden.py).  This is what pins the oracle before anything on
the GPU is compared with it."""
import glob
import os

import numpy as np
import pytest

import gnode_oracle as O

RTOL = 1e-5   # north_star: 1e-5 relative fp32


def _load(path):
    d = dict(np.load(path))
    P = {k[2:]: d[k] for k in d if k.startswith("P:")}
    return d, P


def _close(a, b, rtol=RTOL, atol=None):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = np.max(np.abs(b)) + 1e-30
    err = np.max(np.abs(a - b)) / scale
    assert err <= rtol, f"max rel-to-scale err {err:.3e} > {rtol}"


def _cases(prefix, golden_dir=os.path.join(os.path.dirname(__file__), "golden")):
    return sorted(glob.glob(os.path.join(golden_dir, prefix + "*.npz")))


@pytest.mark.parametrize("path", _cases("rhs_single_"), ids=os.path.basename)
def test_rhs_single_matches_reference(path):
    d, P = _load(path)
    rp, ci = O.csr_from_edges(int(d["n"]), d["edges"])
    dx = O.rhs_single(d["x"], P["odefunc.linear.weight"], P["odefunc.linear.bias"], rp, ci, int(d["n"]))
    assert dx.shape == d["dx"].shape and dx.dtype == np.float32
    _close(dx, d["dx"])
    q = dx.shape[0] // 4
    assert not dx[3 * q:].any()                      # 4th slab derivative is exactly 0


@pytest.mark.parametrize("path", _cases("fwd_single_"), ids=os.path.basename)
def test_forward_single_matches_reference(path):
    d, P = _load(path)
    rp, ci = O.csr_from_edges(int(d["n"]), d["edges"])
    S, I, R = O.odeblock_forward_single(d["x"], P, rp, ci, int(d["maxTime"]), float(d["deltaT"]))
    for got, want in ((S, d["S"]), (I, d["I"]), (R, d["R"])):
        assert got.shape == want.shape
        _close(got, want)
    np.testing.assert_allclose(S + I + R, 1.0, atol=1e-6)
    sub = O.get_sir_t_nodes(S[..., 0], int(d["maxTime"]), float(d["deltaT"]))
    _close(sub, d["S_sub"])
    if "loss" in d:
        loss = O.l1_loss(S, I, R, d["y"], int(d["maxTime"]), float(d["deltaT"]))
        assert abs(loss - float(d["loss"])) <= 1e-6


@pytest.mark.parametrize("path", _cases("multi_"), ids=os.path.basename)
def test_multi_graph_matches_reference(path):
    d, P = _load(path)
    graphs = [O.csr_from_edges(int(d[f"n{j}"]), d[f"edges{j}"]) for j in range(3)]
    dst = O.rhs_multi(d["state"], P["odefunc.linear.weight"], P["odefunc.linear.bias"], graphs)
    _close(dst, d["dstate"])
    S, I, R = O.odeblock_forward_multi(d["x"], P, graphs, int(d["maxTime"]), float(d["deltaT"]))
    for got, want in ((S, d["S"]), (I, d["I"]), (R, d["R"])):
        assert got.shape == want.shape
        _close(got, want)


@pytest.mark.parametrize("path", _cases("sir_"), ids=os.path.basename)
def test_sir_coin_stream_is_bit_exact(path):
    d, _ = _load(path)
    table = O.edge_table(d["edges"])
    S, I, R, used, trace = O.sir_coins(int(d["n"]), table, d["seeds"].tolist(), float(d["beta"]), float(d["gamma"]),
                                       int(d["sims"]), int(d["T"]), d["coins"])
    assert used == d["coins"].shape[0]               # consumed exactly the coins the reference drew
    for got, want in ((S, d["S"]), (I, d["I"]), (R, d["R"])):
        assert got.shape == want.shape
        assert np.array_equal(got, want)             # integer counts: bit-exact


def test_sir_philox_statistics_match_coin_model():
    """The production coin source changes only WHICH uniform numbers are drawn:
    means of many sims must agree with the reference-stream model within MC error."""
    d, _ = _load(_cases("sir_karate")[0])
    n = int(d["n"])
    rp, ci = O.csr_from_edges(n, d["edges"])
    sims, T = 3000, 12
    seeds, beta, gamma = d["seeds"].tolist(), float(d["beta"]), float(d["gamma"])
    cnt = O.sir_philox(n, rp, ci, seeds, beta, gamma, sims, T, rng_seed=1234)
    rng = np.random.default_rng(0)
    S, I, R, _, _ = O.sir_coins(n, O.edge_table(d["edges"]), seeds, beta, gamma, sims, T, rng.random(10_000_000))
    a = cnt[:, 1:, :].astype(np.float64) / sims
    b = np.stack([S[0], I[0], R[0]])[:, 1:, :] / sims
    assert np.max(np.abs(a - b)) < 0.06 and np.mean(np.abs(a - b)) < 0.012
    assert np.array_equal(cnt[0, 1:] + cnt[1, 1:] + cnt[2, 1:], np.full((T - 1, n), sims, dtype=np.uint32))


def test_sir_philox_sharding_is_exact():
    n = 34
    d, _ = _load(_cases("sir_karate")[0])
    rp, ci = O.csr_from_edges(n, d["edges"])
    whole = O.sir_philox(n, rp, ci, [0, 33], 0.3, 0.2, 40, 10, rng_seed=99)
    a = O.sir_philox(n, rp, ci, [0, 33], 0.3, 0.2, 25, 10, rng_seed=99, sim_offset=0)
    b = O.sir_philox(n, rp, ci, [0, 33], 0.3, 0.2, 15, 10, rng_seed=99, sim_offset=25)
    merged = a + b
    merged[:2, 0] = whole[:2, 0]                      # row 0 is assigned, not accumulated (Q3)
    assert np.array_equal(merged, whole)


def test_philox_known_answer():
    """Random123 known-answer vectors for philox4x32-10."""
    assert int(O.philox4x32_10(0, 0, 0, 0, 0, 0)) == 0x6627E8D5
    assert int(O.philox4x32_10(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)) == 0x408F276D
    # whole blocks (the Monte-Carlo coins use all four words: four consecutive CSR positions / node ids per block)
    blk = lambda *a: [int(w) for w in O.philox4x32_10_block(*a)]
    assert blk(0, 0, 0, 0, 0, 0) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert blk(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert blk(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    # the coin of item pos = word (pos & 3) of block pos >> 2
    w = O.philox4x32_10_block(5, 3, 7, 1, 11, 13)
    assert [int(v) for v in O.philox_coin(np.asarray([20, 21, 22, 23]), 3, 7, 1, 11, 13)] == [int(x) for x in w]
    assert int(O.philox4x32_10(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0)) == 0xD16CFE09


def test_er_graph_contract():
    rp, ci, e = O.er_graph(500, 2000, seed=0)
    assert e.shape == (2000, 2) and ci.shape[0] == 4000 and rp[-1] == 4000
    assert np.all(e[:, 0] != e[:, 1])
    for r in range(500):
        seg = ci[rp[r]:rp[r + 1]]
        assert np.all(np.diff(seg) > 0)


@pytest.mark.parametrize("name", ["karate", "er120", "loops40"])
def test_dmp_oracle_matches_reference_vectors(name):
    """DMP baseline: the numpy restatement against vectors the reference class produced
    (tests/golden/make_golden_dmp.py; torch_scatter's scatter-mul itself is parity unpinned)."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", f"dmp_{name}.npz"))
    got = O.dmp_sir(d["rowptr"], d["col"], d["weights"], d["gamma"], d["seeds"].tolist(), int(d["maxTime"]))
    assert got.shape == d["out"].shape
    assert np.array_equal(got, d["out"])                     # same float32 operation order: bit-exact on the CPU


@pytest.mark.parametrize("name", ["karate", "er150"])
def test_meanfield_oracle_matches_reference_vectors(name):
    """Mean-field baseline: scipy LSODA on the sparse matrix vs the reference's
    `runge_kutta_order4` on the dense one (tests/golden/make_golden_meanfield.py).  Same integrator, same
    tolerances; only the summation order inside A I differs."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", f"meanfield_{name}.npz"))
    I, S, R = O.meanfield_rk(d["rowptr"], d["col"], d["seeds"].tolist(), float(d["beta"]), float(d["gamma"]),
                             float(d["deltaT"]), int(d["maxTime"]))
    for got, want in ((I, d["I"]), (S, d["S"]), (R, d["R"])):
        assert got.shape == want.shape
        assert np.max(np.abs(got - want)) <= 1e-9


# ---- full horizon at the BASELINE graph sizes (tests/golden/make_golden_fullsize.py): the reference's own fp32
# output over 59 Euler steps on fb-social- and wiki-vote-sized graphs pins both CPU restatements there.
def _synth():
    """gnode/synth.py on its own (numpy + scipy only; importing the gnode package would pull in torch and the library)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "_synth", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gn-ode-sir_amd", "gnode", "synth.py"))
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    return synth


def _full_inputs(d):
    """rebuild graph / weights / sample from the stored seeds"""
    synth = _synth()
    n, H = int(d["n"]), int(d["H"])
    rp, ci = synth.er_csr(n, int(d["m"]), seed=int(d["graph_seed"]))
    return rp, ci, synth.linear_params(H, seed=int(d["param_seed"])), synth.samples(n, 1, H, seed=int(d["sample_seed"]))


@pytest.mark.parametrize("path", _cases("full_"), ids=os.path.basename)
def test_full_horizon_matches_reference(path):
    """Both CPU restatements (numpy, C) against the reference's fp32 output over the FULL 59-step horizon, with
    the reference's own float64 run as the yardstick.  The bar is north_star's 1e-5 wherever the reference's own
    fp32 is itself within 1e-5 of its float64 run (grid points <= 20 always are); over the whole horizon the
    restatement may be off by at most what the reference's fp32 is off from float64 (+1e-5)."""
    import oracle_c as OC
    d = dict(np.load(path))
    rp, ci, P, x = _full_inputs(d)
    n, maxTime, deltaT = int(d["n"]), int(d["maxTime"]), float(d["deltaT"])
    rows, rows64 = d["rows"], d["rows64"]
    floor = float(d["ref_f32_vs_f64_maxabs"])
    S, I, R = O.odeblock_forward_single(x, P, rp, ci, maxTime, deltaT)
    Sc, Ic, Rc = OC.forward_euler(rp, ci, n, x, P, O.step_sizes(O.time_grid(maxTime, deltaT)))
    early = rows <= 20
    for name, got in (("numpy", (S, I, R)), ("C", (Sc, Ic, Rc))):
        for c, g in zip("SIR", got):
            g = g[..., 0]
            e_early = np.max(np.abs(g[rows[early]].astype(np.float64) - d[c][early]))
            e_all = np.max(np.abs(g[rows].astype(np.float64) - d[c]))
            e64 = np.max(np.abs(g[rows64].astype(np.float64) - d[c + "64"]))
            assert e_early <= RTOL, f"{name} {c}: {e_early:.2e} on grid points <= 20"
            assert e_all <= RTOL + floor, f"{name} {c}: {e_all:.2e} vs reference fp32 over 59 steps (reference fp32 vs f64: {floor:.2e})"
            assert e64 <= RTOL + floor, f"{name} {c}: {e64:.2e} vs reference float64"
    # the reference's loss expression (ode_nn_ngraph_sim.py:234) on its own outputs
    from golden.labels import closed_form_labels
    y = closed_form_labels(1, n, maxTime)
    loss = O.l1_loss(S, I, R, y, maxTime, deltaT)
    assert abs(loss - float(d["loss"])) <= 1e-6


@pytest.mark.parametrize("path", _cases("adjoint_"), ids=os.path.basename)
def test_adjoint_restatement_matches_reference_classes(path):
    """The oracle's adjoint-Euler gradient (autograd over the oracle's OWN restatement of the RHS, head and encoder)
    against the gradients the REFERENCE classes produced under the same integrator rule
    (tests/golden/make_golden_adjoint.py: reference ODEBlock / ODEfunc / loss expression, float64): pins every
    derivative on the training path except torchdiffeq's adjoint rule itself, which both sides restate."""
    from golden.labels import closed_form_labels
    synth = _synth()
    d = dict(np.load(path))
    n, B, H, maxTime, deltaT = int(d["n"]), int(d["B"]), int(d["H"]), int(d["maxTime"]), float(d["deltaT"])
    full = "graph_seed" in d                                                      # 59-interval fixtures: graph by seed, 3 output rows stored
    rp, ci = synth.er_csr(n, int(d["m"]), seed=int(d["graph_seed"])) if full else O.csr_from_edges(n, d["edges"])
    P = synth.linear_params(H, seed=int(d["param_seed"]))
    x = synth.samples(n, B, H, seed=int(d["sample_seed"]))
    y = closed_form_labels(B, n, maxTime).reshape(B * n, maxTime, 3)
    rows = np.asarray([int(i / deltaT) for i in range(maxTime)])
    if full:
        # the loss's cotangent needs all maxTime output rows: take them from the oracle's own float64 forward, held to the
        # reference's float64 outputs at the stored rows (and to the reference's loss below)
        with O.precision(np.float64):
            So, Io, Ro = O.odeblock_forward_single(x.astype(np.float64), {k: v.astype(np.float64) for k, v in P.items()}, rp, ci, maxTime, deltaT)
        pred = np.stack([So[rows, :, 0], Io[rows, :, 0], Ro[rows, :, 0]], -1)
        for j, c in enumerate("SIR"):
            assert np.abs(pred[d["rows_kept"], :, j] - d[c]).max() <= 1e-12
    else:
        pred = np.stack([d[c] for c in "SIR"], -1)                                # [T, rows, 3]: the reference's outputs
    diff = pred - y.transpose(1, 0, 2)
    diff[0] = 0.0                                                                 # t = 0 excluded (:234)
    N = B * n * (maxTime - 1) * 3
    loss = np.abs(diff).sum() / N
    assert abs(loss - float(d["loss"])) <= 1e-9
    g = np.sign(diff) / N
    got = O.adjoint_grads_torch(x, P, rp, ci, maxTime, deltaT, g[..., 0], g[..., 1], g[..., 2], out_rows=rows, dtype="float64")
    for k, v in got.items():
        want = d["G:" + k]
        assert np.max(np.abs(np.asarray(v) - want)) <= 1e-9 * (np.max(np.abs(want)) + 1e-12) + 1e-15, k
