"""GPU parity: the HIP path (through the C ABI) against the golden vectors the
reference produced and against the CPU oracle on the same seeded inputs.

Tolerance: 1e-5 relative (to the tensor's max magnitude) for fp32, as north_star
states; bit-exact for the integer Monte-Carlo counts.
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-5
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gnode import _lib
    _lib.load()                       # fails loudly if libgnode_hip.so is missing
    return torch.device("cuda:0")


def _load(path):
    d = dict(np.load(path))
    return d, {k[2:]: d[k] for k in d if k.startswith("P:")}


def _cases(prefix):
    return sorted(glob.glob(os.path.join(GOLD, prefix + "*.npz")))


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30)


def _tp(P, dev):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in P.items()}


# ------------------------------------------------------------------ A1: ODEfunc.forward
@pytest.mark.parametrize("path", _cases("rhs_single_"), ids=os.path.basename)
def test_rhs_single_golden(path, dev):
    import torch
    import gnode_oracle as O
    from gnode import ops
    from gnode.graph import DeviceGraph
    d, P = _load(path)
    rp, ci = O.csr_from_edges(int(d["n"]), d["edges"])
    g = DeviceGraph(rp, ci)
    p = _tp(P, dev)
    dx = ops.rhs(g, torch.from_numpy(d["x"]).to(dev), p["odefunc.linear.weight"], p["odefunc.linear.bias"]).cpu().numpy()
    assert _rel(dx, d["dx"]) <= RTOL
    q = dx.shape[0] // 4
    assert not dx[3 * q:].any()


def test_rhs_module_surface(dev):
    """ODEfunc(A, beta, gamma, hidden1, device).forward(t, x) keeps the reference signature."""
    import torch
    import scipy.sparse as sp
    import gnode_oracle as O
    from gnode.ode_nn_ngraph_sim import ODEfunc
    d, P = _load(_cases("rhs_single_karate_B2_H8")[0])
    n = int(d["n"])
    rp, ci = O.csr_from_edges(n, d["edges"])
    A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
    f = ODEfunc(A, 0.2, 0.1, 8, dev).to(dev)
    assert set(f.state_dict().keys()) == {"ln.weight", "ln.bias", "linear.weight", "linear.bias"}
    f.linear.weight.data.copy_(torch.from_numpy(P["odefunc.linear.weight"]))
    f.linear.bias.data.copy_(torch.from_numpy(P["odefunc.linear.bias"]))
    dx = f(torch.tensor(0.0), torch.from_numpy(d["x"]).to(dev))
    assert _rel(dx.cpu().numpy(), d["dx"]) <= RTOL


# ------------------------------------------------------------------ A3/A4/A5: ODEBlock.forward
@pytest.mark.parametrize("path", _cases("fwd_single_"), ids=os.path.basename)
def test_forward_single_golden(path, dev):
    import torch
    import scipy.sparse as sp
    import gnode_oracle as O
    from gnode.ode_nn_ngraph_sim import ODEfunc, ODEBlock
    from gnode.ode_nn import get_sir_t_nodes_torch
    from gnode import ops
    d, P = _load(path)
    n, H = int(d["n"]), d["x"].shape[2] - 3
    maxTime, deltaT = int(d["maxTime"]), float(d["deltaT"])
    rp, ci = O.csr_from_edges(n, d["edges"])
    A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
    f = ODEfunc(A, 0.2, 0.1, H, dev)
    m = ODEBlock(maxTime, deltaT, n, [0], H, f, dev).to(dev)
    want_keys = {"odefunc.ln.weight", "odefunc.ln.bias", "odefunc.linear.weight", "odefunc.linear.bias",
                 "linearS1.weight", "linearS1.bias", "ln.weight", "ln.bias", "linear3.weight", "linear3.bias",
                 "linearS2.weight", "linearS2.bias"}
    assert set(m.state_dict().keys()) == want_keys
    m.load_state_dict({**m.state_dict(), **{k: torch.from_numpy(v) for k, v in P.items()}})
    x = torch.from_numpy(d["x"]).to(dev)
    with torch.no_grad():
        S, I, R = m(x)
    for got, want in ((S, d["S"]), (I, d["I"]), (R, d["R"])):
        assert tuple(got.shape) == want.shape
        assert _rel(got.cpu().numpy(), want) <= RTOL
    sub = get_sir_t_nodes_torch(torch.squeeze(S, -1), maxTime, deltaT, count=False)
    assert _rel(sub.cpu().numpy(), d["S_sub"]) <= RTOL
    # fused subsample == subsample of the full output, exactly
    with torch.no_grad():
        S2, _, _ = m(x, out_rows=ops.subsample_rows(maxTime, deltaT))
    assert torch.equal(S2.squeeze(-1), sub)
    if "loss" in d:
        loss = O.l1_loss(S.cpu().numpy(), I.cpu().numpy(), R.cpu().numpy(), d["y"], maxTime, deltaT)
        assert abs(loss - float(d["loss"])) <= 1e-6


@pytest.mark.parametrize("path", _cases("multi_"), ids=os.path.basename)
def test_multi_graph_golden(path, dev):
    import torch
    import scipy.sparse as sp
    import gnode_oracle as O
    from gnode.ode_nn_ngraphs import ODEfunc, ODEBlock
    d, P = _load(path)
    H = d["x"].shape[1] - 3
    A_list = []
    for j in range(3):
        n = int(d[f"n{j}"])
        rp, ci = O.csr_from_edges(n, d[f"edges{j}"])
        A_list.append(sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n)))
    f = ODEfunc(A_list, H, dev)
    m = ODEBlock(int(d["maxTime"]), float(d["deltaT"]), H, f, dev).to(dev)
    m.load_state_dict({**m.state_dict(), **{k: torch.from_numpy(v) for k, v in P.items()}})
    dst = f(torch.tensor(0.0), torch.from_numpy(d["state"]).to(dev))
    assert _rel(dst.cpu().numpy(), d["dstate"]) <= RTOL
    with torch.no_grad():
        S, I, R = m(torch.from_numpy(d["x"]).to(dev))
    for got, want in ((S, d["S"]), (I, d["I"]), (R, d["R"])):
        assert tuple(got.shape) == want.shape
        assert _rel(got.cpu().numpy(), want) <= RTOL


def _truth64(x, P, rp, ci, maxTime, deltaT, method):
    """The same algorithm in float64: the yardstick for fp32 noise on long horizons."""
    import gnode_oracle as O
    with O.precision(np.float64):
        P64 = {k: v.astype(np.float64) for k, v in P.items()}
        return O.odeblock_forward_single(x.astype(np.float64), P64, rp, ci, maxTime, deltaT, method=method)


@pytest.mark.parametrize("n,m,B,H,maxTime,deltaT,method", [
    (1893, 13835, 4, 64, 30, 0.5, "euler"),     # fb-social sized (config 2), full 59-step horizon
    (7066, 100736, 1, 64, 30, 0.5, "euler"),    # wiki-vote sized (config 3)
    (500, 3000, 3, 32, 8, 0.5, "euler"),        # generic (non-MFMA) node-MLP path
    (500, 3000, 2, 24, 6, 0.25, "euler"),       # H not a power of two
    (333, 1500, 5, 128, 4, 0.5, "euler"),
    (400, 2000, 2, 64, 6, 0.5, "rk4"),
    (400, 2000, 2, 16, 6, 0.5, "rk4"),
])
def test_forward_vs_oracle(n, m, B, H, maxTime, deltaT, method, dev):
    """Tolerance: 1e-5 relative-to-scale against the fp32 CPU oracle on horizons where fp32
    itself is reproducible to that level (<= 20 Euler steps here).  The SIR dynamics amplify
    rounding differences (x2 every ~6 steps on these graphs): at 59 steps two CPU fp32
    restatements (numpy vs C) already differ by 1.8e-5 and sit 2e-5 from the float64
    result, so there the bar is the fp32 NOISE FLOOR: the GPU must be no further from the
    float64 result than 2x the CPU fp32 oracle is (and never worse than 1e-4)."""
    import torch
    import gnode_oracle as O
    import oracle_c as OC
    from gnode import ops
    from gnode.graph import DeviceGraph
    rp, ci, _ = O.er_graph(n, m, seed=n)
    P = O.init_params(H, seed=H)
    x = O.make_samples(n, B, H, seed=B)
    grid = O.time_grid(maxTime, deltaT)
    if method == "euler":
        want = OC.forward_euler(rp, ci, n, x, P, O.step_sizes(grid))
    else:
        want = O.odeblock_forward_single(x, P, rp, ci, maxTime, deltaT, method="rk4")
    g = DeviceGraph(rp, ci)
    S, I, R, _ = ops.forward(g, torch.from_numpy(x).to(dev).reshape(B * n, 3 + H), _tp(P, dev), ops.step_sizes(grid), method)
    got = [t.cpu().numpy() for t in (S, I, R)]
    if grid.shape[0] - 1 <= 20:
        for a, w in zip(got, want):
            assert _rel(a, w[..., 0]) <= RTOL
    else:
        truth = _truth64(x, P, rp, ci, maxTime, deltaT, method)
        for a, w, t in zip(got, want, truth):
            floor = _rel(w[..., 0], t[..., 0])
            err = _rel(a, t[..., 0])
            print(f"fp32 oracle vs f64: {floor:.2e}   gpu vs f64: {err:.2e}   gpu vs fp32 oracle: {_rel(a, w[..., 0]):.2e}")
            assert err <= max(RTOL, 2.0 * floor) and err <= 1e-4
            # and the first 20 steps, where fp32 is reproducible, meet the plain 1e-5 bar
            assert _rel(a[:21], w[:21, :, 0]) <= RTOL
    s = got[0] + got[1] + got[2]
    assert np.max(np.abs(s - 1.0)) < 1e-5


def test_forward_sol_matches_states(dev):
    """sol (what odeint returns) is consistent with the fused outputs and with the oracle.  Slabs S, I, R: every grid
    point.  4th slab: grid point 0 carries beta / gamma as odeint's y0 does; its derivative is 0, so odeint repeats it
    at every grid point -- the H = 64 training forward uses those copies (1 <= k <= n_steps - 1) to keep A Z_I(y_k) for
    the adjoint backward instead (include/gnode.h), which is checked here against the oracle's own A Z_I."""
    import torch
    import gnode_oracle as O
    from gnode import ops
    from gnode.graph import DeviceGraph
    n, B, H = 150, 2, 64
    rp, ci, _ = O.er_graph(n, 600, seed=9)
    P = O.init_params(H, seed=3)
    x = O.make_samples(n, B, H, seed=1)
    g = DeviceGraph(rp, ci)
    # (without a keep buffer: with one, the neighbour sums go there instead, test_kept_activations_... in test_gpu_backward.py)
    S, I, R, sol = ops.forward(g, torch.from_numpy(x).to(dev).reshape(B * n, 3 + H), _tp(P, dev),
                               ops.step_sizes(ops.time_grid(5, 0.5)), want_sol=True, want_keep=False)
    So, Io, Ro, sol_o = O.odeblock_forward_single(x, P, rp, ci, 5, 0.5, return_sol=True)
    assert tuple(sol.shape) == sol_o.shape
    q = 3 * B * n
    sol_h = sol.cpu().numpy()
    assert _rel(sol_h[:, :q], sol_o[:, :q]) <= RTOL
    assert np.array_equal(sol_h[0, q:], sol_o[0, q:])                       # beta-gamma slab of y0
    G = sol_o.shape[0]
    for k in range(1, G - 1):                                               # kept neighbour sums A Z_I(y_k)
        yI = sol_o[k, B * n:2 * B * n].astype(np.float64)
        zI = 1.0 / (1.0 + np.exp(-(yI @ P["odefunc.linear.weight"].T.astype(np.float64) + P["odefunc.linear.bias"])))
        AI = O._spmm_blockdiag(rp, ci, n, zI.astype(np.float32))
        assert _rel(sol_h[k, q:], AI) <= RTOL
    # generic hidden size: the 4th slab is odeint's (the beta-gamma slab rides along at every grid point)
    H2 = 16
    P2 = O.init_params(H2, seed=3)
    x2 = O.make_samples(n, B, H2, seed=1)
    _, _, _, sol2 = ops.forward(g, torch.from_numpy(x2).to(dev).reshape(B * n, 3 + H2), _tp(P2, dev),
                                ops.step_sizes(ops.time_grid(5, 0.5)), want_sol=True)
    sol2_o = O.odeblock_forward_single(x2, P2, rp, ci, 5, 0.5, return_sol=True)[3]
    assert np.array_equal(sol2[:, q:].cpu().numpy(), sol2_o[:, q:])


def test_block_diagonal_independence(dev):
    """Samples of a batch never mix: a batched forward equals per-sample forwards bit for bit."""
    import torch
    import gnode_oracle as O
    from gnode import ops
    from gnode.graph import DeviceGraph
    n, B, H = 700, 4, 64
    rp, ci, _ = O.er_graph(n, 4000, seed=21)
    P = _tp(O.init_params(H, seed=5), dev)
    x = torch.from_numpy(O.make_samples(n, B, H, seed=2)).to(dev)
    g = DeviceGraph(rp, ci)
    dts = ops.step_sizes(ops.time_grid(6, 0.5))
    S, I, R, _ = ops.forward(g, x.reshape(B * n, 3 + H), P, dts)
    for b in range(B):
        Sb, Ib, Rb, _ = ops.forward(g, x[b].contiguous(), P, dts)
        assert torch.equal(S[:, b * n:(b + 1) * n], Sb) and torch.equal(I[:, b * n:(b + 1) * n], Ib)


def test_edge_cases(dev):
    """Isolated nodes (empty CSR rows), a hub row longer than one index chunk, zero steps."""
    import torch
    import gnode_oracle as O
    from gnode import ops
    from gnode.graph import DeviceGraph
    n, H = 130, 64
    edges = [(0, j) for j in range(1, 100)] + [(100, 101)]       # hub of degree 99; nodes 102..129 isolated
    rp, ci = O.csr_from_edges(n, edges)
    P = O.init_params(H, seed=8)
    x = O.make_samples(n, 2, H, seed=4)
    g = DeviceGraph(rp, ci)
    S, I, R, _ = ops.forward(g, torch.from_numpy(x).to(dev).reshape(2 * n, 3 + H), _tp(P, dev),
                             ops.step_sizes(ops.time_grid(4, 0.5)))
    So, Io, Ro = O.odeblock_forward_single(x, P, rp, ci, 4, 0.5)
    assert max(_rel(S.cpu().numpy(), So[..., 0]), _rel(I.cpu().numpy(), Io[..., 0])) <= RTOL
    S0, _, _, _ = ops.forward(g, torch.from_numpy(x).to(dev).reshape(2 * n, 3 + H), _tp(P, dev), np.zeros(0, np.float32))
    assert tuple(S0.shape) == (1, 2 * n)
    assert _rel(S0.cpu().numpy(), So[:1, :, 0]) <= RTOL


def test_bad_arguments_raise(dev):
    import torch
    import gnode_oracle as O
    from gnode import ops, GnodeError
    from gnode.graph import DeviceGraph
    rp, ci, _ = O.er_graph(50, 100, seed=1)
    g = DeviceGraph(rp, ci)
    P = _tp(O.init_params(64), dev)
    with pytest.raises(GnodeError):   # rows not a multiple of n
        ops.forward(g, torch.zeros(75, 67, device=dev), P, np.zeros(2, np.float32))
    with pytest.raises(GnodeError):   # H % 4 != 0
        ops.forward(g, torch.zeros(50, 3 + 6, device=dev), _tp(O.init_params(6), dev), np.zeros(2, np.float32))
    with pytest.raises(GnodeError):   # CPU tensor
        ops.forward(g, torch.zeros(50, 67), P, np.zeros(2, np.float32))


# ------------------------------------------------------------------ A8: sir_torch
@pytest.mark.parametrize("path", _cases("sir_"), ids=os.path.basename)
def test_sir_recorded_coins_bit_exact(path, dev):
    """Integer infection state under the reference's own coin stream: bit-exact."""
    import gnode_oracle as O
    from gnode.ode_nn import sir_counts_coins
    d, _ = _load(path)
    table = O.edge_table(d["edges"])
    counts, used = sir_counts_coins(int(d["n"]), table, d["seeds"].tolist(), float(d["beta"]), float(d["gamma"]),
                                    int(d["sims"]), int(d["T"]), d["coins"])
    assert used == d["coins"].shape[0]
    c = counts.cpu().numpy().astype(np.float64)
    assert np.array_equal(c[0][None], d["S"]) and np.array_equal(c[1][None], d["I"]) and np.array_equal(c[2][None], d["R"])


def test_sir_torch_surface_with_coins(dev):
    import networkx as nx
    from gnode.ode_nn import sir_torch
    d, _ = _load(_cases("sir_karate")[0])
    G = nx.Graph()
    G.add_nodes_from(range(int(d["n"])))
    G.add_edges_from(d["edges"].tolist())
    assert np.array_equal(np.asarray(list(G.edges())), d["edges"])
    S, I, R = sir_torch(G, d["seeds"].tolist(), float(d["beta"]), float(d["gamma"]), int(d["sims"]), int(d["T"]),
                        coins=d["coins"])
    assert S.shape == d["S"].shape and S.dtype == np.float64
    assert np.array_equal(S, d["S"]) and np.array_equal(I, d["I"]) and np.array_equal(R, d["R"])


@pytest.mark.parametrize("n,m,sims,T", [(34, 78, 500, 20), (1000, 6000, 300, 15), (7066, 100736, 64, 20)])
def test_sir_philox_bit_exact_vs_oracle(n, m, sims, T, dev):
    import gnode_oracle as O
    import oracle_c as OC
    from gnode.graph import DeviceGraph
    from gnode.ode_nn import sir_counts
    rp, ci, _ = O.er_graph(n, m, seed=m)
    seeds = [1, n // 2]
    g = DeviceGraph(rp, ci)
    got = sir_counts(g, seeds, 0.3, 0.2, sims, T, rng_seed=0xABCDEF0123, sim_offset=11).cpu().numpy().astype(np.uint32)
    want = OC.sir_philox(n, rp, ci, seeds, 0.3, 0.2, sims, T, rng_seed=0xABCDEF0123, sim_offset=11)
    assert np.array_equal(got, want)
    assert np.array_equal(got[0, 1:] + got[1, 1:] + got[2, 1:], np.full((T - 1, n), sims, np.uint32))


def test_sir_philox_sharded_equals_whole(dev):
    """Sharding the sims range (what multi-GPU does) reproduces the single-call counts exactly."""
    import gnode_oracle as O
    from gnode.graph import DeviceGraph
    from gnode.ode_nn import sir_counts
    rp, ci, _ = O.er_graph(400, 2400, seed=3)
    g = DeviceGraph(rp, ci)
    whole = sir_counts(g, [7], 0.4, 0.1, 1000, 12, rng_seed=5)
    acc = sir_counts(g, [7], 0.4, 0.1, 600, 12, rng_seed=5, sim_offset=0)
    acc = sir_counts(g, [7], 0.4, 0.1, 400, 12, rng_seed=5, sim_offset=600, counts=acc)
    import torch
    assert torch.equal(whole, acc)


def test_sir_torch_statistics(dev):
    """Default (Philox) sir_torch agrees with the reference-stream model within Monte-Carlo error, cell by cell,
    and keeps the reference's output contract ([1,T,n] float64 counts, row-0 quirk).  Bar: for every (t, node,
    compartment) cell the two independent 10 000-trajectory estimates of the same probability p differ by at
    most 5 standard deviations of that difference, 5*sqrt(2 p (1-p) / sims) (p pooled), plus 2/sims for the
    cells whose p is within a few counts of 0 or 1.  Both streams are seeded, so the outcome is fixed."""
    import networkx as nx
    import torch
    import gnode_oracle as O
    from gnode.ode_nn import sir_torch
    G = nx.karate_club_graph()
    sims, T = 10000, 12
    torch.manual_seed(0)
    S, I, R = sir_torch(G, [0, 33], 0.3, 0.2, sims, T)
    assert S.shape == (1, T, 34) and S.dtype == np.float64
    assert S[0, 0].sum() == 32 and I[0, 0].sum() == 2 and R[0, 0].sum() == 0
    # extension (quirk Q3): normalize_t0 makes counts/sims the initial state at t = 0, everything else unchanged
    torch.manual_seed(0)
    S1, I1, R1 = sir_torch(G, [0, 33], 0.3, 0.2, sims, T, normalize_t0=True)
    assert np.array_equal(S1[0, 1:], S[0, 1:]) and np.array_equal(I1[0, 1:], I[0, 1:]) and np.array_equal(R1[0, 1:], R[0, 1:])
    assert np.array_equal(S1[0, 0], S[0, 0] * sims) and np.array_equal(I1[0, 0], I[0, 0] * sims) and not R1[0, 0].any()
    e = np.asarray(list(G.edges()))
    rng = np.random.default_rng(1)
    So, Io, Ro, _, _ = O.sir_coins(34, O.edge_table(e), [0, 33], 0.3, 0.2, sims, T, rng.random(4_000_000))
    worst = 0.0
    for a, b in ((S, So), (I, Io), (R, Ro)):
        pa, pb = a[0, 1:] / sims, b[0, 1:] / sims
        p = 0.5 * (pa + pb)
        bound = 5.0 * np.sqrt(2.0 * p * (1.0 - p) / sims) + 2.0 / sims
        worst = max(worst, float(np.max(np.abs(pa - pb) / bound)))
        assert np.all(np.abs(pa - pb) <= bound), f"worst cell at {np.max(np.abs(pa - pb) / bound):.2f} of its 5-sigma bound"
    print(f"sir_torch statistics: worst cell at {worst:.2f} of its 5-sigma bound")


# ------------------------------------------------------------------ BASELINE sizes (configs[3] shape)
def test_full_size_short_horizon_vs_oracle(dev):
    """75 000 nodes / 1 000 000 directed edges / H = 64, 2 samples: 4 Euler steps against the C oracle
    (1e-5), then size-independent properties of the full 59-step run: S+I+R = 1, finite, and samples of a
    batch never mix (bit-identical to a solo run)."""
    import torch
    import gnode_oracle as O
    import oracle_c as OC
    from gnode import ops
    from gnode.graph import DeviceGraph
    n, B, H = 75000, 2, 64
    rp, ci, _ = O.er_graph(n, 500000, seed=0)
    assert ci.shape[0] == 1000000
    P = O.init_params(H, seed=0)
    x = O.make_samples(n, B, H, seed=11)
    g = DeviceGraph(rp, ci)
    xt = torch.from_numpy(x).to(dev)
    dts4 = ops.step_sizes(ops.time_grid(2.5, 0.5))
    S, I, R, _ = ops.forward(g, xt.reshape(B * n, 3 + H), _tp(P, dev), dts4)
    want = OC.forward_euler(rp, ci, n, x, P, dts4)
    for got, w in zip((S, I, R), want):
        assert _rel(got.cpu().numpy(), w[..., 0]) <= RTOL
    dts = ops.step_sizes(ops.time_grid(30, 0.5))
    S, I, R, _ = ops.forward(g, xt.reshape(B * n, 3 + H), _tp(P, dev), dts)
    tot = S + I + R
    assert bool(torch.isfinite(tot).all()) and float((tot - 1).abs().max()) < 1e-5
    S1, I1, _, _ = ops.forward(g, xt[1].contiguous(), _tp(P, dev), dts)
    assert torch.equal(S[:, n:], S1) and torch.equal(I[:, n:], I1)
    # the trajectory-saving path (Y_R carried in full) agrees with the inference path (projected R)
    S2, I2, R2, sol = ops.forward(g, xt[:1].reshape(n, 3 + H), _tp(P, dev), dts[:6], want_sol=True)
    S3, I3, R3, _ = ops.forward(g, xt[:1].reshape(n, 3 + H), _tp(P, dev), dts[:6])
    for u, v in ((S2, S3), (I2, I3), (R2, R3)):
        assert _rel(u.cpu().numpy(), v.cpu().numpy()) <= 2e-6


# ------------------------------------------------------------------ degree skew (hub rows)
@pytest.mark.parametrize("n,m,B,H,method", [(1500, 20000, 3, 64, "euler"), (1500, 20000, 2, 8, "euler"),
                                             (900, 9000, 2, 32, "rk4"), (7066, 100736, 2, 64, "euler"),
                                             (7066, 100736, 3, 64, "euler")])
def test_skewed_degree_graph_vs_oracle(n, m, B, H, method, dev):
    """Power-law-like graphs (hubs of degree ~ n/3, far above the hub threshold) go through the segmented hub
    path; results must still match the oracle, and a batched run must equal per-sample runs bit for bit.  The last
    case is batched past the persistent grid (1 326 tiles: the pipelined instantiation, several tiles per workgroup)
    while its single sample runs in latency mode (one tile per workgroup, 16 rows in flight): the two instantiations
    of the step kernel must agree bit for bit, hub rows and 33..96-edge rows included."""
    import torch
    import gnode_oracle as O
    import oracle_c as OC
    from gnode import ops
    from gnode.graph import DeviceGraph
    rp, ci, _ = O.chung_lu_graph(n, m, seed=n)
    deg = np.diff(rp)
    assert deg.max() > 200 and (deg == 0).sum() >= 0
    P = O.init_params(H, seed=2)
    x = O.make_samples(n, B, H, seed=3)
    x[:, :, 3] *= 0.02        # beta: a degree-500 hub makes AI ~ 250, keep the dynamics out of the stiff regime
    grid = O.time_grid(5, 0.5)
    g = DeviceGraph(rp, ci)
    xt = torch.from_numpy(x).to(dev)
    S, I, R, _ = ops.forward(g, xt.reshape(B * n, 3 + H), _tp(P, dev), ops.step_sizes(grid), method)
    if method == "euler":
        want = OC.forward_euler(rp, ci, n, x, P, O.step_sizes(grid))
    else:
        want = O.odeblock_forward_single(x, P, rp, ci, 5, 0.5, method="rk4")
    for got, w in zip((S, I, R), want):
        assert _rel(got.cpu().numpy(), w[..., 0]) <= RTOL
    S1, I1, R1, _ = ops.forward(g, xt[B - 1].contiguous(), _tp(P, dev), ops.step_sizes(grid), method)
    assert torch.equal(S[:, (B - 1) * n:], S1) and torch.equal(R[:, (B - 1) * n:], R1)
    # RHS unit on the same graph
    rng = np.random.default_rng(1)
    st = rng.uniform(0, 1.5, (4 * B * n, H)).astype(np.float32)
    st[3 * B * n:, 0] = 0.3; st[3 * B * n:, 1] = 0.2
    p = _tp(P, dev)
    dx = ops.rhs(g, torch.from_numpy(st).to(dev), p["odefunc.linear.weight"], p["odefunc.linear.bias"]).cpu().numpy()
    assert _rel(dx, OC.rhs(rp, ci, n, st, P["odefunc.linear.weight"], P["odefunc.linear.bias"])) <= RTOL


@pytest.mark.parametrize("n,edges,B,H", [(1, [], 2, 64), (3, [(0, 1), (1, 2)], 3, 64), (5, [(0, 4), (2, 2)], 2, 4),
                                         (2, [(0, 1)], 1, 8), (33, [(i, i + 1) for i in range(32)], 2, 64)])
def test_tiny_graphs_and_small_hidden(n, edges, B, H, dev):
    """Degenerate shapes: a single isolated node (nnz = 0), graphs smaller than one 32-row tile, a self-loop,
    H = 4 (one lane per row), a path graph that spills one row into a second tile."""
    import torch
    import gnode_oracle as O
    from gnode import ops
    from gnode.graph import DeviceGraph
    if edges:
        rp, ci = O.csr_from_edges(n, edges)
    else:
        rp, ci = np.zeros(n + 1, np.int32), np.zeros(0, np.int32)
    P = O.init_params(H, seed=n)
    x = O.make_samples(n, B, H, seed=1, n_seeds=1)
    g = DeviceGraph(rp, ci)
    S, I, R, sol = ops.forward(g, torch.from_numpy(x).to(dev).reshape(B * n, 3 + H), _tp(P, dev),
                               ops.step_sizes(ops.time_grid(4, 0.5)), want_sol=True)
    So, Io, Ro, sol_o = O.odeblock_forward_single(x, P, rp, ci, 4, 0.5, return_sol=True)
    for got, w in zip((S, I, R), (So, Io, Ro)):
        assert _rel(got.cpu().numpy(), w[..., 0]) <= RTOL
    q = 3 * B * n                                   # (H = 64: the 4th slabs past grid point 0 are not odeint's, include/gnode.h)
    assert _rel(sol.cpu().numpy()[:, :q], sol_o[:, :q]) <= RTOL and np.array_equal(sol.cpu().numpy()[0, q:], sol_o[0, q:])
    if H != 64:
        assert _rel(sol.cpu().numpy(), sol_o) <= RTOL
    S2, _, _, _ = ops.forward(g, torch.from_numpy(x).to(dev).reshape(B * n, 3 + H), _tp(P, dev),
                              ops.step_sizes(ops.time_grid(4, 0.5)))
    assert _rel(S2.cpu().numpy(), So[..., 0]) <= RTOL


def test_sir_large_state_paths(dev):
    """n = 100 000 nodes: the trajectory state no longer fits a workgroup's LDS (global-memory state path);
    n = 60 000 in coin mode needs > 64 KB of dynamic LDS.  Both bit-exact against the oracle."""
    import gnode_oracle as O
    import oracle_c as OC
    from gnode.graph import DeviceGraph
    from gnode.ode_nn import sir_counts, sir_counts_coins
    n = 100_000
    rp, ci, _ = O.er_graph(n, 300_000, seed=8)
    g = DeviceGraph(rp, ci)
    got = sir_counts(g, [5, 77, 4242], 0.35, 0.2, 24, 8, rng_seed=99).cpu().numpy().astype(np.uint32)
    assert np.array_equal(got, OC.sir_philox(n, rp, ci, [5, 77, 4242], 0.35, 0.2, 24, 8, rng_seed=99))
    n2 = 60_000
    _, _, e2 = O.er_graph(n2, 150_000, seed=9)
    table = O.edge_table(e2)
    coins = np.random.default_rng(0).random(3_000_000)
    S, I, R, used, _ = O.sir_coins(n2, table, [1, 2, 3], 0.4, 0.3, 2, 6, coins)
    cnt, used_gpu = sir_counts_coins(n2, table, [1, 2, 3], 0.4, 0.3, 2, 6, coins[:used + 10])
    c = cnt.cpu().numpy().astype(np.float64)
    assert used_gpu == used and np.array_equal(c[0][None], S) and np.array_equal(c[1][None], I) and np.array_equal(c[2][None], R)


@pytest.mark.parametrize("kind,n,m,seeds,sims,T", [
    ("er-small", 500, 2500, [3, 499], 200, 15),                 # lists in LDS (uint16 ids)
    ("wiki-vote-size", 7066, 100736, [1, 3533], 96, 20),        # lists in LDS, three workgroups per CU
    ("hubs", 3000, 40000, [0, 1, 2999], 64, 12),                # rows longer than 512 edges: walked by the whole workgroup
    ("global-lists", 12000, 60000, [5, 6, 5, 11999], 48, 10),   # lists in the workspace (int32 ids); a duplicated seed
    ("isolated", 300, 40, [7], 64, 6),                          # mostly isolated nodes: the frontier dies out
])
def test_sir_frontier_equals_edge_scan(kind, n, m, seeds, sims, T, dev):
    """The frontier-driven Monte-Carlo kernel against the edge-parallel scan of the same model in the same library
    (gnode_sir_mc_philox_scan: every edge tested every step, what the reference's `isin` does) and against the C
    oracle: identical uint32 counts.  beta high enough that most of the graph burns, gamma low enough that the
    frontier stays large for several steps."""
    import gnode_oracle as O
    import oracle_c as OC
    from gnode.graph import DeviceGraph
    from gnode.ode_nn import sir_counts
    if kind == "hubs":
        rp, ci, _ = O.chung_lu_graph(n, m, exponent=0.95, seed=3)
        assert int(np.max(np.diff(rp))) > 512
    else:
        rp, ci, _ = O.er_graph(n, m, seed=n)
    g = DeviceGraph(rp, ci)
    for beta, gamma, rs in ((0.45, 0.15, 11), (0.05, 0.6, 12)):
        a = sir_counts(g, seeds, beta, gamma, sims, T, rng_seed=rs).cpu().numpy().astype(np.uint32)
        b = sir_counts(g, seeds, beta, gamma, sims, T, rng_seed=rs, edge_scan=True).cpu().numpy().astype(np.uint32)
        assert np.array_equal(a, b), f"{kind}: frontier walk != edge scan (beta={beta})"
        assert np.array_equal(a, OC.sir_philox(n, rp, ci, seeds, beta, gamma, sims, T, rng_seed=rs)), f"{kind}: != oracle"
        assert np.all(a[0, 1:].astype(np.int64) + a[1, 1:] + a[2, 1:] == sims)


def test_c_abi_standalone_host(tmp_path, dev):
    """The boundary is a C ABI: a C++ host with no Python/torch in the process builds against include/gnode.h,
    links libgnode_hip.so and gets the same numbers as the Python host path."""
    import subprocess
    import torch
    from gnode import _lib, ops
    from gnode.graph import DeviceGraph
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cabi_forward")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-I", os.path.join(root, "include"),
                    os.path.join(root, "examples", "cabi_forward.cpp"), "-L", libdir, "-lgnode_hip", "-o", exe], check=True)
    n, B, H, n_steps = 1000, 2, 64, 9
    r = subprocess.run([exe, str(n), str(B)], env=dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", "")),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    line = r.stdout.splitlines()[0]
    sumS = float(line.split("sumS=")[1].split()[0]); sumI = float(line.split("sumI=")[1].split()[0])
    # the same problem through the Python host (same LCG parameter fill)
    s = np.uint32(12345)
    def lcg():
        nonlocal s
        s = np.uint32((int(s) * 1664525 + 1013904223) & 0xFFFFFFFF)
        return np.float32(int(s) >> 8) / np.float32(16777216.0)
    fill = lambda cnt, sc: np.asarray([(lcg() * np.float32(2) - np.float32(1)) * np.float32(sc) for _ in range(cnt)], np.float32)
    P = {"odefunc.linear.weight": fill(H * H, 0.125).reshape(H, H), "odefunc.linear.bias": fill(H, 0.125),
         "linearS1.weight": fill(H, 1.0).reshape(H, 1), "linearS1.bias": fill(H, 1.0),
         "linear3.weight": fill(4 * H, 0.125).reshape(4, H), "linear3.bias": fill(4, 0.125),
         "linearS2.weight": fill(4, 0.5).reshape(1, 4), "linearS2.bias": fill(1, 0.5)}
    rp, col = [0], []
    for i in range(n):
        col += sorted({(i - 7) % n, (i - 1) % n, (i + 1) % n, (i + 7) % n} - {i}); rp.append(len(col))
    x = np.zeros((B * n, 3 + H), np.float32)
    for r_ in range(B * n):
        seeded = (r_ % n) == (r_ // n) * 3
        x[r_, 0], x[r_, 1] = (0, 1) if seeded else (1, 0)
        x[r_, 3], x[r_, 4] = np.float32(0.2) + np.float32(0.05) * np.float32(r_ // n), 0.1
    g = DeviceGraph(np.asarray(rp, np.int32), np.asarray(col, np.int32))
    S, I, R, _ = ops.forward(g, torch.from_numpy(x).to(dev), _tp(P, dev), np.full(n_steps, 0.5, np.float32))
    assert abs(float(S.double().sum()) - sumS) <= 1e-3 * abs(sumS) * 1e-2 + 1e-2
    assert abs(float(I.double().sum()) - sumI) <= 1e-3 * abs(sumI) * 1e-2 + 1e-2


def test_randomized_configurations(dev):
    """Seeded sweep over the corners that fixed cases miss: node counts around tile multiples (31..33, 63..65,
    95..97: the single-launch path's limits), degrees around the hub threshold (96/97), isolated nodes, every H
    class, arbitrary out_rows subsets, with and without the trajectory, Euler and RK4."""
    import torch
    import gnode_oracle as O
    from gnode import ops
    from gnode.graph import DeviceGraph
    rng = np.random.default_rng(2024)
    n_choices = [2, 17, 31, 32, 33, 63, 64, 65, 95, 96, 97, 130, 257]
    for case in range(28):
        n = int(rng.choice(n_choices))
        H = int(rng.choice([4, 8, 16, 24, 32, 64, 64, 64]))
        B = int(rng.integers(1, 4))
        maxTime = int(rng.integers(2, 6))
        deltaT = float(rng.choice([0.25, 0.5, 1.0]))
        method = "rk4" if case % 7 == 6 else "euler"
        m = int(rng.integers(0, 3 * n))
        edges = [(int(a), int(b)) for a, b in rng.integers(0, n, size=(m, 2))]
        if n >= 130 and case % 2 == 0:                      # a hub right at / above the threshold
            d = 96 if case % 4 == 0 else 97
            edges += [(0, j) for j in range(1, d + 1)]
        rp, ci = O.csr_from_edges(n, edges) if edges else (np.zeros(n + 1, np.int32), np.zeros(0, np.int32))
        P = O.init_params(H, seed=case)
        x = O.make_samples(n, B, H, seed=case, n_seeds=1)
        G = len(O.time_grid(maxTime, deltaT))
        sel = np.sort(rng.choice(G, size=int(rng.integers(1, G + 1)), replace=False)).astype(np.int32) if case % 3 else None
        want_sol = bool(case % 2)
        g = DeviceGraph(rp, ci)
        S, I, R, sol = ops.forward(g, torch.from_numpy(x).to(dev).reshape(B * n, 3 + H), _tp(P, dev),
                                   ops.step_sizes(O.time_grid(maxTime, deltaT)), method, sel, want_sol)
        So, Io, Ro, sol_o = O.odeblock_forward_single(x, P, rp, ci, maxTime, deltaT, method=method, return_sol=True)
        idx = np.arange(G) if sel is None else sel
        tag = f"case {case}: n={n} H={H} B={B} T={maxTime} dT={deltaT} {method} sel={None if sel is None else sel.tolist()} sol={want_sol}"
        for got, w in zip((S, I, R), (So, Io, Ro)):
            assert tuple(got.shape) == (len(idx), B * n), tag
            assert _rel(got.cpu().numpy(), w[idx, :, 0]) <= RTOL, tag
        if want_sol:
            from gnode import _lib
            q, sol_h = 3 * B * n, sol.cpu().numpy()
            assert _rel(sol_h[:, :q], sol_o[:, :q]) <= RTOL, tag
            assert np.array_equal(sol_h[0, q:], sol_o[0, q:]), tag
            carries = _lib.load().gnode_sol_carries_neighbour_sums(g.handle, B * n, H, G - 1, len(idx), 0 if ops.PERSIST_DEFAULT else 1) if method == "euler" else 0
            if not carries:      # odeint's own 4th slab at every grid point (the fused H = 64 path keeps A Z_I there instead:
                assert np.array_equal(sol_h[:, q:], sol_o[:, q:]), tag      # test_forward_sol_matches_states)


@pytest.mark.parametrize("name", ["karate", "er120", "loops40"])
def test_dmp_baseline_golden(name, dev):
    """DMP comparison column (reference dmp.py:74-170) through the reference's class surface,
    against vectors the reference class produced.  fp32 1e-5 relative (the GPU's product over a node's
    in-edges runs in the same ascending order; differences come from fused multiply-adds being off / on nowhere)."""
    import scipy.sparse as sp
    from gnode.dmp import DMP_SIR
    d = np.load(os.path.join(GOLD, f"dmp_{name}.npz"))
    n = len(d["rowptr"]) - 1
    W = sp.csr_matrix((d["weights"], d["col"], d["rowptr"]), shape=(n, n))
    m = DMP_SIR(W, d["gamma"])
    out = m.run(d["seeds"].tolist(), int(d["maxTime"]))
    assert tuple(out.shape) == d["out"].shape
    assert _rel(out.cpu().numpy(), d["out"]) <= RTOL
    tot = out.sum(-1).cpu().numpy()
    assert np.allclose(tot, 1.0, atol=1e-5)                  # Ps + Pi + Pr = 1 by construction (dmp.py:129)


def test_dmp_vs_oracle_larger_and_errors(dev):
    import scipy.sparse as sp
    import gnode_oracle as O
    from gnode.dmp import DMP_SIR
    from gnode._lib import GnodeError
    rp, ci, _ = O.er_graph(3000, 20000, seed=4)
    rng = np.random.default_rng(0)
    w = rng.uniform(0.05, 0.4, size=ci.shape[0]).astype(np.float32)      # per-edge weights, not symmetric in value
    gam = rng.uniform(0.1, 0.5, size=3000).astype(np.float32)
    W = sp.csr_matrix((w, ci, rp), shape=(3000, 3000))
    out = DMP_SIR(W, gam).run([7, 100, 2999], 25).cpu().numpy()
    want = O.dmp_sir(rp, ci, w, gam, [7, 100, 2999], 25)
    assert _rel(out, want) <= RTOL
    # a directed (non-symmetric) pattern is refused, loudly
    D = sp.csr_matrix((np.ones(2, np.float32), np.array([1, 2], np.int32), np.array([0, 1, 2, 2], np.int32)), shape=(3, 3))
    with pytest.raises(GnodeError):
        DMP_SIR(D, [0.1, 0.1, 0.1]).run([0], 5)


@pytest.mark.parametrize("name", ["karate", "er150"])
def test_meanfield_baseline_golden(name, dev):
    """Mean-field comparison column through the reference's function surface against vectors
    the reference's own `runge_kutta_order4` produced with scipy's LSODA.  The device integrator is a different
    (explicit, adaptive, much tighter) method: agreement is bounded by LSODA's own tolerance (1.5e-8 per step) --
    1e-6 absolute on probabilities in [0, 1]."""
    import scipy.sparse as sp
    from gnode import ode_nn
    d = np.load(os.path.join(GOLD, f"meanfield_{name}.npz"))
    n = len(d["rowptr"]) - 1
    A = sp.csr_matrix((np.ones(len(d["col"])), d["col"], d["rowptr"]), shape=(n, n))
    I, S, R = ode_nn.runge_kutta_order4(ode_nn.sir, A, n, d["seeds"].tolist(), float(d["beta"]), float(d["gamma"]),
                                        float(d["deltaT"]), int(d["maxTime"]))
    for got, want in ((I, d["I"]), (S, d["S"]), (R, d["R"])):
        assert got.shape == want.shape
        assert np.max(np.abs(got - want)) <= 1e-6
    assert np.max(np.abs(I + S + R - 1.0)) <= 1e-9            # conserved by the RHS; RK methods keep linear invariants


def test_meanfield_vs_oracle_larger(dev):
    import scipy.sparse as sp
    import gnode_oracle as O
    from gnode import ode_nn
    rp, ci, _ = O.er_graph(2000, 12000, seed=9)
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(2000, 2000))
    I, S, R = ode_nn.runge_kutta_order4(ode_nn.sir, A, 2000, [1, 500], 0.03, 0.2, 0.5, 12)
    Io, So, Ro = O.meanfield_rk(rp, ci, [1, 500], 0.03, 0.2, 0.5, 12)
    for got, want in ((I, Io), (S, So), (R, Ro)):
        assert np.max(np.abs(got - want)) <= 1e-6


# ------------------------------------------------------------------ full horizon at the BASELINE sizes vs the REFERENCE
def _full_inputs(d):
    from gnode import synth
    n, H = int(d["n"]), int(d["H"])
    rp, ci = synth.er_csr(n, int(d["m"]), seed=int(d["graph_seed"]))
    return rp, ci, synth.linear_params(H, seed=int(d["param_seed"])), synth.samples(n, 1, H, seed=int(d["sample_seed"]))


@pytest.mark.parametrize("path", _cases("full_"), ids=os.path.basename)
def test_full_horizon_vs_reference(path, dev):
    """59 Euler steps on fb-social- / wiki-vote-sized graphs (configs[1], configs[2]) against what the REFERENCE's
    ODEBlock.forward computed on the same inputs (tests/golden/make_golden_fullsize.py), fp32, plus the
    reference's own float64 run as the yardstick.  Bars: 1e-5 on grid points <= 20; over the whole horizon
    1e-5 plus the distance of the reference's own fp32 from its float64 run (stored in the fixture); and the
    GPU may be no further from the float64 result than that same sum.  The measured numbers are printed and
    recorded in DESIGN.md section 2."""
    import torch
    from gnode import ops
    from gnode.graph import DeviceGraph
    d = dict(np.load(path))
    rp, ci, P, x = _full_inputs(d)
    n, H, maxTime, deltaT = int(d["n"]), int(d["H"]), int(d["maxTime"]), float(d["deltaT"])
    rows, rows64, floor = d["rows"], d["rows64"], float(d["ref_f32_vs_f64_maxabs"])
    g = DeviceGraph(rp, ci)
    S, I, R, _ = ops.forward(g, torch.from_numpy(x).to(dev).reshape(n, 3 + H), _tp(P, dev),
                             ops.step_sizes(ops.time_grid(maxTime, deltaT)))
    early = rows <= 20
    for c, t in zip("SIR", (S, I, R)):
        a = t.cpu().numpy().astype(np.float64)
        e_early = np.max(np.abs(a[rows[early]] - d[c][early]))
        e_all = np.max(np.abs(a[rows] - d[c]))
        e_last = np.max(np.abs(a[59] - d[c][-1]))
        e64 = np.max(np.abs(a[rows64] - d[c + "64"]))
        print(f"{os.path.basename(path)} {c}: gpu vs reference fp32: <=20 steps {e_early:.2e}, all {e_all:.2e}, step 59 {e_last:.2e}; "
              f"gpu vs reference f64 {e64:.2e}; reference fp32 vs its f64 {floor:.2e}")
        assert e_early <= RTOL
        assert e_all <= RTOL + floor
        assert e64 <= RTOL + floor
    # the fused subsample hands the loss exactly the rows the reference's helper keeps; reference loss expression value
    from golden.labels import closed_form_labels
    sub = ops.subsample_rows(maxTime, deltaT)
    Ss, Is, Rs, _ = ops.forward(g, torch.from_numpy(x).to(dev).reshape(n, 3 + H), _tp(P, dev),
                                ops.step_sizes(ops.time_grid(maxTime, deltaT)), "euler", sub)
    assert torch.equal(Ss, S[torch.from_numpy(sub).long().to(dev)])
    y = torch.from_numpy(closed_form_labels(1, n, maxTime)).to(dev).view(n, maxTime, 3)
    pred = torch.stack((Ss, Is, Rs), -1).transpose(0, 1)[:, 1:, :]
    loss = (pred.to(torch.float64) - y[:, 1:, :]).abs().mean().item()
    assert abs(loss - float(d["loss"])) <= 1e-6
