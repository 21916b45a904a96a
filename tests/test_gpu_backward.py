"""GPU: adjoint-Euler backward (A7) against the torch-autograd restatement of torchdiffeq's
odeint_adjoint semantics in the oracle (that boundary is parity-unpinned: torchdiffeq is absent)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-12)


@pytest.mark.parametrize("n,m,B,H,maxTime,deltaT,sub", [
    (60, 200, 2, 8, 4, 0.5, False),
    (150, 700, 3, 64, 5, 0.5, False),
    (150, 700, 2, 64, 6, 0.5, True),       # fused get_sir_t_nodes subsample: grads only at kept rows
    (90, 300, 2, 32, 4, 0.25, False),
    (40, 100, 1, 128, 3, 0.5, False),
    (40, 120, 2, 64, 5, 0.5, False),       # n <= 64: the forward that feeds this backward is the single-launch kernel
    (64, 200, 1, 64, 4, 0.5, True),
    # n <= 64 at H = 64: the whole adjoint sweep is ONE launch (gnode_bwd_tiny.hip)
    (34, 78, 1, 64, 20, 0.5, True),        # karate-sized, the reference's shipped configuration (39 intervals)
    (20, 40, 3, 64, 3, 0.5, False),        # one row tile
    (62, 159, 5, 64, 5, 0.5, False),       # dolphins-sized, two row tiles
    (24, 200, 2, 64, 3, 0.5, False),       # dense: rows longer than the 16 neighbour ids cached in registers
    (33, 60, 2, 64, 1, 0.5, False),        # a single grid point: head + encoder only, no interval
    (150, 700, 2, 64, 1, 0.5, False),      # tiled path, two grid points: the only interval is the last one (recomputing kernel)
    (150, 700, 2, 64, 1.5, 0.5, False),    # three grid points: one interval over kept activations, do_next = 0
    (8, 12, 800, 64, 2, 0.5, False),       # more samples than partial-gradient slots: tiny graphs on the tiled path
    (30, 60, 2, 4, 3, 0.5, False),         # H = 4: one lane per row in the generic one-launch backward
    (30, 60, 2, 24, 3, 0.5, True),         # H = 24: lane groups of 8 with two idle lanes
    (50, 150, 2, 48, 3, 0.5, False),       # 32 < H != 64: the five-launch generic backward
    # more tiles / row blocks than persistent workgroups: accumulators carried across a workgroup's tiles
    (7000, 30000, 2, 64, 2, 0.5, False),
    (60000, 150000, 2, 8, 2, 0.5, True),
])
def test_param_grads_vs_oracle(n, m, B, H, maxTime, deltaT, sub, dev, skewed=False):
    import torch
    import gnode_oracle as O
    from gnode import ops
    from gnode.graph import DeviceGraph
    rp, ci, _ = (O.chung_lu_graph if skewed else O.er_graph)(n, m, seed=n + H)
    P = O.init_params(H, seed=H + 1)
    x = O.make_samples(n, B, H, seed=B)
    grid = O.time_grid(maxTime, deltaT)
    out_rows = ops.subsample_rows(maxTime, deltaT) if sub else None
    n_out = len(out_rows) if sub else len(grid)
    rng = np.random.default_rng(0)
    gs = [rng.normal(size=(n_out, B * n)).astype(np.float32) for _ in range(3)]
    want = O.adjoint_grads_torch(x, P, rp, ci, maxTime, deltaT, *gs, out_rows=out_rows, dtype="float64")
    g = DeviceGraph(rp, ci)
    params = {k: torch.from_numpy(v).to(dev) for k, v in P.items()}
    x2d = torch.from_numpy(x).to(dev).reshape(B * n, 3 + H)
    dts = ops.step_sizes(grid)
    S, I, R, sol = ops.forward(g, x2d, params, dts, "euler", out_rows, want_sol=True)
    gst = [torch.from_numpy(a).to(dev) for a in gs]
    # both forms of the sweep: over the forward's kept activations (where this path keeps any), and recomputing them
    variants = {"kept": ops.backward(g, x2d, params, dts, "euler", out_rows, sol, *gst)}
    if sol.gnode_keep is not None:
        _, _, _, sol_nk = ops.forward(g, x2d, params, dts, "euler", out_rows, want_sol=True, want_keep=False)
        variants["recomputed"] = ops.backward(g, x2d, params, dts, "euler", out_rows, sol_nk, *gst)
    for name, got in variants.items():
        for k in want:
            assert got[k].shape == params[k].shape
            if k == "linearS2.bias":
                continue
            err = _rel(got[k].cpu().numpy(), want[k])
            # fp32 kernels vs a float64 yardstick; sums over up to rows*G terms
            assert err <= 2e-4, f"{name} {k}: rel err {err:.2e}"
    # linearS2.bias: softmax is shift invariant -> exact gradient 0; ours must be ~0 relative to the others
    assert abs(float(got["linearS2.bias"].cpu())) <= 1e-4 * max(1.0, float(np.abs(want["linearS2.weight"]).max()))


@pytest.mark.parametrize("H", [64, 16])
def test_param_grads_on_skewed_graph(H, dev):
    """Hub rows (degree >> hub threshold) in the backward's two transposed gathers."""
    test_param_grads_vs_oracle(600, 6000, 2, H, 4, 0.5, False, dev, skewed=True)


def test_autograd_training_step_reduces_loss(dev):
    """End to end through the reference's call surface: ODEBlock.forward + L1 loss + backward + Adam."""
    import torch
    import scipy.sparse as sp
    import gnode_oracle as O
    from gnode.ode_nn_ngraph_sim import ODEfunc, ODEBlock
    from gnode.ode_nn import get_sir_t_nodes_torch
    n, B, H, maxTime, deltaT = 120, 4, 64, 6, 0.5
    rp, ci, _ = O.er_graph(n, 500, seed=5)
    A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
    torch.manual_seed(0)
    f = ODEfunc(A, 0.2, 0.1, H, dev)
    model = ODEBlock(maxTime, deltaT, n, [0], H, f, dev).to(dev)
    x = torch.from_numpy(O.make_samples(n, B, H, seed=9)).to(dev)
    y = torch.from_numpy(np.random.default_rng(1).dirichlet(np.ones(3), size=(B, n, maxTime))).to(dev)   # float64 labels
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    crit = torch.nn.L1Loss()
    losses = []
    for _ in range(8):
        opt.zero_grad()
        S, I, R = model(x)
        sub = lambda t: get_sir_t_nodes_torch(torch.squeeze(t), maxTime, deltaT, count=False)
        St, It, Rt = sub(S), sub(I), sub(R)
        pred = torch.transpose(torch.cat((St.unsqueeze(-1), It.unsqueeze(-1), Rt.unsqueeze(-1)), -1), 0, 1)[:, 1:, :]
        loss = crit(pred, y.view(-1, y.size(2), y.size(3))[:, 1:, :])        # ode_nn_ngraph_sim.py:234
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(p.grad is not None for k, p in model.named_parameters() if not k.endswith(("ln.weight", "ln.bias")))
    assert losses[-1] < losses[0]


def test_loss_on_one_compartment_only(dev):
    """A loss that uses only I (autograd hands None for the S and R gradients) still back-propagates."""
    import torch
    import scipy.sparse as sp
    import gnode_oracle as O
    from gnode.ode_nn_ngraph_sim import ODEfunc, ODEBlock
    n, H = 50, 64
    rp, ci, _ = O.er_graph(n, 150, seed=3)
    A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
    model = ODEBlock(4, 0.5, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
    x = torch.from_numpy(O.make_samples(n, 2, H, seed=1)).to(dev)
    S, I, R = model(x)
    I.sum().backward()
    g = model.odefunc.linear.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0


def test_forward_and_backward_are_bitwise_reproducible(dev):
    """No float atomics anywhere on the path: two runs on the same inputs give identical bits
    (the reference's GPU scatter_add_ is order-dependent, quirk Q7)."""
    import torch
    import gnode_oracle as O
    from gnode import ops
    from gnode.graph import DeviceGraph
    n, B, H = 2000, 3, 64
    rp, ci, _ = O.chung_lu_graph(n, 20000, seed=4)          # hubs included
    P = {k: torch.from_numpy(v).to(dev) for k, v in O.init_params(H, seed=1).items()}
    x = torch.from_numpy(O.make_samples(n, B, H, seed=2)).to(dev).reshape(B * n, 3 + H)
    g = DeviceGraph(rp, ci)
    dts = ops.step_sizes(ops.time_grid(6, 0.5))
    gs = [torch.randn(len(dts) + 1, B * n, device=dev) for _ in range(3)]
    runs = []
    for _ in range(2):
        S, I, R, sol = ops.forward(g, x, P, dts, want_sol=True)
        grads = ops.backward(g, x, P, dts, "euler", None, sol, *gs)
        runs.append((S.clone(), I.clone(), R.clone(), {k: v.clone() for k, v in grads.items()}))
    for a, b in zip(runs[0][:3], runs[1][:3]):
        assert torch.equal(a, b)
    for k in runs[0][3]:
        assert torch.equal(runs[0][3][k], runs[1][3][k]), k


def test_randomized_backward_configurations(dev):
    """Seeded sweep over the backward's path boundaries: node counts around the one-launch limit (64) and the tile
    sizes, every H class (one-launch generic H <= 32, H = 64, five-launch 32 < H != 64, lane groups with idle
    lanes), hub rows, arbitrary out_rows subsets, several samples."""
    import torch
    import gnode_oracle as O
    from gnode import ops
    from gnode.graph import DeviceGraph
    rng = np.random.default_rng(77)
    n_choices = [3, 15, 16, 17, 33, 63, 64, 65, 100, 140]
    for case in range(18):
        n = int(rng.choice(n_choices))
        H = int(rng.choice([4, 8, 12, 20, 32, 40, 64, 64, 64, 128]))
        B = int(rng.integers(1, 4))
        maxTime = int(rng.integers(1, 5))
        deltaT = float(rng.choice([0.25, 0.5, 1.0]))
        m = int(rng.integers(0, 3 * n))
        edges = [(int(a), int(b)) for a, b in rng.integers(0, n, size=(m, 2))]
        if n >= 100 and case % 2 == 0:
            edges += [(0, j) for j in range(1, 98)]                # a hub row just above the threshold
        rp, ci = O.csr_from_edges(n, edges) if edges else (np.zeros(n + 1, np.int32), np.zeros(0, np.int32))
        P = O.init_params(H, seed=100 + case)
        x = O.make_samples(n, B, H, seed=case, n_seeds=1)
        grid = O.time_grid(maxTime, deltaT)
        G = len(grid)
        sel = np.sort(rng.choice(G, size=int(rng.integers(1, G + 1)), replace=False)).astype(np.int32) if case % 3 else None
        n_out = G if sel is None else len(sel)
        gs = [rng.normal(size=(n_out, B * n)).astype(np.float32) for _ in range(3)]
        want = O.adjoint_grads_torch(x, P, rp, ci, maxTime, deltaT, *gs, out_rows=sel, dtype="float64")
        g = DeviceGraph(rp, ci)
        params = {k: torch.from_numpy(v).to(dev) for k, v in P.items()}
        x2d = torch.from_numpy(x).to(dev).reshape(B * n, 3 + H)
        dts = ops.step_sizes(grid)
        _, _, _, sol = ops.forward(g, x2d, params, dts, "euler", sel, want_sol=True)
        got = ops.backward(g, x2d, params, dts, "euler", sel, sol, *[torch.from_numpy(a).to(dev) for a in gs])
        tag = f"case {case}: n={n} H={H} B={B} T={maxTime} dT={deltaT} sel={None if sel is None else sel.tolist()} nnz={len(ci)}"
        scale = max(float(np.abs(want[k]).max()) for k in want)
        for k in want:
            if k == "linearS2.bias":
                continue
            # relative to the parameter's own largest gradient, with a floor at 1e-3 of the overall scale for
            # parameters whose gradient is (nearly) zero in a degenerate case (e.g. no edges)
            den = max(float(np.abs(want[k]).max()), 1e-3 * scale) + 1e-30
            err = float(np.max(np.abs(got[k].cpu().numpy().astype(np.float64) - want[k]))) / den
            assert err <= 2e-4, f"{tag}: {k} rel err {err:.2e}"


@pytest.mark.parametrize("H", [8, 16, 32, 64])
def test_every_degree_forward_and_backward(H, dev):
    """Threshold graph (i ~ j iff i + j >= n): node i has degree ~i, so one graph walks every chunk boundary of
    the gathers (8 / 16 ids per chunk, 4- and 8-row batches, the prefetch of the next chunk) and the hub threshold."""
    import torch
    import gnode_oracle as O
    from gnode import ops
    from gnode.graph import DeviceGraph
    n, B, maxTime, deltaT = 130, 2, 2, 0.5
    edges = [(i, j) for i in range(n) for j in range(i + 1, n) if i + j >= n]
    rp, ci = O.csr_from_edges(n, edges)
    deg = np.diff(rp)
    assert deg.min() == 0 and deg.max() >= 97 and len(set(deg.tolist())) >= 100
    P = O.init_params(H, seed=3)
    for k in ("odefunc.linear.weight",):
        P[k] = (P[k] * 0.2).astype(np.float32)            # dense rows: keep the sums out of sigmoid saturation
    x = O.make_samples(n, B, H, seed=5)
    x[..., 3] *= 0.02                                      # beta: degree-100 rows would otherwise be stiff at dt = 0.5
    grid = O.time_grid(maxTime, deltaT)
    g = DeviceGraph(rp, ci)
    params = {k: torch.from_numpy(v).to(dev) for k, v in P.items()}
    x2d = torch.from_numpy(x).to(dev).reshape(B * n, 3 + H)
    dts = ops.step_sizes(grid)
    S, I, R, sol = ops.forward(g, x2d, params, dts, "euler", None, want_sol=True)
    So, Io, Ro = O.odeblock_forward_single(x, P, rp, ci, maxTime, deltaT)
    for got, want in zip((S, I, R), (So, Io, Ro)):
        assert _rel(got.cpu().numpy(), want[:, :, 0]) <= 1e-5
    rng = np.random.default_rng(1)
    gs = [rng.normal(size=(len(grid), B * n)).astype(np.float32) for _ in range(3)]
    want = O.adjoint_grads_torch(x, P, rp, ci, maxTime, deltaT, *gs, dtype="float64")
    got = ops.backward(g, x2d, params, dts, "euler", None, sol, *[torch.from_numpy(a).to(dev) for a in gs])
    scale = max(float(np.abs(want[k]).max()) for k in want)
    for k in want:
        if k == "linearS2.bias":
            continue
        # a parameter whose exact gradient is ~0 here (dead relu units) is held to the overall gradient scale
        den = max(float(np.abs(want[k]).max()), 1e-3 * scale) + 1e-30
        err = float(np.max(np.abs(got[k].cpu().numpy().astype(np.float64) - want[k]))) / den
        assert err <= 2e-4, f"{k}: rel err {err:.2e}"


def test_backward_full_size_properties(dev):
    """BASELINE's 75k-node graph is too large for the float64 autograd checker to be quick; size-independent
    properties instead: the gradient map is LINEAR in the output cotangents (grads(2g) = 2 grads(g),
    grads(g1 + g2) = grads(g1) + grads(g2)) and bitwise reproducible run to run."""
    import torch
    import gnode_oracle as O
    from gnode import ops
    from gnode.graph import DeviceGraph
    n, m, B, H, maxTime, deltaT = 75000, 500000, 1, 64, 3, 0.5
    rp, ci, _ = O.er_graph(n, m, seed=0)
    P = O.init_params(H, seed=0)
    x = O.make_samples(n, B, H, seed=4)
    g = DeviceGraph(rp, ci)
    params = {k: torch.from_numpy(v).to(dev) for k, v in P.items()}
    x2d = torch.from_numpy(x).to(dev).reshape(B * n, 3 + H)
    dts = ops.step_sizes(O.time_grid(maxTime, deltaT))
    G = len(dts) + 1
    _, _, _, sol = ops.forward(g, x2d, params, dts, "euler", None, want_sol=True)
    gen = torch.Generator(device="cpu").manual_seed(0)
    g1 = [torch.randn(G, B * n, generator=gen).to(dev) for _ in range(3)]
    g2 = [torch.randn(G, B * n, generator=gen).to(dev) for _ in range(3)]
    r1 = ops.backward(g, x2d, params, dts, "euler", None, sol, *g1)
    r1b = ops.backward(g, x2d, params, dts, "euler", None, sol, *g1)
    r2 = ops.backward(g, x2d, params, dts, "euler", None, sol, *g2)
    rs = ops.backward(g, x2d, params, dts, "euler", None, sol, *[a + b for a, b in zip(g1, g2)])
    rd = ops.backward(g, x2d, params, dts, "euler", None, sol, *[2 * a for a in g1])
    for k in r1:
        assert torch.equal(r1[k], r1b[k]), k                                   # bitwise reproducible
        if k == "linearS2.bias":                                               # exact gradient 0 (softmax shift invariance): pure round-off
            continue
        scale = float(max(r1[k].abs().max(), r2[k].abs().max())) + 1e-30
        assert float((rd[k] - 2 * r1[k]).abs().max()) <= 2e-5 * scale, k       # homogeneity (x2 is exact up to reduction order)
        assert float((rs[k] - (r1[k] + r2[k])).abs().max()) <= 2e-4 * scale, k  # additivity, fp32 sums over 75k rows x 6 points


def test_kept_activations_are_the_forwards_and_optional(dev):
    """The `keep` buffer of gnode_forward_f32 (include/gnode.h): outputs and trajectory do not depend on whether it is given;
    its tables ARE sigmoid(W y_k + b) of the trajectory's S / I rows (checked against torch on the saved rows); and the
    backward over it agrees with the recomputing backward to fp32 round-off -- on a graph with hub rows and a ragged
    last tile, with a subsampled output grid."""
    import torch
    import gnode_oracle as O
    from gnode import _lib, ops
    from gnode.graph import DeviceGraph
    n, B, H, maxTime, deltaT = 2003, 3, 64, 6, 0.5
    rp, ci, _ = O.chung_lu_graph(n, 24000, seed=9)
    P = {k: torch.from_numpy(v).to(dev) for k, v in O.init_params(H, seed=5).items()}
    x = torch.from_numpy(O.make_samples(n, B, H, seed=6)).to(dev).reshape(B * n, 3 + H)
    g = DeviceGraph(rp, ci)
    assert int(np.diff(rp).max()) > 96                         # hub rows present
    dts = ops.step_sizes(ops.time_grid(maxTime, deltaT))
    rows_out = ops.subsample_rows(maxTime, deltaT)
    G, rows = len(dts) + 1, B * n
    S1, I1, R1, sol1 = ops.forward(g, x, P, dts, "euler", rows_out, want_sol=True)
    S0, I0, R0, sol0 = ops.forward(g, x, P, dts, "euler", rows_out, want_sol=True, want_keep=False)
    keep = sol1.gnode_keep
    assert keep is not None and sol0.gnode_keep is None
    assert keep.numel() * 4 == _lib.load().gnode_forward_keep_bytes(g.handle, rows, H, G - 1, len(rows_out))
    for a, b in ((S1, S0), (I1, I0), (R1, R0)):
        assert torch.equal(a, b)
    assert torch.equal(sol1[:, :3 * rows], sol0[:, :3 * rows])
    kv = keep.view(G, 3, rows + 1, H)
    W, b = P["odefunc.linear.weight"], P["odefunc.linear.bias"]
    for k in range(G - 1):                                                        # Z_S(y_k): steps 0 .. G-2 evaluate it
        z = torch.sigmoid(sol1[k, :rows].double() @ W.T.double() + b.double())
        assert float((kv[k, 0, :rows].double() - z).abs().max()) <= 2e-5     # structural check (right table, right grid point): fp32 pre-activations reach tens on this hub graph
    for k in range(G):                                                            # Z_I(y_k): every grid point's gather table
        z = torch.sigmoid(sol1[k, rows:2 * rows].double() @ W.T.double() + b.double())
        assert float((kv[k, 1, :rows].double() - z).abs().max()) <= 2e-5
        assert float(kv[k, 1, rows].abs().max()) == 0.0                           # the table's zero row
    for k in range(1, G - 1):                                                     # P_S(y_k) = A Z_I * Z_S (1 - Z_S): the no-keep forward's
        ai, zs = sol0[k, 3 * rows:], kv[k, 0, :rows]                              # 4th slab holds the A Z_I factor
        want = ai * (zs * (1.0 - zs))
        assert float((kv[k, 2, :rows] - want).abs().max()) <= 1e-6 * float(want.abs().max() + 1.0)
    gs = [torch.randn(len(rows_out), rows, device=dev) for _ in range(3)]
    a = ops.backward(g, x, P, dts, "euler", rows_out, sol1, *gs)
    r = ops.backward(g, x, P, dts, "euler", rows_out, sol0, *gs)
    with pytest.raises(_lib.GnodeError):                # a trajectory produced WITH keep must come back with it
        ops.backward(g, x, P, dts, "euler", rows_out, sol1, *gs, keep=None)
    for k in a:
        if k == "linearS2.bias":
            continue
        scale = float(r[k].abs().max()) + 1e-30
        assert float((a[k] - r[k]).abs().max()) <= 2e-5 * scale, (k, float((a[k] - r[k]).abs().max()) / scale)
    # a short keep buffer is refused, not overrun
    with pytest.raises(_lib.GnodeError):
        ops.backward(g, x, P, dts, "euler", rows_out, sol1, *gs, keep=keep[: keep.numel() // 2])


@pytest.mark.parametrize("keep", [True, False], ids=["kept", "recompute"])
@pytest.mark.parametrize("name", ["adjoint_karate_H64_T20", "adjoint_er200_H64_T6", "adjoint_loops40_H8_T5",
                                  "adjoint_fb_H64_T30", "adjoint_wiki_H64_T30"])
def test_training_gradient_vs_reference_classes(name, keep, dev, monkeypatch):
    """The whole training gradient through the PRODUCT's call surface (ODEBlock mirror -> fused forward, the L1 loss
    op, the adjoint backward) against what the REFERENCE's ODEBlock / ODEfunc / loss expression produced under the
    restated adjoint rule, in float64 (tests/golden/make_golden_adjoint.py).  karate: the one-launch tiny paths;
    er200: the tiled path over kept activations; loops40 (H = 8, self-loops): the generic path; fb / wiki: the FULL 59-interval
    horizon the reference trains through (ode_nn_ngraph_sim.py:168, 234-246) at fb-social's / wiki-vote's node and edge counts,
    over kept activations and with the recomputing backward."""
    import os
    import torch
    import scipy.sparse as sp
    import gnode_oracle as O
    from gnode import synth
    from gnode.autograd import l1_loss_sum
    from gnode.ode_nn_ngraph_sim import ODEBlock, ODEfunc
    from gnode import ops
    from golden.labels import closed_form_labels
    d = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz")))
    n, B, H, maxTime, deltaT = int(d["n"]), int(d["B"]), int(d["H"]), int(d["maxTime"]), float(d["deltaT"])
    monkeypatch.setattr(ops, "KEEP_DEFAULT", keep)
    full = "graph_seed" in d                                                  # 59 intervals: graph by seed, 3 stored output rows
    rp, ci = synth.er_csr(n, int(d["m"]), seed=int(d["graph_seed"])) if full else O.csr_from_edges(n, d["edges"])
    A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
    P = synth.linear_params(H, seed=int(d["param_seed"]))
    model = ODEBlock(maxTime, deltaT, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
    model.load_state_dict({**model.state_dict(), **{k: torch.from_numpy(v) for k, v in P.items()}})
    x = torch.from_numpy(synth.samples(n, B, H, seed=int(d["sample_seed"]))).to(dev)
    y = torch.from_numpy(closed_form_labels(B, n, maxTime).reshape(B * n, maxTime, 3)).to(dev)
    S, I, R = model(x, out_rows=ops.subsample_rows(maxTime, deltaT))
    for c, got in zip("SIR", (S, I, R)):                                      # forward: the reference's float64 outputs
        got = got.detach()[..., 0].double().cpu()
        if full: got = got[torch.from_numpy(d["rows_kept"]).long()]
        # (full horizon: the fp32 floor after 59 steps, DESIGN section 2 -- the reference's own fp32 sits 1e-5 from its float64)
        assert float((got - torch.from_numpy(d[c])).abs().max()) <= (2e-5 if full else 1e-5)
    loss = l1_loss_sum(S, I, R, y, 1) / (B * n * (maxTime - 1) * 3)
    assert abs(float(loss.detach()) - float(d["loss"])) <= 1e-6
    loss.backward()
    named = dict(model.named_parameters())
    for k in P:
        want = d["G:" + k]
        if k == "linearS2.bias":                                              # exact gradient 0 (softmax shift invariance)
            assert float(named[k].grad.abs().max()) <= 1e-6
            continue
        err = _rel(named[k].grad.cpu().numpy(), want)
        assert err <= 2e-4, f"{k}: rel err {err:.2e}"
        if full:                                                              # yardstick: the reference's own fp32 run of the same rule
            print(f"[{name} {'kept' if keep else 'recompute'}] {k}: GPU vs reference float64 {err:.2e}; reference fp32 vs its float64 "
                  f"{_rel(d['G32:' + k], want):.2e}")


def test_double_backward_through_one_forward(dev):
    """loss.backward(retain_graph=True) twice through the same forward node: the kept activations stay
    with the autograd context, so the second sweep sees the same buffers and gives the same gradient bits."""
    import torch
    import scipy.sparse as sp
    from gnode import synth
    from gnode.ode_nn_ngraph_sim import ODEBlock, ODEfunc
    n, m, B, H = 300, 1200, 2, 64
    rp, ci = synth.er_csr(n, m, seed=3)
    A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
    model = ODEBlock(6, 0.5, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
    x = torch.from_numpy(synth.samples(n, B, H, seed=4)).to(dev)
    S, I, R = model(x)
    loss = (S * 0.3 + I * 0.5 - R * 0.2).sum()
    loss.backward(retain_graph=True)
    g1 = {k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None}
    model.zero_grad()
    loss.backward()
    for k, v in model.named_parameters():
        if v.grad is not None:
            assert torch.equal(v.grad, g1[k]), k
