"""The persistent one-launch integration of mid-size graphs (csrc/gnode_pers64.hip) against the one-launch-per-step
form of the same library: outputs, trajectory and kept activations must agree BIT FOR BIT (same arithmetic, same
summation order, same MFMA chains), for every placement shape the plan produces -- one sample per XCD, several samples
per XCD, a sample across 2 / 4 / 8 XCDs -- in inference and in training.  Parity with the reference itself is held by
test_gpu_parity.py / test_gpu_backward.py, which run through the persistent path by default at these sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _setup(n, m, B, seed, dev, tail=0.0):
    import torch
    from gnode import ops, synth
    from gnode.graph import DeviceGraph
    # tail > 0: Chung-Lu degrees with the reference datasets' tails (hub rows: longest ~740 at fb-social size, ~1 020 at wiki-vote size)
    rp, ci = synth.heavy_tail_csr(n, m, tail, seed=seed) if tail else synth.er_csr(n, m, seed=seed)
    g = DeviceGraph(rp, ci)
    P = {k: torch.from_numpy(v).to(dev) for k, v in synth.linear_params(64, seed=seed + 1).items()}
    x = torch.from_numpy(synth.samples(n, B, 64, seed=seed + 2)).to(dev).reshape(B * n, 67)
    return g, P, x


# (n, undirected edges, samples, expected XCDs per sample)   [first hub shape: 32 rows per workgroup so that its 738-edge row costs one
# round of segment sums per step instead of two -> 60 workgroups on 2 XCDs]
SHAPES = [(1893, 13835, 1, 4), (1893, 13835, 8, 1), (1893, 13835, 3, 2), (600, 2400, 5, 1), (300, 1500, 16, 1),
          (7066, 100736, 1, 8), (7066, 100736, 2, 4), (130, 500, 2, 1), (4099, 30000, 2, 4)]


HUB_SHAPES = [(1893, 13835, 1, 2, 0.8), (1893, 13835, 8, 1, 0.8), (7066, 100736, 1, 8, 0.5), (7066, 100736, 2, 4, 0.5), (500, 6000, 3, 1, 0.9)]


@pytest.mark.parametrize("n,m,B,span,tail", [s + (0.0,) for s in SHAPES] + HUB_SHAPES)
def test_persistent_inference_bitwise(n, m, B, span, tail, dev):
    import torch
    from gnode import ops
    g, P, x = _setup(n, m, B, 7, dev, tail)
    maxTime, deltaT = 30, 0.5
    dts = ops.step_sizes(ops.time_grid(maxTime, deltaT))
    path, plan = ops.forward_path(g, B * n, 64, len(dts))
    assert path == 2 and plan[2] == span, (path, plan)
    for out_rows in (None, ops.subsample_rows(maxTime, deltaT)):
        S0, I0, R0, _ = ops.forward(g, x, P, dts, "euler", out_rows, persist=False)
        S1, I1, R1, _ = ops.forward(g, x, P, dts, "euler", out_rows, persist=True)
        assert ops.forward_status() == 0
        for a, b in ((S0, S1), (I0, I1), (R0, R1)):
            assert torch.equal(a, b), float((a - b).abs().max())
    # the same launch again on the same workspace shape (fresh tickets / flags every call) and with unequal steps
    dts2 = np.asarray([0.5, 0.25, 1.0, 0.5, 0.125, 0.5, 0.5], dtype=np.float32)
    S0, I0, R0, _ = ops.forward(g, x, P, dts2, "euler", np.asarray([0, 3, 7], dtype=np.int32), persist=False)
    S1, I1, R1, _ = ops.forward(g, x, P, dts2, "euler", np.asarray([0, 3, 7], dtype=np.int32), persist=True)
    assert ops.forward_status() == 0
    assert torch.equal(S0, S1) and torch.equal(I0, I1) and torch.equal(R0, R1)


@pytest.mark.parametrize("keep", [True, False], ids=["kept", "nokeep"])
@pytest.mark.parametrize("n,m,B,span,tail", [(1893, 13835, 1, 4, 0.0), (1893, 13835, 8, 1, 0.0), (600, 2400, 5, 1, 0.0), (7066, 100736, 1, 8, 0.0),
                                             (1893, 13835, 2, 4, 0.8), (7066, 100736, 1, 8, 0.5)])
def test_persistent_training_forward_bitwise(n, m, B, span, tail, keep, dev):
    import torch
    from gnode import ops
    g, P, x = _setup(n, m, B, 11, dev, tail)
    maxTime, deltaT = 12, 0.5
    dts = ops.step_sizes(ops.time_grid(maxTime, deltaT))
    rows_out = ops.subsample_rows(maxTime, deltaT)
    assert ops.forward_path(g, B * n, 64, len(dts), len(rows_out), want_sol=True)[0] == 2
    rows = B * n
    # poison what each form leaves unwritten the same way, so that whole buffers can be compared
    outs = []
    for persist in (False, True):
        S, I, R, sol = ops.forward(g, x, P, dts, "euler", rows_out, want_sol=True, want_keep=keep, persist=persist)
        assert ops.forward_status() == 0
        outs.append((S, I, R, sol, sol.gnode_keep, sol.gnode_info))
    (S0, I0, R0, sol0, k0, i0), (S1, I1, R1, sol1, k1, i1) = outs
    assert i0 == i1 == (2 if keep else 1)
    assert torch.equal(S0, S1) and torch.equal(I0, I1) and torch.equal(R0, R1)
    G = len(dts) + 1
    assert torch.equal(sol0[:, :3 * rows], sol1[:, :3 * rows])                         # odeint's S, I, R slabs at every grid point
    assert torch.equal(sol0[0, 3 * rows:], sol1[0, 3 * rows:])                         # beta, gamma at grid point 0
    if keep:
        st = (rows + 1) * 64
        a, b = k0.view(G, 3, st), k1.view(G, 3, st)
        assert torch.equal(a[:G - 1, 0, :rows * 64], b[:G - 1, 0, :rows * 64])         # Z_S(y_k), k < n_steps
        assert torch.equal(a[:, 1], b[:, 1])                                           # Z_I tables incl. their zero rows
        assert torch.equal(a[1:G - 1, 2, :rows * 64], b[1:G - 1, 2, :rows * 64])       # P_S(y_k), 1 <= k < n_steps
    else:
        assert torch.equal(sol0[1:G - 1, 3 * rows:], sol1[1:G - 1, 3 * rows:])         # A Z_I(y_k) parked in the 4th slabs


@pytest.mark.parametrize("n,m,B,launches,tail", [(1893, 13835, 1, 1, 0.0), (1893, 13835, 2, 1, 0.0), (1893, 13835, 8, 2, 0.0), (600, 2400, 5, 1, 0.0),
                                                 (7066, 100736, 1, 1, 0.0), (130, 500, 3, 1, 0.0), (1893, 13835, 1, 1, 0.8), (1893, 13835, 8, 2, 0.8),
                                                 (7066, 100736, 1, 1, 0.5)])
def test_persistent_backward(n, m, B, launches, tail, dev):
    """The adjoint sweep as ONE persistent launch (csrc/gnode_pers64_bwd.hip; `launches` consecutive ones when the batch does
    not fit one resident grid) against one launch per interval: the same per-row VJPs, rows enter the parameter sums in a
    different order -> 1e-5 of each gradient's scale; bitwise reproducible run to run.  Forwards persistent and per-step
    (bit-identical trajectories, checked above) are crossed with both backwards."""
    import torch
    from gnode import ops
    g, P, x = _setup(n, m, B, 5, dev, tail)
    maxTime, deltaT = 30, 0.5
    dts = ops.step_sizes(ops.time_grid(maxTime, deltaT))
    rows_out = ops.subsample_rows(maxTime, deltaT)
    gs = [torch.randn(len(rows_out), B * n, device=dev) for _ in range(3)]
    grads = {}
    for pf in (False, True):
        S, I, R, sol = ops.forward(g, x, P, dts, "euler", rows_out, want_sol=True, persist=pf)
        for pb in (False, True):
            grads[pf, pb] = ops.backward(g, x, P, dts, "euler", rows_out, sol, *gs, persist=pb)
    again = ops.backward(g, x, P, dts, "euler", rows_out, sol, *gs, persist=True)
    for k in grads[False, False]:
        ref = grads[False, False][k]
        # (linearS2.bias: the exact gradient is 0 -- softmax shift invariance -- what is computed is rounding noise: its scale
        #  is that of the sums it cancels from, linearS2.weight's)
        scale = float((grads[False, False]["linearS2.weight"] if k == "linearS2.bias" else ref).abs().max()) + 1e-30
        assert torch.equal(grads[True, False][k], ref), k                      # same trajectory bits -> same per-interval gradient bits
        assert torch.equal(grads[True, True][k], grads[False, True][k]), k
        assert torch.equal(grads[True, True][k], again[k]), k                  # run-to-run reproducible
        err = float((grads[True, True][k] - ref).abs().max()) / scale
        assert err <= 1e-5, (k, err)


def test_sol_info_pairing_is_checked(dev):
    """ABI 220: the backward refuses a trajectory / keep pair that does not belong together"""
    import torch
    from gnode import ops, _lib
    n, m, B = 600, 2400, 2
    g, P, x = _setup(n, m, B, 3, dev)
    dts = ops.step_sizes(ops.time_grid(6, 0.5))
    gs = [torch.randn(len(dts) + 1, B * n, device=dev) for _ in range(3)]
    S, I, R, sol_k = ops.forward(g, x, P, dts, "euler", None, want_sol=True, want_keep=True)
    S, I, R, sol_n = ops.forward(g, x, P, dts, "euler", None, want_sol=True, want_keep=False)
    keep = sol_k.gnode_keep
    sol_k.gnode_keep = None
    with pytest.raises(_lib.GnodeError):                    # kept trajectory, no buffer (python-side guard gone: the C ABI says no)
        ops.backward(g, x, P, dts, "euler", None, sol_k, *gs, keep="auto")
    with pytest.raises(_lib.GnodeError):                    # a keep buffer with a trajectory whose forward filled none
        ops.backward(g, x, P, dts, "euler", None, sol_n, *gs, keep=keep)
    sol_k.gnode_keep = keep
    ops.backward(g, x, P, dts, "euler", None, sol_k, *gs)
    ops.backward(g, x, P, dts, "euler", None, sol_n, *gs)


@pytest.mark.parametrize("n,m", [(34, 78), (1893, 13835)])
def test_persistent_training_step_under_graph_replay(n, m, dev):
    """forward + loss + adjoint sweep captured into ONE HIP graph (what the trainer does) and replayed several times: every
    replay must re-run the persistent launches (fresh tickets and flags -- they are zeroed by a kernel node: a memset node was
    not re-executed correctly on the second replay) and reproduce the eager gradient bits."""
    import torch
    import scipy.sparse as sp
    from gnode import ops, synth
    from gnode.autograd import l1_loss_sum
    from gnode.ode_nn_ngraph_sim import ODEBlock, ODEfunc
    H, maxTime = 64, 20
    rp, ci = synth.er_csr(n, m, seed=1)
    A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
    model = ODEBlock(maxTime, 0.5, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
    x = torch.from_numpy(synth.samples(n, 1, H, seed=2)).to(dev)
    y = torch.from_numpy(np.random.default_rng(0).dirichlet(np.ones(3), size=(n, maxTime))).to(dev)
    rows = ops.subsample_rows(maxTime, 0.5)
    assert ops.forward_path(model.odefunc.graph, n, H, 2 * maxTime - 1, len(rows), want_sol=True)[0] == 2

    def step():
        for p in model.parameters():
            if p.grad is not None:
                p.grad.zero_()
        S, I, R = model(x, out_rows=rows)
        l1_loss_sum(S, I, R, y, 1).backward()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    want = {k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None}
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for rep in range(4):
        for p in model.parameters():
            if p.grad is not None:
                p.grad.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert ops.forward_status() == 0, rep
        for k, v in model.named_parameters():
            if v.grad is not None:
                assert torch.equal(v.grad, want[k]), (rep, k)


# ---- small hidden sizes (csrc/gnode_persg.hip): the multi-graph launcher's H = 8 (monitorer-ngraphs.py:20), and 16 / 32
def _setup_h(n, m, B, H, seed, dev, tail=0.0):
    import torch
    from gnode import synth
    from gnode.graph import DeviceGraph
    rp, ci = synth.heavy_tail_csr(n, m, tail, seed=seed) if tail else synth.er_csr(n, m, seed=seed)
    g = DeviceGraph(rp, ci)
    P = {k: torch.from_numpy(v).to(dev) for k, v in synth.linear_params(H, seed=seed + 1).items()}
    x = torch.from_numpy(synth.samples(n, B, H, seed=seed + 2)).to(dev).reshape(B * n, 3 + H)
    return g, P, x


SMALL_H = [(22125, 250000, 1, 8, 0.0), (7066, 100736, 1, 8, 0.5), (1893, 13835, 3, 8, 0.8), (62, 159, 1, 8, 0.0), (300, 1500, 5, 8, 0.0),
           (7066, 100736, 1, 16, 0.5), (4000, 30000, 2, 16, 0.0), (1893, 13835, 2, 32, 0.8), (5000, 40000, 1, 32, 0.0)]


@pytest.mark.parametrize("n,m,B,H,tail", SMALL_H)
def test_small_hidden_persistent_forward_bitwise(n, m, B, H, tail, dev):
    """one persistent launch == one launch per Euler step, bit for bit (outputs and trajectory), hub rows included"""
    import torch
    from gnode import ops
    g, P, x = _setup_h(n, m, B, H, 11, dev, tail)
    maxTime, deltaT = 20, 0.5
    dts = ops.step_sizes(ops.time_grid(maxTime, deltaT))
    assert ops.forward_path(g, B * n, H, len(dts))[0] == 3
    assert ops.forward_path(g, B * n, H, len(dts), persist=False)[0] == 0
    for out_rows in (None, ops.subsample_rows(maxTime, deltaT)):
        for want_sol in (False, True):
            r0 = ops.forward(g, x, P, dts, "euler", out_rows, want_sol=want_sol, persist=False)
            r1 = ops.forward(g, x, P, dts, "euler", out_rows, want_sol=want_sol, persist=True)
            assert ops.forward_status() == 0
            for a, b in zip(r0[:3], r1[:3]):
                assert torch.equal(a, b), float((a - b).abs().max())
            if want_sol:
                assert torch.equal(r0[3], r1[3])
    dts2 = np.asarray([0.5, 0.25, 1.0, 0.5, 0.125], dtype=np.float32)
    r0 = ops.forward(g, x, P, dts2, "euler", np.asarray([0, 2, 5], dtype=np.int32), persist=False)
    r1 = ops.forward(g, x, P, dts2, "euler", np.asarray([0, 2, 5], dtype=np.int32), persist=True)
    assert all(torch.equal(a, b) for a, b in zip(r0[:3], r1[:3]))


@pytest.mark.parametrize("n,m,B,H,tail", SMALL_H)
def test_small_hidden_persistent_backward(n, m, B, H, tail, dev):
    """the adjoint sweep as one persistent launch against one launch per interval: same per-row VJPs, parameter sums in a
    different order -> 1e-5 of each gradient's scale; bitwise reproducible run to run"""
    import torch
    from gnode import ops
    g, P, x = _setup_h(n, m, B, H, 13, dev, tail)
    maxTime, deltaT = 20, 0.5
    dts = ops.step_sizes(ops.time_grid(maxTime, deltaT))
    rows_out = ops.subsample_rows(maxTime, deltaT)
    gs = [torch.randn(len(rows_out), B * n, device=dev) for _ in range(3)]
    S, I, R, sol = ops.forward(g, x, P, dts, "euler", rows_out, want_sol=True)
    ref = ops.backward(g, x, P, dts, "euler", rows_out, sol, *gs, persist=False)
    got = ops.backward(g, x, P, dts, "euler", rows_out, sol, *gs, persist=True)
    again = ops.backward(g, x, P, dts, "euler", rows_out, sol, *gs, persist=True)
    for k in ref:
        scale = float((ref["linearS2.weight"] if k == "linearS2.bias" else ref[k]).abs().max()) + 1e-30
        assert torch.equal(got[k], again[k]), k
        err = float((got[k] - ref[k]).abs().max()) / scale
        assert err <= 1e-5, (k, err)
    # every grid point emitted (head VJP in every interval) and unequal steps
    dts2 = np.asarray([0.5, 0.25, 1.0, 0.5, 0.125], dtype=np.float32)
    gs2 = [torch.randn(6, B * n, device=dev) for _ in range(3)]
    S, I, R, sol = ops.forward(g, x, P, dts2, "euler", None, want_sol=True)
    ref = ops.backward(g, x, P, dts2, "euler", None, sol, *gs2, persist=False)
    got = ops.backward(g, x, P, dts2, "euler", None, sol, *gs2, persist=True)
    for k in ref:
        scale = float((ref["linearS2.weight"] if k == "linearS2.bias" else ref[k]).abs().max()) + 1e-30
        assert float((got[k] - ref[k]).abs().max()) / scale <= 1e-5, k


def test_status_words_are_defined_after_every_call(dev):
    """gnode_forward_status / gnode_backward_status: 0 after persistent and per-step calls alike (the backward clears the
    give-up word when it ran no persistent sweep: a fresh workspace holds anything)"""
    import torch
    from gnode import ops
    for H, n, m in ((64, 600, 2400), (8, 900, 4000)):
        g, P, x = _setup_h(n, m, 2, H, 17, dev)
        dts = ops.step_sizes(ops.time_grid(6, 0.5))
        gs = [torch.randn(len(dts) + 1, 2 * n, device=dev) for _ in range(3)]
        for persist in (True, False):
            torch.empty(64 << 20, dtype=torch.uint8, device=dev).fill_(0xAB)        # dirty the allocator's blocks
            S, I, R, sol = ops.forward(g, x, P, dts, "euler", None, want_sol=True, persist=persist)
            assert ops.forward_status() == 0
            ops.backward(g, x, P, dts, "euler", None, sol, *gs, persist=persist)
            assert ops.backward_status() == 0
