"""GPU: the promises include/gnode.h makes about the launch functions -- no allocation, no synchronisation, nothing
retained in the graph handle -- checked where breaking them is observable: stream capture on FIRST use on a graph
with hub rows (an allocation or a synchronisation aborts the capture), a larger batch on the same handle in between
(a handle-owned scratch buffer would be reallocated under the captured graph), and two streams sharing one handle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _hub_graph(n=1200, m=9000, seed=4):
    import gnode_oracle as O
    rp, ci, _ = O.chung_lu_graph(n, m, exponent=0.9, seed=seed)
    assert int(np.max(np.diff(rp))) > 96, "the test graph must have hub rows (degree above the hub threshold)"
    return rp, ci


def _tp(P, dev):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in P.items()}


def test_workspace_sizes_follow_the_graph(dev):
    import gnode_oracle as O
    from gnode import _lib
    from gnode.graph import DeviceGraph
    lib = _lib.load()
    rp, ci = _hub_graph()
    hub = DeviceGraph(rp, ci)
    rp2, ci2, _ = O.er_graph(1200, 9000, seed=1)
    flat = DeviceGraph(rp2, ci2)
    rows, H = 4 * 1200, 64
    for fn, extra in ((lib.gnode_rhs_workspace_bytes, ()), (lib.gnode_forward_workspace_bytes, (0,)),
                      (lib.gnode_backward_workspace_bytes, ())):
        a, b = fn(hub.handle, rows, H, *extra), fn(flat.handle, rows, H, *extra)
        assert b > 0 and a > b, "hub scratch must be part of the caller's workspace"
        assert fn(hub.handle, 2 * rows, H, *extra) > a
    assert lib.gnode_forward_workspace_bytes(flat.handle, rows, H, 0) >= 5 * rows * H * 4


@pytest.mark.parametrize("H,want_sol", [(64, False), (64, True), (8, False)])
def test_capture_on_first_use_with_hub_rows(H, want_sol, dev):
    """gnode_forward_f32 captured into a HIP graph with no eager warm-up on this handle; then a LARGER batch runs eagerly
    on the same handle; then the captured graph is replayed and must still give the eager result of its own batch."""
    import torch
    import gnode_oracle as O
    from gnode import ops
    from gnode.graph import DeviceGraph
    rp, ci = _hub_graph()
    n = rp.shape[0] - 1
    P = _tp(O.init_params(H, seed=2), dev)
    dts = ops.step_sizes(ops.time_grid(4, 0.5))
    x_small = torch.from_numpy(O.make_samples(n, 2, H, seed=5)).to(dev).reshape(2 * n, 3 + H)
    x_big = torch.from_numpy(O.make_samples(n, 5, H, seed=6)).to(dev).reshape(5 * n, 3 + H)
    g = DeviceGraph(rp, ci)
    stream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(graph, stream=stream):          # FIRST use of this handle
            S, I, R, sol = ops.forward(g, x_small, P, dts, "euler", None, want_sol)
    torch.cuda.synchronize()
    Sb, _, _, _ = ops.forward(g, x_big, P, dts, "euler", None, want_sol)      # larger batch, same handle, eager
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    g2 = DeviceGraph(rp, ci)
    Se, Ie, Re, sole = ops.forward(g2, x_small, P, dts, "euler", None, want_sol)
    assert torch.equal(S, Se) and torch.equal(I, Ie) and torch.equal(R, Re)
    if want_sol:
        rows = 2 * n
        assert torch.equal(sol[:, :3 * rows], sole[:, :3 * rows])              # (4th slabs: unwritten when a keep buffer is given)
        ka, kb = (t.gnode_keep.view(len(dts) + 1, 3, rows + 1, H) for t in (sol, sole))
        assert torch.equal(ka[:, 1], kb[:, 1]) and torch.equal(ka[:-1, 0, :rows], kb[:-1, 0, :rows])
        assert torch.equal(ka[1:-1, 2, :rows], kb[1:-1, 2, :rows])
    want = O.odeblock_forward_single(x_small.cpu().numpy().reshape(2, n, 3 + H), {k: v.cpu().numpy() for k, v in P.items()},
                                     rp, ci, 4, 0.5)
    assert np.max(np.abs(S.cpu().numpy() - want[0][..., 0])) <= 1e-5
    assert Sb.shape[1] == 5 * n


def test_backward_capture_and_two_streams(dev):
    """The adjoint backward on a hub graph inside a capture on first use; and one handle driven from two streams with
    separate workspaces gives the single-stream result."""
    import torch
    import gnode_oracle as O
    from gnode import ops
    from gnode.graph import DeviceGraph
    rp, ci = _hub_graph(n=900, m=7000, seed=8)
    n, H = rp.shape[0] - 1, 64
    P = _tp(O.init_params(H, seed=1), dev)
    dts = ops.step_sizes(ops.time_grid(3, 0.5))
    xs = [torch.from_numpy(O.make_samples(n, 2, H, seed=s)).to(dev).reshape(2 * n, 3 + H) for s in (1, 2)]
    g = DeviceGraph(rp, ci)
    # two streams, one handle
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [None, None]
    for rep in range(3):
        for i in (0, 1):
            with torch.cuda.stream(st[i]):
                outs[i] = ops.forward(g, xs[i], P, dts, "euler", None, True)
    torch.cuda.synchronize()
    for i in (0, 1):
        ref = ops.forward(g, xs[i], P, dts, "euler", None, True)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(outs[i][:3], ref[:3]))
        rows = xs[i].shape[0]
        assert torch.equal(outs[i][3][:, :3 * rows], ref[3][:, :3 * rows])      # (4th slabs: unwritten when a keep buffer is given)
        ka, kb = (t[3].gnode_keep.view(len(dts) + 1, 3, rows + 1, H) for t in (outs[i], ref))   # kept activations
        assert torch.equal(ka[:, 1], kb[:, 1]) and torch.equal(ka[:-1, 0, :rows], kb[:-1, 0, :rows])
        assert torch.equal(ka[1:-1, 2, :rows], kb[1:-1, 2, :rows])
    # backward captured on a fresh handle (first use of the backward entry point on it)
    g3 = DeviceGraph(rp, ci)
    S, I, R, sol = ops.forward(g3, xs[0], P, dts, "euler", None, True)
    gS, gI, gR = (torch.randn_like(S) for _ in range(3))
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        with torch.cuda.graph(graph, stream=stream):
            grads = ops.backward(g3, xs[0], P, dts, "euler", None, sol, gS, gI, gR)
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    ref = ops.backward(g, xs[0], P, dts, "euler", None, sol, gS, gI, gR)
    torch.cuda.synchronize()
    for k in grads:
        assert torch.equal(grads[k], ref[k]), k
