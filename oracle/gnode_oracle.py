"""CPU oracle for the GN-ODE hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement of the reference's algorithm for the one hot path this repo
accelerates.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product path
(``gn-ode-sir_amd/``) never does and fails loudly without its HIP library.

Pinning status (see DESIGN.md "Oracle"):
  * ``rhs_single`` / ``rhs_multi`` / ``odeblock_forward_*`` (encoder, read-out)
    and ``sir_coins`` are pinned against the *imported reference classes* run in
    the build container; the vectors are committed under ``tests/golden/`` with
    the generating script ``tests/golden/make_golden.py``.
  * ``euler_grid`` / ``rk4_38_grid`` restate torchdiffeq==0.2.2
    (requirements.txt:59), which is a third-party dependency absent from
    /root/reference and from this image: that boundary is PARITY UNPINNED by any
    reference artefact; only its call sites (ode_nn_ngraph_sim.py:168,
    ode_nn_ngraphs.py:137) anchor it.
  * ``sir_philox`` is the same state transition as ``sir_coins`` driven by a
    counter-based Philox4x32-10 coin source defined HERE (the reference draws
    from torch's CPU generator, ode_nn.py:65,70); it is the spec for the
    production Monte-Carlo kernel and is checked statistically against
    ``sir_coins``.
  * The reference's shipped label pickles (multi-graph-1/.../karate-*.pkl) and
    graph pickles (real_graphs/*.pkl) are NOT used: the only loaders that
    execute nothing (numpy.load(allow_pickle=False)) refuse them.

All citations are file:line into the reference repository.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

import contextlib

F32 = np.float32


@contextlib.contextmanager
def precision(dtype):
    """Run the restatement in another working precision (float64 = the exact-arithmetic
    yardstick used to measure the fp32 noise floor of long integrations)."""
    global F32
    old, F32 = F32, dtype
    try:
        yield
    finally:
        F32 = old


# --------------------------------------------------------------------------- graphs
def csr_from_edges(n, edges):
    """Adjacency as the reference builds it: ode_nn.py:413 ``nx.adjacency_matrix``
    of an undirected graph -> symmetric CSR, sorted columns, a self-loop is ONE
    diagonal entry, values ignored (ode_nn_ngraph_sim.py:70 uses .row/.col only).

    edges: iterable of (u, v) undirected pairs (duplicates collapse).
    Returns (rowptr int32 [n+1], col int32 [nnz]).
    """
    e = np.asarray(list(edges), dtype=np.int64).reshape(-1, 2)
    r = np.concatenate([e[:, 0], e[:, 1]])
    c = np.concatenate([e[:, 1], e[:, 0]])
    a = sp.coo_matrix((np.ones(r.shape[0], dtype=np.int8), (r, c)), shape=(n, n)).tocsr()
    a.sum_duplicates()
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32)


def csr_from_scipy(A):
    """Same contract for a scipy sparse adjacency (what ``create_graph`` returns)."""
    a = sp.csr_matrix(A)
    a.sum_duplicates()
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32)


def edge_table(edges):
    """Directed edge table of ``sir_torch`` (ode_nn.py:32-38): row 2i = (u,v),
    row 2i+1 = (v,u) in ``G.edges()`` order; a self-loop gives two equal rows."""
    e = np.asarray(list(edges), dtype=np.int64).reshape(-1, 2)
    t = np.empty((2 * e.shape[0], 2), dtype=np.int64)
    t[0::2, 0], t[0::2, 1] = e[:, 0], e[:, 1]
    t[1::2, 0], t[1::2, 1] = e[:, 1], e[:, 0]
    return t


def er_graph(n, m, seed=0):
    """Erdos-Renyi G(n, m) of SURVEY 8(d): m undirected edges, no self-loops, no
    duplicates, numpy default_rng(seed); returns (rowptr, col, undirected_edges)."""
    rng = np.random.default_rng(seed)
    have = np.empty(0, dtype=np.int64)
    while have.shape[0] < m:
        k = int((m - have.shape[0]) * 1.2) + 16
        u = rng.integers(0, n, k, dtype=np.int64)
        v = rng.integers(0, n, k, dtype=np.int64)
        ok = u != v
        lo, hi = np.minimum(u, v)[ok], np.maximum(u, v)[ok]
        have = np.unique(np.concatenate([have, lo * n + hi]))
        if have.shape[0] > m:
            have = np.sort(rng.permutation(have)[:m])
    e = np.stack([have // n, have % n], 1)
    rp, ci = csr_from_edges(n, e)
    return rp, ci, e


# --------------------------------------------------------------------------- RHS
def _sigmoid(x):
    x = x.astype(F32, copy=False)
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


def _spmm_blockdiag(rowptr, col, n, Z):
    """AI[r] = sum_{c in adj(r mod n)} Z[(r - r mod n) + c]  -- the unweighted COO
    gather + scatter_add_ of ode_nn_ngraph_sim.py:68-73 over the implicit
    block-diagonal replication of A; ascending-column order, fp32 accumulate."""
    rows = Z.shape[0]
    assert rows % n == 0
    a = sp.csr_matrix((np.ones(col.shape[0], dtype=F32), col, rowptr), shape=(n, n))
    out = np.empty_like(Z)
    for b in range(rows // n):
        out[b * n:(b + 1) * n] = a @ Z[b * n:(b + 1) * n]
    return out


def rhs_single(x, W, b, rowptr, col, n):
    """ODEfunc.forward of the single-graph script, ode_nn_ngraph_sim.py:58-96.

    x [4*B*n, H] fp32: slabs S | I | R | beta-gamma (col 0 = beta, col 1 = gamma,
    :59-60).  Returns dx of the same shape; the 4th slab's derivative is 0 (:96).
    R' = sigmoid(W R + b) is computed by the reference and never used (:62-66).
    """
    x = np.asarray(x, dtype=F32)
    q = x.shape[0] // 4
    beta, gamma = x[3 * q:, 0:1], x[3 * q:, 1:2]
    Z = _sigmoid(x[:2 * q] @ W.T.astype(F32) + b.astype(F32))          # :62-63 (S and I only)
    ZS, ZI = Z[:q], Z[q:2 * q]
    AI = _spmm_blockdiag(rowptr, col, n, ZI)                            # :68-73
    dS = -beta * (AI * ZS)                                              # :75
    dI = -dS - gamma * ZI                                               # :76
    dR = gamma * ZI                                                     # :77
    return np.concatenate([dS, dI, dR, np.zeros_like(x[3 * q:])]).astype(F32)


def concat_csr(graphs, picks):
    """Block-diagonal CSR of ``graphs[p] for p in picks`` (ode_nn_ngraphs.py:65-69).
    graphs: list of (rowptr, col).  Returns (rowptr, col, offsets[len(picks)+1])."""
    rps, cis, off = [np.zeros(1, dtype=np.int64)], [], [0]
    nnz = 0
    for p in picks:
        rp, ci = graphs[p]
        n = rp.shape[0] - 1
        rps.append(rp[1:].astype(np.int64) + nnz)
        cis.append(ci.astype(np.int64) + off[-1])
        nnz += ci.shape[0]
        off.append(off[-1] + n)
    return (np.concatenate(rps).astype(np.int32), np.concatenate(cis).astype(np.int32),
            np.asarray(off, dtype=np.int64))


def picks_from_marker(marker):
    """ode_nn_ngraphs.py:65-67: every non-zero entry of x[3,:,2] names one sample's
    graph as ``graph_idx + 1`` in node order."""
    nz = np.nonzero(np.asarray(marker))[0]
    return [int(marker[i]) - 1 for i in nz]


def rhs_multi(x, W, b, graphs):
    """ODEfunc.forward of the multi-graph script, ode_nn_ngraphs.py:54-83.
    x [4, sumN, H]; beta, gamma, marker = x[3,:,0..2] (:55)."""
    x = np.asarray(x, dtype=F32)
    tot, H = x.shape[1], x.shape[2]
    rp, ci, off = concat_csr(graphs, picks_from_marker(x[3, :, 2]))
    assert off[-1] == tot, "marker/graph sizes do not tile the batch"
    flat = np.concatenate([x[0], x[1], x[2], x[3]])
    d = rhs_single(flat, W, b, rp, ci, tot)
    return d.reshape(4, tot, H)


# --------------------------------------------------------------------------- integrators
def time_grid(maxTime, deltaT):
    """ode_nn_ngraph_sim.py:110: float64 ``np.arange(0, maxTime, deltaT)``."""
    return np.arange(0, maxTime, deltaT)


def step_sizes(t):
    """dt_k = t[k+1]-t[k] in float64, applied to an fp32 state as an fp32 scalar
    (a 0-dim float64 tensor times an fp32 tensor stays fp32 in torch)."""
    t = np.asarray(t, dtype=np.float64)
    return (t[1:] - t[:-1]).astype(F32)


def euler_grid(f, y0, t):
    """torchdiffeq 0.2.2 fixed-grid Euler restated: output at
    every grid point, sol[0] = y0, y_{k+1} = y_k + dt_k * f(t_k, y_k).
    PARITY UNPINNED (third-party, absent)."""
    sol = [np.asarray(y0, dtype=F32)]
    for k, dt in enumerate(step_sizes(t)):
        sol.append((sol[-1] + dt * f(t[k], sol[-1])).astype(F32))
    return np.stack(sol)


def rk4_38_grid(f, y0, t):
    """torchdiffeq 0.2.2 'rk4' = the 3/8-rule.  PARITY UNPINNED."""
    sol = [np.asarray(y0, dtype=F32)]
    third = F32(1.0 / 3.0)
    for k, dt in enumerate(step_sizes(t)):
        y, t0 = sol[-1], t[k]
        k1 = f(t0, y)
        k2 = f(t0 + dt * third, (y + dt * k1 * third).astype(F32))
        k3 = f(t0 + dt * 2 * third, (y + dt * (k2 - k1 * third)).astype(F32))
        k4 = f(t0 + dt, (y + dt * (k1 - k2 + k3)).astype(F32))
        sol.append((y + dt * (k1 + F32(3) * (k2 + k3) + k4) * F32(0.125)).astype(F32))
    return np.stack(sol)


# --------------------------------------------------------------------------- ODEBlock
def encode(s, w1, b1):
    """relu(Linear(1,H)) shared by S0, I0, R0: ode_nn_ngraph_sim.py:151-156."""
    return np.maximum(s.astype(F32)[:, None] * w1.astype(F32)[None, :, 0] + b1.astype(F32), F32(0))


def readout(Y, w3, b3, w2, b2):
    """Linear(4,1)(relu(Linear(H,4)(Y))): ode_nn_ngraph_sim.py:172-182."""
    p = np.maximum(Y.astype(F32) @ w3.T.astype(F32) + b3.astype(F32), F32(0))
    return (p @ w2.T.astype(F32) + b2.astype(F32)).astype(F32)


def softmax3(qS, qI, qR):
    """Softmax over the three compartments, ode_nn_ngraph_sim.py:184-187."""
    q = np.concatenate([qS, qI, qR], -1).astype(F32)
    e = np.exp(q - q.max(-1, keepdims=True), dtype=F32)
    p = (e / e.sum(-1, keepdims=True)).astype(F32)
    return p[..., 0:1], p[..., 1:2], p[..., 2:3]


def odeblock_forward_single(x, P, rowptr, col, maxTime, deltaT, method="euler", return_sol=False):
    """ODEBlock.forward, single-graph script: ode_nn_ngraph_sim.py:148-188.

    x [B, n, 3+H] fp32.  P: dict of fp32 arrays with the reference state_dict
    names: 'odefunc.linear.weight' [H,H], 'odefunc.linear.bias' [H],
    'linearS1.weight' [H,1], 'linearS1.bias' [H], 'linear3.weight' [4,H],
    'linear3.bias' [4], 'linearS2.weight' [1,4], 'linearS2.bias' [1].
    Returns S, I, R each [G, B*n, 1] fp32.
    """
    x = np.asarray(x, dtype=F32)
    n = x.shape[1]
    x2 = x.reshape(-1, x.shape[2])                                      # :149
    enc = lambda v: encode(v, P["linearS1.weight"], P["linearS1.bias"])
    y0 = np.concatenate([enc(x2[:, 0]), enc(x2[:, 1]), enc(x2[:, 2]), x2[:, 3:]]).astype(F32)
    f = lambda t, y: rhs_single(y, P["odefunc.linear.weight"], P["odefunc.linear.bias"], rowptr, col, n)
    grid = time_grid(maxTime, deltaT)
    sol = (euler_grid if method == "euler" else rk4_38_grid)(f, y0, grid)   # :168
    q = sol.shape[1] // 4
    ro = lambda Y: readout(Y, P["linear3.weight"], P["linear3.bias"], P["linearS2.weight"], P["linearS2.bias"])
    out = softmax3(ro(sol[:, :q]), ro(sol[:, q:2 * q]), ro(sol[:, 2 * q:3 * q]))
    return out + (sol,) if return_sol else out


def odeblock_forward_multi(x, P, graphs, maxTime, deltaT, method="euler"):
    """ODEBlock.forward, multi-graph script: ode_nn_ngraphs.py:124-152.
    x [sumN, 3+H] (samples of different graphs concatenated along nodes, :179-196)."""
    x = np.asarray(x, dtype=F32)
    rp, ci, off = concat_csr(graphs, picks_from_marker(x[:, 3 + 2]))
    assert off[-1] == x.shape[0]
    return odeblock_forward_single(x[None], P, rp, ci, maxTime, deltaT, method)


def get_sir_t_nodes(x, maxTime, deltaT):
    """ode_nn.py:249-261 with count=False: out[i] = x[int(i/deltaT)], i < maxTime."""
    idx = np.asarray([int(i / deltaT) for i in range(int(maxTime))])
    return np.asarray(x)[idx]


def l1_loss(S, I, R, y, maxTime, deltaT):
    """Loss assembly ode_nn_ngraph_sim.py:234: mean |pred - y| over [BN, T-1, 3]
    (t = 0 excluded); y [B, n, T, 3] float64 labels -> float64 loss."""
    sub = lambda a: get_sir_t_nodes(a[..., 0], maxTime, deltaT)            # [T, BN]
    pred = np.stack([sub(S), sub(I), sub(R)], -1).transpose(1, 0, 2)       # [BN, T, 3]
    yy = np.asarray(y, dtype=np.float64).reshape(-1, y.shape[-2], y.shape[-1])
    return float(np.mean(np.abs(pred[:, 1:, :].astype(np.float64) - yy[:, 1:, :])))


# --------------------------------------------------------------------------- sir_torch
def sir_coins(n, table, seed_set, beta, gamma, sims, T, coins):
    """``sir_torch`` (ode_nn.py:30-88) driven by a RECORDED coin stream.

    table: directed edge table [2E,2] (``edge_table``).  coins: 1-D float64 array =
    the concatenation of every ``torch.rand`` the reference drew, in call order
    (per step: one call sized #targets :65, one sized #infected :70).
    Returns (S, I, R) float64 counts [1, T, n] exactly as the reference returns
    them (row t=0 of S and I is ASSIGNED each sim :55-56, not accumulated; R row 0
    stays 0), plus the number of coins consumed and the per-(sim,step) state
    trace uint8 [sims, T, n] (0=S, 1=I, 2=R) for bit-exact state parity.
    """
    src, dst = table[:, 0], table[:, 1]
    S_acc = np.zeros((1, T, n)); I_acc = np.zeros((1, T, n)); R_acc = np.zeros((1, T, n))
    trace = np.zeros((sims, T, n), dtype=np.uint8)
    pos = 0
    for s in range(sims):
        I = np.zeros(n, dtype=bool); S = np.ones(n, dtype=bool); R = np.zeros(n, dtype=bool)
        I[list(seed_set)] = True; S[list(seed_set)] = False
        I_acc[0, 0] = I; S_acc[0, 0] = S
        trace[s, 0] = I.astype(np.uint8)
        for it in range(1, T):
            idx_I = np.nonzero(I)[0]                                    # :60
            act = I[src] & S[dst]                                       # :61-62 (dst column survives)
            targets = dst[act]                                          # table order, with multiplicity
            c1 = coins[pos:pos + targets.shape[0]]; pos += targets.shape[0]
            new_inf = targets[c1 < beta]                                # :65-67
            c2 = coins[pos:pos + idx_I.shape[0]]; pos += idx_I.shape[0]
            new_rec = idx_I[c2 < gamma]                                 # :70-72
            R[new_rec] = True                                           # :73
            I[new_inf] = True; I[new_rec] = False; S[new_inf] = False   # :76-78
            I_acc[0, it] += I; S_acc[0, it] += S; R_acc[0, it] += R     # :80-82
            trace[s, it] = I.astype(np.uint8) + 2 * R.astype(np.uint8)
    return S_acc, I_acc, R_acc, pos, trace


M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10_block(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al. 2011), vectorised over numpy uint64 holders: the four output words as uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64); c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64); c3 = np.asarray(c3, dtype=np.uint64)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint64(k0); k1 = np.uint64(k1)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & mask, lo1, (hi0 ^ c3 ^ k1) & mask, lo0
        k0 = (k0 + np.uint64(W0)) & mask
        k1 = (k1 + np.uint64(W1)) & mask
    return tuple(np.asarray(c).astype(np.uint32) for c in (c0, c1, c2, c3))


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Word 0 of the Philox4x32-10 output block as uint32 array."""
    return philox4x32_10_block(c0, c1, c2, c3, k0, k1)[0]


def philox_coin(pos, it, sim, kind, k0, k1):
    """The coin of item `pos` (a CSR position for kind 0 = infection, a node id for kind 1 = recovery): FOUR consecutive
    items share one Philox block -- word (pos & 3) of philox(ctr = (pos >> 2, it, sim, kind), key).  (One block per coin
    spent all ten rounds on one word of four; the kernels' lanes walk four consecutive items per block instead.)"""
    pos = np.asarray(pos, dtype=np.uint64)
    w = philox4x32_10_block(pos >> np.uint64(2), it, sim, kind, k0, k1)
    sel = (pos & np.uint64(3)).astype(np.int64)
    return np.choose(sel, [np.broadcast_to(x, sel.shape) for x in w]).astype(np.uint64) if sel.size else np.zeros(0, dtype=np.uint64)


def coin_threshold(p):
    """A coin fires iff philox_word < floor(p * 2^32) (64-bit compare, p in [0,1])."""
    return np.uint64(min(max(int(np.floor(float(p) * 4294967296.0)), 0), 1 << 32))


def sir_philox(n, rowptr, col, seed_set, beta, gamma, sims, T, rng_seed, sim_offset=0):
    """Production-mode Monte-Carlo SIR: the state transition of ode_nn.py:58-82
    with counter-based coins instead of torch's CPU stream.

    Infection coin of directed CSR edge e (row u -> col[e]) in sim s at step it:
        word (e & 3) of philox(ctr=(e >> 2, it, s, 0), key=(seed_lo, seed_hi)) < thr(beta)
    Recovery coin of node u:  word (u & 3) of philox(ctr=(u >> 2, it, s, 1), key) < thr(gamma)      (philox_coin).
    Both are decided on the pre-step state.  s = sim_offset + local index, so a
    sims-sharded run over several GPUs reproduces the single-GPU counts exactly.
    Returns uint32 counts [3, T, n] (S, I, R) with the reference's row-0 quirk
    (Q3): S/I row 0 hold the initial state ONCE, R row 0 is 0.
    """
    src = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
    dst = col.astype(np.int64)
    eid = np.arange(col.shape[0], dtype=np.uint64)
    k0, k1 = np.uint64(rng_seed & 0xFFFFFFFF), np.uint64((rng_seed >> 32) & 0xFFFFFFFF)
    tb, tg = coin_threshold(beta), coin_threshold(gamma)
    cnt = np.zeros((3, T, n), dtype=np.uint32)
    for s in range(sim_offset, sim_offset + sims):
        I = np.zeros(n, dtype=bool); S = np.ones(n, dtype=bool); R = np.zeros(n, dtype=bool)
        I[list(seed_set)] = True; S[list(seed_set)] = False
        cnt[0, 0] = S; cnt[1, 0] = I
        for it in range(1, T):
            act = np.nonzero(I[src] & S[dst])[0]
            w = philox_coin(eid[act], it, s, 0, k0, k1)
            new_inf = dst[act[w < tb]]
            idx_I = np.nonzero(I)[0]
            w2 = philox_coin(idx_I.astype(np.uint64), it, s, 1, k0, k1)
            new_rec = idx_I[w2 < tg]
            R[new_rec] = True
            I[new_inf] = True; I[new_rec] = False; S[new_inf] = False
            cnt[0, it] += S; cnt[1, it] += I; cnt[2, it] += R
    return cnt


# --------------------------------------------------------------------------- reference-op-sequence port (cpu_baseline)
def usable_cores():
    """Host cores this process may actually use: the cgroup CPU quota when there is one (the GPU boxes
    give 16 of 256 visible cores; 256 OpenMP threads on a 16-core quota run 10x SLOWER than 16), else
    the affinity mask."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except Exception:
            pass
    return n



def torch_port_forward(x, P, rowptr, col, maxTime, deltaT, n_steps=None, threads=None):
    """The reference's op sequence stated in PyTorch-CPU for the ``cpu_baseline``
    leg of bench.py (SURVEY 8d "Baseline A"): Linear + sigmoid, repeat-expanded
    int64 index, gather, scatter_add_, elementwise, cat -- one Euler step per
    grid interval, then the read-out head (ode_nn_ngraph_sim.py:58-96, 148-188).
    ``n_steps`` bounds the number of Euler steps (a bounded sample of the same
    workload).  Returns (S, I, R, seconds_in_euler_loop, steps_done)."""
    import time
    import torch
    prev_threads = torch.get_num_threads()
    if threads:
        torch.set_num_threads(threads)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=F32))
    W, b = tt(P["odefunc.linear.weight"]), tt(P["odefunc.linear.bias"])
    w1, b1 = tt(P["linearS1.weight"]), tt(P["linearS1.bias"])
    w3, b3 = tt(P["linear3.weight"]), tt(P["linear3.bias"])
    w2, b2 = tt(P["linearS2.weight"]), tt(P["linearS2.bias"])
    xt = tt(x)
    B, n = xt.shape[0], xt.shape[1]
    x2 = xt.view(-1, xt.shape[2])
    enc = lambda v: torch.relu(torch.nn.functional.linear(v.unsqueeze(-1), w1, b1))
    y = torch.cat((enc(x2[:, 0]), enc(x2[:, 1]), enc(x2[:, 2]), x2[:, 3:]))
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
    cols = col.astype(np.int64)
    dts = step_sizes(time_grid(maxTime, deltaT))
    if n_steps is not None:
        dts = dts[:n_steps]
    sol = [y]
    t0 = time.perf_counter()
    for dt in dts:
        q = y.shape[0] // 4
        sir = torch.sigmoid(torch.nn.functional.linear(y[:3 * q], W, b))
        S, I = sir[:q], sir[q:2 * q]
        beta, gamma = y[3 * q:, 0], y[3 * q:, 1]
        # the reference rebuilds the block-diagonal COO index on every call (:68-71)
        idx = torch.from_numpy(np.vstack([np.concatenate([rows + k * n for k in range(B)]),
                                          np.concatenate([cols + k * n for k in range(B)])]))
        AI = torch.zeros(I.size()).scatter_add_(0, idx[0].unsqueeze(1).repeat(1, I.size(1)), I[idx[1]])
        dS = -beta.unsqueeze(-1) * (AI * S)
        dI = -dS - gamma.unsqueeze(-1) * I
        dR = gamma.unsqueeze(-1) * I
        y = y + float(dt) * torch.cat((dS, dI, dR, torch.zeros_like(y[3 * q:])))
        sol.append(y)
    secs = time.perf_counter() - t0
    sol = torch.stack(sol)
    q = sol.shape[1] // 4
    ro = lambda Y: torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(Y, w3, b3)), w2, b2)
    out = torch.softmax(torch.cat((ro(sol[:, :q]), ro(sol[:, q:2 * q]), ro(sol[:, 2 * q:3 * q])), -1), dim=2)
    S, I, R = out.chunk(3, dim=-1)
    torch.set_num_threads(prev_threads)       # a 256-thread pool makes every later tiny CPU op cost milliseconds
    return S.numpy(), I.numpy(), R.numpy(), secs, len(dts)


# --------------------------------------------------------------------------- helpers shared by tests/bench
PARAM_SHAPES = lambda H: {
    "odefunc.linear.weight": (H, H), "odefunc.linear.bias": (H,),
    "linearS1.weight": (H, 1), "linearS1.bias": (H,),
    "linear3.weight": (4, H), "linear3.bias": (4,),
    "linearS2.weight": (1, 4), "linearS2.bias": (1,),
}


def init_params(H, seed=0):
    """Default ``nn.Linear`` init (U(-1/sqrt(fan_in), 1/sqrt(fan_in)); the
    reference never calls its init_weights, ode_nn_ngraph_sim.py:52,137) from a
    numpy generator so that it is reproducible without torch."""
    rng = np.random.default_rng(seed)
    P = {}
    for name, shp in PARAM_SHAPES(H).items():
        fan_in = shp[1] if len(shp) == 2 else PARAM_SHAPES(H)[name.replace("bias", "weight")][1]
        k = 1.0 / np.sqrt(fan_in)
        P[name] = rng.uniform(-k, k, size=shp).astype(F32)
    return P


def make_samples(n, B, H, seed=0, n_seeds=2):
    """x [B, n, 3+H] as ode_nn_ngraph_sim.py:382-390 builds it: S0|I0|R0 columns
    then an H-wide slab whose col 0 = beta, col 1 = gamma (U(0.1,0.5),
    monitorer-sim.py:116-119)."""
    rng = np.random.default_rng(seed)
    x = np.zeros((B, n, 3 + H), dtype=F32)
    for b in range(B):
        seeds = rng.choice(n, size=min(n_seeds, n), replace=False)
        x[b, :, 0] = 1.0
        x[b, seeds, 0] = 0.0
        x[b, seeds, 1] = 1.0
        x[b, :, 3] = rng.uniform(0.1, 0.5)
        x[b, :, 4] = rng.uniform(0.1, 0.5)
    return x


# --------------------------------------------------------------------------- adjoint backward (A7)
def adjoint_grads_torch(x, P, rowptr, col, maxTime, deltaT, gS, gI, gR, out_rows=None, dtype="float32"):
    """Parameter gradients of  L = <gS,S> + <gI,I> + <gR,R>  as torchdiffeq 0.2.2's
    ``odeint_adjoint(..., method='euler')`` produces them -- PARITY
    UNPINNED (third-party, absent); restated here with torch autograd supplying every
    vector-Jacobian product, exactly like torchdiffeq's augmented dynamics do:

      forward under no_grad, ``sol`` saved;  a <- dL/dsol[G-1];  for i = G-1 .. 1:
          f = func(t_i, sol[i]);  (vjp_y, vjp_theta) = grad(f, (y, theta), a)
          a <- a + dt_i * vjp_y + dL/dsol[i-1];   g_theta <- g_theta + dt_i * vjp_theta
      (one Euler step of the augmented system from t_i to t_{i-1}: the Jacobians are taken
      at the RIGHT endpoint y_i, then y is reset to the stored sol[i-1]);  a_0 flows into
      the encoder.  The head (Linear(H,4), relu, Linear(4,1), softmax) is ordinary autograd
      on ``sol``.   x [B,n,3+H]; gS,gI,gR [n_out, B*n].  Returns {state_dict key: grad}.
    """
    import torch
    dt_t = getattr(torch, dtype)
    tt = lambda a: torch.tensor(np.asarray(a), dtype=dt_t)
    Pt = {k: tt(v).requires_grad_(True) for k, v in P.items()}
    xt = tt(x)
    B, n = xt.shape[0], xt.shape[1]
    x2 = xt.reshape(-1, xt.shape[2])
    rows = x2.shape[0]
    src = torch.from_numpy(np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr)))
    dst = torch.from_numpy(col.astype(np.int64))
    ridx = torch.cat([src + b * n for b in range(B)])
    cidx = torch.cat([dst + b * n for b in range(B)])

    def func(y, W, b):
        q = y.shape[0] // 4
        Z = torch.sigmoid(torch.nn.functional.linear(y[:2 * q], W, b))
        ZS, ZI = Z[:q], Z[q:]
        beta, gamma = y[3 * q:, 0:1], y[3 * q:, 1:2]
        AI = torch.zeros_like(ZI).index_add(0, ridx, ZI[cidx])
        dS = -beta * (AI * ZS)
        dI = -dS - gamma * ZI
        dR = gamma * ZI
        return torch.cat((dS, dI, dR, torch.zeros_like(y[3 * q:])))

    enc = lambda v: torch.relu(torch.nn.functional.linear(v.unsqueeze(-1), Pt["linearS1.weight"], Pt["linearS1.bias"]))
    y0 = torch.cat((enc(x2[:, 0]), enc(x2[:, 1]), enc(x2[:, 2]), x2[:, 3:]))
    grid = time_grid(maxTime, deltaT)
    dts = step_sizes(grid)
    W, b = Pt["odefunc.linear.weight"], Pt["odefunc.linear.bias"]
    with torch.no_grad():
        sol = [y0.detach()]
        for dt in dts:
            sol.append(sol[-1] + float(dt) * func(sol[-1], W, b))
        sol = torch.stack(sol)
    sol_leaf = sol.clone().requires_grad_(True)
    ro = lambda Y: torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(Y, Pt["linear3.weight"], Pt["linear3.bias"])),
                                              Pt["linearS2.weight"], Pt["linearS2.bias"])
    out = torch.softmax(torch.cat((ro(sol_leaf[:, :rows]), ro(sol_leaf[:, rows:2 * rows]), ro(sol_leaf[:, 2 * rows:3 * rows])), -1), dim=2)
    idx = torch.arange(sol.shape[0]) if out_rows is None else torch.as_tensor(np.asarray(out_rows), dtype=torch.int64)
    L = (out[idx, :, 0] * tt(gS)).sum() + (out[idx, :, 1] * tt(gI)).sum() + (out[idx, :, 2] * tt(gR)).sum()
    head = ["linear3.weight", "linear3.bias", "linearS2.weight", "linearS2.bias"]
    gr = torch.autograd.grad(L, [sol_leaf] + [Pt[k] for k in head])
    gsol = gr[0]
    grads = {k: g for k, g in zip(head, gr[1:])}
    a = gsol[-1].clone()
    gW, gb = torch.zeros_like(W), torch.zeros_like(b)
    for i in range(sol.shape[0] - 1, 0, -1):
        yi = sol[i].clone().requires_grad_(True)
        f = func(yi, W, b)
        vy, vW, vb = torch.autograd.grad(f, (yi, W, b), a)
        dt = float(dts[i - 1])
        a = a + dt * vy + gsol[i - 1]
        gW += dt * vW
        gb += dt * vb
    grads["odefunc.linear.weight"], grads["odefunc.linear.bias"] = gW, gb
    ge = torch.autograd.grad(y0, [Pt["linearS1.weight"], Pt["linearS1.bias"]], a)
    grads["linearS1.weight"], grads["linearS1.bias"] = ge
    return {k: v.detach().numpy() for k, v in grads.items()}


def chung_lu_graph(n, m, exponent=0.8, seed=0):
    """Skewed-degree test graph (Chung-Lu: endpoints drawn with probability ~ (rank+1)^-exponent):
    m distinct undirected edges, no self-loops; hubs of degree ~ n/10 like wiki-vote / epinions.
    Returns (rowptr, col, undirected_edges)."""
    rng = np.random.default_rng(seed)
    w = (np.arange(n) + 1.0) ** (-exponent)
    p = w / w.sum()
    have = np.empty(0, dtype=np.int64)
    while have.shape[0] < m:
        k = int((m - have.shape[0]) * 1.5) + 64
        u = rng.choice(n, size=k, p=p)
        v = rng.choice(n, size=k, p=p)
        ok = u != v
        lo, hi = np.minimum(u, v)[ok], np.maximum(u, v)[ok]
        have = np.unique(np.concatenate([have, lo.astype(np.int64) * n + hi]))
        if have.shape[0] > m:
            have = np.sort(rng.permutation(have)[:m])
    e = np.stack([have // n, have % n], 1)
    rp, ci = csr_from_edges(n, e)
    return rp, ci, e


# --------------------------------------------------------------------------- DMP baseline
def dmp_reverse_index(rowptr, col):
    """rev[e] = CSR position of the reverse of directed edge e = (row -> col[e]), or nnz when the reverse is absent
    (reference dmp.py:35-50 `cave_index`, there via a networkx DiGraph lookup)."""
    n, nnz = len(rowptr) - 1, len(col)
    src = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
    pos = {(int(s), int(t)): e for e, (s, t) in enumerate(zip(src, col))}
    return np.array([pos.get((int(t), int(s)), nnz) for s, t in zip(src, col)], dtype=np.int64)


def dmp_sir(rowptr, col, weights, gamma, seeds, maxTime):
    """Dynamic message passing for SIR, reference dmp.py:74-170 (`DMP_SIR.run`), restated in float32 numpy with the
    reference's operation order.  Directed edges are the CSR positions in row-major order (what
    `sp.coo_matrix(weight_adj)` yields, dmp.py:67-72): src = row, tar = col, `weights` [nnz], `gamma` [n].
    `scatter(..., reduce='mul')` (torch_scatter, ABSENT here: parity unpinned) multiplies in ascending edge order.
    Returns float32 [maxTime, n, 3] = (Ps, Pi, Pr) per step, step 0 = the initial condition."""
    f = np.float32
    n, E = len(rowptr) - 1, len(col)
    src = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
    tar = np.asarray(col, dtype=np.int64)
    cave = dmp_reverse_index(rowptr, col)
    w = np.asarray(weights, dtype=f)
    g_node = np.asarray(gamma, dtype=f)
    g_edge = g_node[src]

    def scatter_mul(vals, index, size):
        out = np.ones(size, dtype=f)
        for e in range(len(vals)):                      # ascending edge order, one float32 product at a time
            out[index[e]] = f(out[index[e]] * vals[e])
        return out

    def mulmul(theta):
        P = scatter_mul(theta, tar, n)[src]
        cav = scatter_mul(theta, cave, E + 1)[:E]
        return (P / cav).astype(f)

    seedv = np.zeros(n, dtype=f); seedv[list(seeds)] = 1
    Ps0, Pi0, Pr0 = (f(1) - seedv).astype(f), seedv.copy(), np.zeros(n, dtype=f)
    Ps_i0 = Ps0[src]
    Phi = (f(1) - Ps_i0).astype(f)
    theta = ((np.ones(E, dtype=f) - w * Phi).astype(f) + f(1e-10)).astype(f)          # dmp.py:117
    Ps_prev = Ps_i0
    Ps_e = (Ps_i0 * mulmul(theta)).astype(f)
    Phi = ((f(1) - w) * (f(1) - g_edge) * Phi - (Ps_e - Ps_prev)).astype(f)
    Ps_t = (Ps0 * scatter_mul(theta, tar, n)).astype(f)
    Pr_t = (Pr0 + g_node * Pi0).astype(f)
    Pi_t = (f(1) - Ps_t - Pr_t).astype(f)
    out = [np.stack([Ps0, Pi0, Pr0], 1), np.stack([Ps_t, Pi_t, Pr_t], 1)]
    for _ in range(maxTime - 2):
        theta = (theta - w * Phi).astype(f)
        new_Ps = (Ps_i0 * mulmul(theta)).astype(f)
        Ps_prev, Ps_e = Ps_e, new_Ps
        Phi = ((f(1) - w) * (f(1) - g_edge) * Phi - (Ps_e - Ps_prev)).astype(f)
        Ps_t = (Ps0 * scatter_mul(theta, tar, n)).astype(f)
        Pr_t = (Pr_t + g_node * Pi_t).astype(f)
        Pi_t = (f(1) - Ps_t - Pr_t).astype(f)
        out.append(np.stack([Ps_t, Pi_t, Pr_t], 1))
    return np.stack(out, 0).astype(f)


# --------------------------------------------------------------------------- mean-field baseline
def meanfield_rk(rowptr, col, seeds, beta, gamma, deltaT, maxTime):
    """`runge_kutta_order4` of the reference (ode_nn.py:222-233) with its RHS `sir` (:214-220): scipy's LSODA (default
    tolerances) on t = arange(0, maxTime, deltaT), rows int(i/deltaT) kept.  The only restatement is the sparse A I
    instead of the dense np.dot.  Returns (I, S, R), float64 [maxTime, n]."""
    import scipy.sparse as sp
    from scipy.integrate import odeint
    n = len(rowptr) - 1
    A = sp.csr_matrix((np.ones(len(col)), np.asarray(col), np.asarray(rowptr)), shape=(n, n))
    gam = gamma * np.ones(n)

    def rhs(x, t):
        S, I = x[:n], x[n:2 * n]
        AI = A @ I
        dS = -beta * AI * S
        return np.hstack([dS, -dS - gam * I, gam * I])

    y0 = np.zeros(3 * n); y0[n + np.asarray(list(seeds), dtype=np.int64)] = 1.0; y0[:n] = 1.0 - y0[n:2 * n]
    sol = odeint(rhs, y0, np.arange(0, maxTime, deltaT))
    rows = [int(i / deltaT) for i in range(int(maxTime))]
    return sol[rows, n:2 * n], sol[rows, :n], sol[rows, 2 * n:]
