#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE classes.

Runs only in the build container (it needs /root/reference); the GPU box never
sees the reference, only the .npz files this script wrote.  What is executed
from the reference, unchanged: ``ODEfunc.forward`` / ``ODEBlock.forward`` of
ode_nn_ngraph_sim.py and ode_nn_ngraphs.py, ``sir_torch`` and
``get_sir_t_nodes_torch`` of ode_nn.py.

Two third-party modules the reference imports are absent from this image and
stay absent:
  * ``ndlib`` (ode_nn.py:17-20) -- never called on this path; an empty module
    object satisfies the import statement.
  * ``torchdiffeq`` (ode_nn.py:15) -- ``odeint_adjoint`` IS called
    (ode_nn_ngraph_sim.py:168).  The callable handed to the reference here is
    this repo's restatement of on-grid fixed-step Euler (oracle.euler_grid's
    loop in torch), so the full-forward vectors pin everything EXCEPT the
    integrator, which stays "parity unpinned" (see DESIGN.md).
``sir_torch`` hard-codes ``.cuda()`` (ode_nn.py:41-50); with no GPU here
``Tensor.cuda`` is made the identity for the duration of that call, and
``torch.rand`` is wrapped to RECORD the coins the reference draws.

No pickle shipped inside the reference is loaded: graphs come from networkx
generators / numpy RNG, labels are produced by the calls above.
"""
import os
import sys
import types
import zlib

import numpy as np
import networkx as nx
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
REF = "/root/reference"


def _install_import_shims():
    for name in ["ndlib", "ndlib.models", "ndlib.models.ModelConfig", "ndlib.models.CompositeModel",
                 "ndlib.models.compartments", "ndlib.models.epidemics"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["ndlib.models.epidemics"].SIRModel = None
    td = types.ModuleType("torchdiffeq")

    def odeint_adjoint(func, y0, t, method="euler", **kw):
        assert method == "euler"
        sol = [y0]
        for k in range(t.shape[0] - 1):
            sol.append(sol[-1] + (t[k + 1] - t[k]) * func(t[k], sol[-1]))
        return torch.stack(sol)

    td.odeint_adjoint = odeint_adjoint
    td.odeint = odeint_adjoint
    sys.modules["torchdiffeq"] = td


def _graphs():
    g = {}
    g["karate"] = nx.karate_club_graph()
    rng = np.random.default_rng(7)
    gl = nx.gnm_random_graph(40, 90, seed=3)
    for u in rng.choice(40, 5, replace=False):          # self-loops: Q6 (diag counts once in A)
        gl.add_edge(int(u), int(u))
    gl = gl.subgraph(max(nx.connected_components(gl), key=len)).copy()
    g["loops40"] = nx.convert_node_labels_to_integers(gl, ordering="sorted")
    ge = nx.gnm_random_graph(200, 700, seed=11)
    ge = ge.subgraph(max(nx.connected_components(ge), key=len)).copy()
    g["er200"] = nx.convert_node_labels_to_integers(ge, ordering="sorted")
    return g


def _model_single(mod, A, H, maxTime, deltaT, seed):
    torch.set_default_dtype(torch.float32)
    torch.manual_seed(seed)
    dev = torch.device("cpu")
    f = mod.ODEfunc(A, 0.2, 0.1, H, dev)
    m = mod.ODEBlock(maxTime, deltaT, A.shape[0], [0], H, f, dev)
    return f, m


def _params(m):
    keep = ["odefunc.linear.weight", "odefunc.linear.bias", "linearS1.weight", "linearS1.bias",
            "linear3.weight", "linear3.bias", "linearS2.weight", "linearS2.bias"]
    sd = m.state_dict()
    return {"P:" + k: sd[k].detach().numpy().astype(np.float32) for k in keep}


def main():
    _install_import_shims()
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir("/tmp")
    import ode_nn_ngraph_sim as single
    import ode_nn_ngraphs as multi
    import ode_nn as helpers
    os.chdir(cwd)
    import gnode_oracle as O

    graphs = _graphs()
    out = {}

    # ---- A1: ODEfunc.forward (single graph), reference executed unchanged
    for gname, B, H in [("karate", 1, 64), ("karate", 2, 8), ("loops40", 3, 64), ("er200", 2, 16), ("er200", 1, 64)]:
        G = graphs[gname]
        A = nx.adjacency_matrix(G)
        n = A.shape[0]
        f, m = _model_single(single, A, H, 4, 0.5, seed=zlib.crc32(f"{gname}-{B}-{H}".encode()) % 1000)
        rng = np.random.default_rng(B * 100 + H)
        x = rng.uniform(0, 1.5, size=(4 * B * n, H)).astype(np.float32)
        x[3 * B * n:, 0] = np.repeat(rng.uniform(0.1, 0.5, B), n)
        x[3 * B * n:, 1] = np.repeat(rng.uniform(0.1, 0.5, B), n)
        with torch.no_grad():
            dx = f.forward(torch.tensor(0.0), torch.from_numpy(x)).numpy()
        key = f"rhs_single_{gname}_B{B}_H{H}"
        out[key] = dict(x=x, dx=dx, edges=np.asarray(list(G.edges()), dtype=np.int32), n=np.int32(n),
                        B=np.int32(B), **_params(m))

    # ---- A3(+A4 restated): ODEBlock.forward (single graph)
    for gname, B, H, maxTime, deltaT in [("karate", 2, 64, 20, 0.5), ("loops40", 3, 8, 6, 0.5),
                                         ("er200", 2, 64, 5, 0.5), ("karate", 1, 16, 3, 0.25),
                                         ("karate", 2, 64, 30, 0.5),      # the headline horizon: 59 Euler steps
                                         ("loops40", 2, 64, 12, 1.0)]:    # deltaT = 1: every grid point is kept
        G = graphs[gname]
        A = nx.adjacency_matrix(G)
        n = A.shape[0]
        f, m = _model_single(single, A, H, maxTime, deltaT, seed=B + H)
        x = O.make_samples(n, B, H, seed=B * 7 + H)
        with torch.no_grad():
            S, I, R = m(torch.from_numpy(x))
            St = helpers.get_sir_t_nodes_torch(torch.squeeze(S, -1), maxTime, deltaT, count=False)
        key = f"fwd_single_{gname}_B{B}_H{H}_T{maxTime}"
        out[key] = dict(x=x, S=S.numpy(), I=I.numpy(), R=R.numpy(), S_sub=St.numpy(),
                        edges=np.asarray(list(G.edges()), dtype=np.int32), n=np.int32(n),
                        maxTime=np.int32(maxTime), deltaT=np.float64(deltaT), **_params(m))

    # ---- A2/A3 multi-graph: ODEfunc.forward / ODEBlock.forward of ode_nn_ngraphs.py
    names = ["karate", "loops40", "er200"]
    A_list = [nx.adjacency_matrix(graphs[k]) for k in names]
    for picks, H, maxTime in [([0, 2, 1, 0], 8, 5), ([1, 1, 2], 64, 4)]:
        torch.set_default_dtype(torch.float32)
        torch.manual_seed(len(picks) + H)
        f = multi.ODEfunc(A_list, H, torch.device("cpu"))
        m = multi.ODEBlock(maxTime, 0.5, H, f, torch.device("cpu"))
        xs = []
        rng = np.random.default_rng(H)
        for p in picks:
            n = A_list[p].shape[0]
            xi = O.make_samples(n, 1, H, seed=int(rng.integers(1 << 30)))[0]
            xi[0, 3 + 2] = p + 1                                     # marker, ode_nn_ngraphs.py:333
            xs.append(xi)
        x = np.concatenate(xs, 0)
        tot = x.shape[0]
        st = rng.uniform(0, 1.5, size=(4, tot, H)).astype(np.float32)
        st[3] = x[:, 3:]
        with torch.no_grad():
            dst = f.forward(torch.tensor(0.0), torch.from_numpy(st)).numpy()
            S, I, R = m(torch.from_numpy(x))
        key = f"multi_{'-'.join(map(str, picks))}_H{H}"
        d = dict(x=x, state=st, dstate=dst, S=S.numpy(), I=I.numpy(), R=R.numpy(), picks=np.asarray(picks, dtype=np.int32),
                 maxTime=np.int32(maxTime), deltaT=np.float64(0.5), **_params(m))
        for j, k in enumerate(names):
            d[f"edges{j}"] = np.asarray(list(graphs[k].edges()), dtype=np.int32)
            d[f"n{j}"] = np.int32(A_list[j].shape[0])
        out[key] = d

    # ---- A6: loss assembly, reference expression ode_nn_ngraph_sim.py:234 evaluated on reference tensors
    key = "fwd_single_karate_B2_H64_T20"
    d = out[key]
    rng = np.random.default_rng(5)
    y = rng.dirichlet(np.ones(3), size=(2, int(d["n"]), 20))             # [B, n, T, 3] float64 labels
    S, I, R = (torch.from_numpy(d[k]) for k in "SIR")
    sub = lambda a: helpers.get_sir_t_nodes_torch(torch.squeeze(a), 20, 0.5, count=False)
    St, It, Rt = sub(S), sub(I), sub(R)
    yt = torch.from_numpy(y)
    loss = torch.nn.L1Loss()(torch.transpose(torch.cat((torch.unsqueeze(St, -1), torch.unsqueeze(It, -1),
                             torch.unsqueeze(Rt, -1)), -1), 0, 1)[:, 1:, :], yt.view(-1, yt.size(2), yt.size(3))[:, 1:, :])
    d["y"] = y
    d["loss"] = np.float64(loss.item())

    # ---- A8: sir_torch with the coin stream recorded
    torch.set_default_dtype(torch.float64)                               # as the L3 scripts run it (ode_nn.py:493)
    real_cuda, real_rand = torch.Tensor.cuda, torch.rand
    for gname, seeds, beta, gamma, sims, T, rs in [("karate", [0, 33], 0.3, 0.2, 25, 20, 1),
                                                   ("loops40", [5], 0.45, 0.15, 30, 12, 2),
                                                   ("er200", [3, 77, 150], 0.25, 0.35, 8, 15, 3)]:
        G = graphs[gname]
        coins = []

        def rec(*a, **k):
            c = real_rand(*a, **k)
            coins.append(c.numpy().astype(np.float64).ravel())
            return c

        torch.manual_seed(rs)
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.rand = rec
        try:
            S, I, R = helpers.sir_torch(G, seeds, beta, gamma, sims, T)
        finally:
            torch.Tensor.cuda, torch.rand = real_cuda, real_rand
        out[f"sir_{gname}"] = dict(edges=np.asarray(list(G.edges()), dtype=np.int32), n=np.int32(G.number_of_nodes()),
                                   seeds=np.asarray(seeds, dtype=np.int32), beta=np.float64(beta), gamma=np.float64(gamma),
                                   sims=np.int32(sims), T=np.int32(T), coins=np.concatenate(coins) if coins else np.zeros(0),
                                   S=S, I=I, R=R)

    for key, d in out.items():
        np.savez_compressed(os.path.join(HERE, key + ".npz"), **d)
        print("wrote", key, sum(np.asarray(v).nbytes for v in d.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
