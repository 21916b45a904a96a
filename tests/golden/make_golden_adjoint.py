#!/usr/bin/env python3
"""Golden PARAMETER GRADIENTS produced by the REFERENCE classes.

Runs only in the build container (needs /root/reference; import shims as in make_golden.py).  What is executed
from the reference, unchanged: ``ODEBlock.forward`` (encoder, the ``odeint`` call at ode_nn_ngraph_sim.py:168, the
read-out head and softmax), ``ODEfunc.forward`` -- whose vector-Jacobian products torch autograd takes from the
reference's own code --, ``get_sir_t_nodes_torch`` and the loss expression of :230-234, followed by
``loss.backward()``.  What is NOT the reference's: ``torchdiffeq`` is absent from this image, so the callable behind
``odeint`` is this repo's restatement of torchdiffeq 0.2.2's ``odeint_adjoint(..., method='euler')`` (SURVEY Appendix
A) as a torch.autograd.Function: forward under no_grad on the grid; backward one Euler step of the augmented system
per interval from t_i to t_{i-1} with the Jacobians taken AT (t_i, y_i), y reset to the stored sol[i-1], the output
cotangent of grid point i-1 added.  So these vectors pin everything of the training gradient -- RHS derivative, head,
encoder, loss, subsample -- EXCEPT the integrator's adjoint rule, which stays "parity unpinned" exactly as the forward
integrator does.

  adjoint_karate_H64_T20.npz   karate club, B = 2, H = 64, maxTime = 20, deltaT = 0.5 (the reference's shipped shape)
  adjoint_er200_H64_T6.npz     Erdos-Renyi G(200, 700) giant component, B = 2, H = 64, maxTime = 6
  adjoint_loops40_H8_T5.npz    40-node graph with self-loops, B = 3, H = 8, maxTime = 5
  adjoint_fb_H64_T30.npz       Erdos-Renyi G(1 893, 13 835) (fb-social's counts, configs[1]), B = 1, H = 64, maxTime = 30,
                               deltaT = 0.5 -> the FULL 59-interval adjoint the reference trains through
  adjoint_wiki_H64_T30.npz     Erdos-Renyi G(7 066, 100 736) (wiki-vote's counts, configs[2]), same horizon
      (graph by seed from gnode/synth.py instead of an edge list; besides the float64 gradients "G:<name>" these two hold
       three of the 30 output rows the loss saw ("rows_kept"), and "G32:<name>": the SAME reference classes and adjoint rule run under torch.float32 -- the yardstick that says how
       far the reference's own fp32 training gradient sits from float64 after 59 intervals)
Each: inputs by seed (gnode/synth.py generators), the edge list, and the outputs at the kept rows and the 8 parameter gradients in float64 (the reference
classes run under torch.float64: a yardstick the fp32 kernels are held to at 2e-4, like the oracle comparisons) and
the loss value.
"""
import os
import sys

import numpy as np
import networkx as nx
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "gn-ode-sir_amd", "gnode"))
import make_golden as MG  # noqa: E402
import synth  # noqa: E402
from labels import closed_form_labels  # noqa: E402
from make_golden_fullsize import ref_loss, set_params  # noqa: E402


class _AdjointEuler(torch.autograd.Function):
    """torchdiffeq's OdeintAdjointMethod under fixed-grid Euler, restated (see the module docstring)."""

    @staticmethod
    def forward(ctx, func, t, n_params, y0, *params):
        with torch.no_grad():
            sol = [y0]
            for k in range(t.shape[0] - 1):
                sol.append(sol[-1] + (t[k + 1] - t[k]) * func(t[k], sol[-1]))
            sol = torch.stack(sol)
        ctx.func, ctx.t, ctx.params = func, t, params
        ctx.save_for_backward(sol)
        return sol

    @staticmethod
    def backward(ctx, gsol):
        (sol,) = ctx.saved_tensors
        func, t, params = ctx.func, ctx.t, ctx.params
        a = gsol[-1].clone()
        gp = [torch.zeros_like(p) for p in params]
        for i in range(sol.shape[0] - 1, 0, -1):
            with torch.enable_grad():
                yi = sol[i].detach().requires_grad_(True)
                f = func(t[i], yi)
                vj = torch.autograd.grad(f, (yi,) + tuple(params), a, allow_unused=True)
            dt = t[i] - t[i - 1]
            a = a + dt * vj[0] + gsol[i - 1]
            for g, v in zip(gp, vj[1:]):
                if v is not None:
                    g += dt * v
        return (None, None, None, a, *gp)


def _odeint_adjoint(func, y0, t, method="euler", **kw):
    assert method == "euler"
    params = tuple(p for p in func.parameters() if p.requires_grad)
    return _AdjointEuler.apply(func, t, len(params), y0, *params)


def main():
    MG._install_import_shims()
    sys.modules["torchdiffeq"].odeint_adjoint = _odeint_adjoint
    sys.modules["torchdiffeq"].odeint = _odeint_adjoint
    sys.path.insert(0, MG.REF)
    cwd = os.getcwd()
    os.chdir("/tmp")
    import ode_nn_ngraph_sim as single
    import ode_nn as helpers
    os.chdir(cwd)
    dev = torch.device("cpu")
    graphs = MG._graphs()
    torch.set_default_dtype(torch.float64)
    keys = ["odefunc.linear.weight", "odefunc.linear.bias", "linearS1.weight", "linearS1.bias",
            "linear3.weight", "linear3.bias", "linearS2.weight", "linearS2.bias"]
    for gname, B, H, maxTime in [("karate", 2, 64, 20), ("er200", 2, 64, 6), ("loops40", 3, 8, 5)]:
        G = graphs[gname]
        A = nx.adjacency_matrix(G)
        n, deltaT = A.shape[0], 0.5
        seed = {"karate": 21, "er200": 22, "loops40": 23}[gname]
        P = synth.linear_params(H, seed=seed)
        f = single.ODEfunc(A, 0.2, 0.1, H, dev)
        mdl = single.ODEBlock(maxTime, deltaT, n, [0], H, f, dev)
        set_params(mdl, P, torch.float64)
        x = synth.samples(n, B, H, seed=seed + 100)
        y = torch.from_numpy(closed_form_labels(B, n, maxTime)).to(torch.float64)
        mdl.zero_grad()
        S, I, R = mdl(torch.from_numpy(x).to(torch.float64))
        loss = ref_loss(helpers, S, I, R, y, maxTime, deltaT)
        loss.backward()
        named = dict(mdl.named_parameters())
        d = dict(n=np.int32(n), B=np.int32(B), H=np.int32(H), maxTime=np.int32(maxTime), deltaT=np.float64(deltaT),
                 param_seed=np.int32(seed), sample_seed=np.int32(seed + 100), loss=np.float64(loss.item()),
                 edges=np.asarray(list(G.edges()), dtype=np.int32))
        sub = lambda a: helpers.get_sir_t_nodes_torch(torch.squeeze(a), maxTime, deltaT, count=False).detach().numpy()
        d["S"], d["I"], d["R"] = sub(S), sub(I), sub(R)          # [maxTime, B*n]: the outputs the loss saw
        for k in keys:
            d["G:" + k] = named[k].grad.detach().numpy().astype(np.float64)
        tag = f"adjoint_{gname}_H{H}_T{maxTime}"
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **d)
        print("wrote", tag, "loss", loss.item(), "|gW|", float(np.abs(d["G:odefunc.linear.weight"]).max()))
    torch.set_default_dtype(torch.float32)

    # ---- full horizon (59 intervals) at the fb-social / wiki-vote node and edge counts
    import scipy.sparse as sp
    for tag, n, m in [("fb", 1893, 13835), ("wiki", 7066, 100736)]:
        H, maxTime, deltaT, B = 64, 30, 0.5, 1
        rp, ci = synth.er_csr(n, m, seed=0)
        A = sp.csr_matrix((np.ones(ci.shape[0], dtype=np.int64), ci, rp), shape=(n, n))
        P = synth.linear_params(H, seed=0)
        x = synth.samples(n, B, H, seed=1000)
        d = dict(n=np.int32(n), m=np.int32(m), graph_seed=np.int32(0), B=np.int32(B), H=np.int32(H), maxTime=np.int32(maxTime),
                 deltaT=np.float64(deltaT), param_seed=np.int32(0), sample_seed=np.int32(1000))
        for dtype, pre in [(torch.float64, "G:"), (torch.float32, "G32:")]:
            torch.set_default_dtype(dtype)
            f = single.ODEfunc(A, 0.2, 0.1, H, dev)
            mdl = single.ODEBlock(maxTime, deltaT, n, [0], H, f, dev)
            set_params(mdl, P, dtype)
            y = torch.from_numpy(closed_form_labels(B, n, maxTime)).to(torch.float64)
            mdl.zero_grad()
            S, I, R = mdl(torch.from_numpy(x).to(dtype))
            loss = ref_loss(helpers, S, I, R, y, maxTime, deltaT)
            loss.backward()
            named = dict(mdl.named_parameters())
            for k in keys:
                d[pre + k] = named[k].grad.detach().numpy().astype(np.float64)
            if dtype == torch.float64:
                d["loss"] = np.float64(loss.item())
                sub = lambda a: helpers.get_sir_t_nodes_torch(torch.squeeze(a), maxTime, deltaT, count=False).detach().numpy()
                d["rows_kept"] = np.asarray([1, 15, 29], dtype=np.int32)     # of the maxTime rows the loss sees (the full-size forward is pinned by full_*.npz)
                d["S"], d["I"], d["R"] = (sub(S)[d["rows_kept"]], sub(I)[d["rows_kept"]], sub(R)[d["rows_kept"]])
            else:
                d["loss32"] = np.float64(loss.item())
        torch.set_default_dtype(torch.float32)
        rel = {k: float(np.abs(d["G32:" + k] - d["G:" + k]).max() / max(np.abs(d["G:" + k]).max(), 1e-30)) for k in keys}
        name = f"adjoint_{tag}_H64_T30"
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print("wrote", name, "loss", d["loss"], "reference fp32 vs float64 gradient, rel:", {k: f"{v:.1e}" for k, v in rel.items()})


if __name__ == "__main__":
    main()
