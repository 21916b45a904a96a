#!/usr/bin/env python3
"""Full-horizon golden vectors at the BASELINE graph sizes, produced by the REFERENCE classes.

Runs only in the build container (needs /root/reference; import shims as in make_golden.py).
What is executed from the reference, unchanged: ``ODEBlock.forward`` / ``ODEfunc.forward`` of
ode_nn_ngraph_sim.py, ``get_sir_t_nodes_torch`` of ode_nn.py, and the ``train()`` / ``test()`` epoch
loops of ode_nn_ngraph_sim.py:208-296 (for the loss normalisation).

  full_fb_H64_T30.npz    Erdos-Renyi G(1 893, 13 835)  -- fb-social's node / edge counts (configs[1])
  full_wiki_H64_T30.npz  Erdos-Renyi G(7 066, 100 736) -- wiki-vote's counts (configs[2])
      B = 1, H = 64, maxTime = 30, deltaT = 0.5 -> 59 Euler steps.  Graph, weights and the sample come
      from gnode/synth.py generators (numpy default_rng, seeds stored), so the GPU box rebuilds the
      inputs and only the reference's OUTPUTS travel: S, I, R (fp32, what the reference computes under
      torch.float32) at the 30 rows get_sir_t_nodes_torch keeps plus the last grid point, the same
      forward re-run by the same reference classes under torch.float64 at grid rows {20, 40, 58, 59}
      (the yardstick: how far the reference's own fp32 sits from exact), and the reference's loss
      expression (:234) against a closed-form label tensor.
  loss_epoch_karate.npz   the reference's train() (SGD, lr = 0: weights stay put) and test() loops on 5
      karate samples with batch sizes 2 / 2 / 1: epoch train loss, val loss, test loss and per-batch test
      losses -- pins Runner.train_epoch / evaluate's element weighting (:248-249, :265-266, :290-294).
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "gn-ode-sir_amd", "gnode"))
import make_golden as MG  # noqa: E402
import synth  # noqa: E402  (gnode/synth.py: numpy + scipy only)
from labels import closed_form_labels  # noqa: E402

KEEP64 = np.asarray([20, 40, 58, 59], dtype=np.int32)


def set_params(model, P, dtype):
    sd = model.state_dict()
    for k, v in P.items():
        sd[k] = torch.from_numpy(v).to(dtype)
    model.load_state_dict(sd)


def ref_loss(helpers, S, I, R, y, maxTime, deltaT):
    """the reference's loss expression, ode_nn_ngraph_sim.py:230-234, verbatim on reference tensors"""
    sub = lambda a: helpers.get_sir_t_nodes_torch(torch.squeeze(a), maxTime, deltaT, count=False)
    St, It, Rt = sub(S), sub(I), sub(R)
    return torch.nn.L1Loss()(torch.transpose(torch.cat((torch.unsqueeze(St, -1), torch.unsqueeze(It, -1),
                             torch.unsqueeze(Rt, -1)), -1), 0, 1)[:, 1:, :], y.view(-1, y.size(2), y.size(3))[:, 1:, :])


def main():
    MG._install_import_shims()
    sys.path.insert(0, MG.REF)
    cwd = os.getcwd()
    os.chdir("/tmp")
    import ode_nn_ngraph_sim as single
    import ode_nn as helpers
    os.chdir(cwd)
    dev = torch.device("cpu")
    H, maxTime, deltaT = 64, 30, 0.5
    rows = np.asarray([int(i / deltaT) for i in range(maxTime)] + [59], dtype=np.int32)

    for tag, n, m in [("fb", 1893, 13835), ("wiki", 7066, 100736)]:
        rp, ci = synth.er_csr(n, m, seed=0)
        A = sp.csr_matrix((np.ones(ci.shape[0], dtype=np.int64), ci, rp), shape=(n, n))
        P = synth.linear_params(H, seed=0)
        x = synth.samples(n, 1, H, seed=1000)
        out = {}
        for dtype, name in [(torch.float32, "f32"), (torch.float64, "f64")]:
            torch.set_default_dtype(dtype)
            f = single.ODEfunc(A, 0.2, 0.1, H, dev)
            mdl = single.ODEBlock(maxTime, deltaT, n, [0], H, f, dev)
            set_params(mdl, P, dtype)
            with torch.no_grad():
                S, I, R = mdl(torch.from_numpy(x).to(dtype))
            out[name] = (S, I, R)
        torch.set_default_dtype(torch.float32)
        S, I, R = out["f32"]
        y = torch.from_numpy(closed_form_labels(1, n, maxTime))
        loss = ref_loss(helpers, S, I, R, y, maxTime, deltaT)
        d = dict(n=np.int32(n), m=np.int32(m), graph_seed=np.int32(0), param_seed=np.int32(0), sample_seed=np.int32(1000),
                 H=np.int32(H), maxTime=np.int32(maxTime), deltaT=np.float64(deltaT), rows=rows, rows64=KEEP64,
                 loss=np.float64(loss.item()))
        for c, a32, a64 in zip("SIR", out["f32"], out["f64"]):
            d[c] = a32.numpy()[rows, :, 0].astype(np.float32)
            d[c + "64"] = a64.numpy()[KEEP64, :, 0].astype(np.float64)
        e32 = max(float(np.abs(out["f32"][k].numpy().astype(np.float64) - out["f64"][k].numpy()).max()) for k in range(3))
        d["ref_f32_vs_f64_maxabs"] = np.float64(e32)
        np.savez_compressed(os.path.join(HERE, f"full_{tag}_H64_T30.npz"), **d)
        print(f"wrote full_{tag}: reference fp32 vs reference float64, max abs over all 60 grid points = {e32:.3e}, loss {loss.item():.9f}")

    # ---- A6: the epoch loops' element weighting (train :208-270 with SGD lr=0, test :272-296), karate
    import networkx as nx
    G = nx.karate_club_graph()
    A = nx.adjacency_matrix(G)
    n, H, maxTime, deltaT, NS = A.shape[0], 64, 20, 0.5, 5
    torch.set_default_dtype(torch.float32)
    P = synth.linear_params(H, seed=3)
    f = single.ODEfunc(A, 0.2, 0.1, H, dev)
    mdl = single.ODEBlock(maxTime, deltaT, n, [0], H, f, dev)
    set_params(mdl, P, torch.float32)
    x = synth.samples(n, NS, H, seed=77)
    y = closed_form_labels(NS, n, maxTime)
    from torch.utils.data import DataLoader, TensorDataset
    ds = TensorDataset(torch.from_numpy(x), torch.from_numpy(y))
    crit = torch.nn.L1Loss()
    opt = torch.optim.SGD(mdl.parameters(), lr=0.0)
    tr_loss, val_loss = single.train(mdl, opt, crit, dev, DataLoader(ds, batch_size=2, shuffle=False),
                                     DataLoader(ds, batch_size=3, shuffle=False), maxTime, deltaT, n)
    te_loss, te_all = single.test(mdl, crit, dev, DataLoader(ds, batch_size=2, shuffle=False), maxTime, deltaT, n)
    np.savez_compressed(os.path.join(HERE, "loss_epoch_karate.npz"), n=np.int32(n), H=np.int32(H), maxTime=np.int32(maxTime),
                        deltaT=np.float64(deltaT), NS=np.int32(NS), param_seed=np.int32(3), sample_seed=np.int32(77),
                        edges=np.asarray(list(G.edges()), dtype=np.int32), train_loss=np.float64(tr_loss),
                        val_loss=np.float64(val_loss), test_loss=np.float64(te_loss), test_all=np.asarray(te_all, dtype=np.float64))
    print("wrote loss_epoch_karate", tr_loss, val_loss, te_loss, te_all)


if __name__ == "__main__":
    main()
