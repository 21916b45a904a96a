#!/usr/bin/env python3
"""Randomised cross-checks on the GPU box (not part of the test suite; run through gpurun):
  * forward with / without the kept-activation buffer: identical outputs and trajectory;
  * backward over kept activations vs the recomputing backward (2e-5 of the gradient's scale) on random shapes around
    the tile / queue / grid boundaries, with hub rows, arbitrary output subsets, n_steps from 1;
usage: python tools/fuzz_gpu.py [cases] [seed]"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(R, "gn-ode-sir_amd"), os.path.join(R, "oracle")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import gnode_oracle as O  # noqa: E402
from gnode import ops  # noqa: E402
from gnode.graph import DeviceGraph  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    worst = 0.0
    for c in range(cases):
        n = int(rng.choice([65, 66, 97, 127, 128, 129, 255, 257, 511, 1000, 1893, 4097, 7066, int(rng.integers(65, 9000))]))
        B = int(rng.integers(1, 10))
        deg = float(rng.choice([1.0, 3.0, 7.0, 14.0, 30.0]))
        m = int(n * deg / 2) + 1
        skew = rng.random() < 0.4
        rp, ci, _ = (O.chung_lu_graph if skew else O.er_graph)(n, m, seed=int(rng.integers(1 << 30)))
        H = 64
        n_steps = int(rng.integers(1, 8))
        grid = np.arange(0, (n_steps + 1) * 0.5, 0.5)[: n_steps + 1]
        dts = ops.step_sizes(grid)
        G = n_steps + 1
        out_rows = None
        if rng.random() < 0.6:
            k = int(rng.integers(1, G + 1))
            out_rows = np.sort(rng.choice(G, size=k, replace=False)).astype(np.int32)
        n_out = G if out_rows is None else len(out_rows)
        P = O.init_params(H, seed=int(rng.integers(1 << 30)))
        x = O.make_samples(n, B, H, seed=int(rng.integers(1 << 30)))
        g = DeviceGraph(rp, ci)
        Pt = {k: torch.from_numpy(v).to(dev) for k, v in P.items()}
        x2d = torch.from_numpy(x).to(dev).reshape(B * n, 3 + H)
        S1, I1, R1, sol1 = ops.forward(g, x2d, Pt, dts, "euler", out_rows, want_sol=True)
        S0, I0, R0, sol0 = ops.forward(g, x2d, Pt, dts, "euler", out_rows, want_sol=True, want_keep=False)
        rows = B * n
        assert torch.equal(S1, S0) and torch.equal(I1, I0) and torch.equal(R1, R0), (c, "outputs differ with keep")
        assert torch.equal(sol1[:, :3 * rows], sol0[:, :3 * rows]), (c, "trajectory differs with keep")
        gs = [torch.randn(n_out, rows, device=dev) for _ in range(3)]
        a = ops.backward(g, x2d, Pt, dts, "euler", out_rows, sol1, *gs)
        r = ops.backward(g, x2d, Pt, dts, "euler", out_rows, sol0, *gs)
        gmax = max(float(v.abs().max()) for v in r.values())
        for k in a:
            if k == "linearS2.bias":
                continue
            scale = max(float(r[k].abs().max()), 1e-4 * gmax) + 1e-30      # (a tensor whose exact gradient is 0 is pure round-off)
            e = float((a[k] - r[k]).abs().max()) / scale
            worst = max(worst, e)
            if e > 2e-5:
                # which of the two is off?  ask the float64 oracle (its grid is arange(0, maxTime, deltaT))
                want = O.adjoint_grads_torch(x, P, rp, ci, (n_steps + 1) * 0.5, 0.5, *[t.cpu().numpy() for t in gs],
                                             out_rows=out_rows, dtype="float64")
                for kk in a:
                    w = np.asarray(want[kk]); sc = np.abs(w).max() + 1e-30
                    print(kk, "kept vs oracle %.2e" % (np.abs(a[kk].cpu().numpy() - w).max() / sc),
                          "recomputed vs oracle %.2e" % (np.abs(r[kk].cpu().numpy() - w).max() / sc), "scale %.3e" % sc, flush=True)
                raise AssertionError((c, n, B, n_steps, out_rows, k, e))
        torch.cuda.synchronize()
        if c % 10 == 0:
            print(f"case {c}: n={n} B={B} steps={n_steps} keep={'yes' if sol1.gnode_keep is not None else 'no'} worst so far {worst:.2e}", flush=True)
    print(f"OK {cases} cases, worst kept-vs-recomputed {worst:.2e}")


if __name__ == "__main__":
    main()
