#!/usr/bin/env python3
"""Persistent one-launch integration (csrc/gnode_pers64.hip) vs one launch per Euler step on mid-size graphs:
forward latency (inference, all grid points emitted: the reference's ODEBlock.forward) and the training forward,
HIP events around REPS back-to-back calls.  One JSON line per shape.  usage: bench_persist.py [chung-lu]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gn-ode-sir_amd"))

import numpy as np
import torch

from gnode import ops, synth
from gnode.graph import DeviceGraph

dev = torch.device("cuda:0")
REPS = 20


def ev_time(fn, reps=REPS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    shapes = [("fb-social", 1893, 13835, (1, 2, 4, 8), 0.0), ("wiki-vote", 7066, 100736, (1, 2), 0.0), ("er-600", 600, 2400, (1, 8, 32), 0.0),
              ("fb-social-tail", 1893, 13835, (1, 8), 0.8), ("wiki-vote-tail", 7066, 100736, (1, 2), 0.5)]
    if "--tail" in sys.argv:
        shapes = shapes[3:]
    maxTime, deltaT = 30, 0.5
    dts = ops.step_sizes(ops.time_grid(maxTime, deltaT))
    rows_out = ops.subsample_rows(maxTime, deltaT)
    for name, n, m, Bs, tail in shapes:
        rp, ci = synth.heavy_tail_csr(n, m, tail, seed=0) if tail else synth.er_csr(n, m, seed=0)
        g = DeviceGraph(rp, ci)
        P = {k: torch.from_numpy(v).to(dev) for k, v in synth.linear_params(64, seed=0).items()}
        for B in Bs:
            x = torch.from_numpy(synth.samples(n, B, 64, seed=1)).to(dev).reshape(B * n, 67)
            path, plan = ops.forward_path(g, B * n, 64, len(dts))
            ws = torch.empty(ops._lib.load().gnode_forward_workspace_bytes(g.handle, B * n, 64, 0), dtype=torch.uint8, device=dev)
            out = {"case": name, "n": n, "max_degree": int(np.diff(rp).max()), "B": B, "steps": len(dts), "path": path, "plan(nt,wgs,span,gpx,conc)": plan}
            emit = np.asarray([len(dts)], dtype=np.int32) if "--last-only" in sys.argv else None     # read-out at the last grid point only
            for persist in (False, True):
                t = ev_time(lambda: ops.forward(g, x, P, dts, "euler", emit, workspace=ws, persist=persist))
                tt = ev_time(lambda: ops.forward(g, x, P, dts, "euler", rows_out, want_sol=True, workspace=ws, persist=persist), 10)
                key = "persist" if persist else "per_step"
                out[key + "_fwd_ms"] = round(t, 4)
                out[key + "_us_per_step"] = round(1e3 * t / len(dts), 2)
                out[key + "_train_fwd_ms"] = round(tt, 4)
                gs = [torch.randn(len(rows_out), B * n, device=dev) for _ in range(3)]
                S_, I_, R_, sol = ops.forward(g, x, P, dts, "euler", rows_out, want_sol=True, workspace=ws, persist=persist)
                tb = ev_time(lambda: ops.backward(g, x, P, dts, "euler", rows_out, sol, *gs, persist=persist), 10)
                out[key + "_bwd_ms"] = round(tb, 4)
                del sol
                assert ops.forward_status() == 0
                if persist and path == 2 and "--prof" in sys.argv:          # library built with GNODE_EXTRA_FLAGS=-DGN_PERS_PROF
                    import ctypes as C
                    ops.forward(g, x, P, dts, "euler", emit, workspace=ws, persist=True)
                    torch.cuda.synchronize()
                    tk = (C.c_uint64 * 8)()
                    ops._lib.load().gnode_forward_phase_ticks(C.c_int64(B * n), 64, 0, C.c_void_p(ws.data_ptr()), tk)
                    names = ["wait", "gather(+prev outputs)", "update+sync", "mfma_I+sync", "store+drain+flag", "mfma_S"]
                    out["phase_us_per_step"] = {nm: round(tk[i] / 100.0 / len(dts), 2) for i, nm in enumerate(names)}
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
