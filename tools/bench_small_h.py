#!/usr/bin/env python3
"""H = 8 (the multi-graph launcher's hidden size): persistent one-launch forward / adjoint sweep vs one launch per step."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gn-ode-sir_amd"))
import numpy as np, torch
from gnode import ops, synth
from gnode.graph import DeviceGraph

dev = torch.device("cuda:0")


def ev_ms(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for name, n, m, H, tail in (("batch of 8 graphs (ER)", 22125, 249150, 8, 0.0), ("wiki-vote size heavy tail", 7066, 100736, 8, 0.5),
                            ("fb-social size", 1893, 13835, 8, 0.0), ("22k rows H=16", 22125, 249150, 16, 0.0)):
    rp, ci = synth.heavy_tail_csr(n, m, tail, seed=1) if tail else synth.er_csr(n, m, seed=1)
    g = DeviceGraph(rp, ci)
    P = {k: torch.from_numpy(v).to(dev) for k, v in synth.linear_params(H, seed=2).items()}
    x = torch.from_numpy(synth.samples(n, 1, H, seed=3)).to(dev).reshape(n, 3 + H)
    dts = ops.step_sizes(ops.time_grid(20, 0.5))
    rows = ops.subsample_rows(20, 0.5)
    gs = [torch.randn(len(rows), n, device=dev) for _ in range(3)]
    out = {"case": name, "n": n, "H": H, "steps": len(dts), "path": ops.forward_path(g, n, H, len(dts))[0]}
    for tag, pf in (("per_step", False), ("persist", True)):
        out[tag + "_fwd_ms"] = round(ev_ms(lambda: ops.forward(g, x, P, dts, "euler", rows, persist=pf)), 4)
        out[tag + "_train_fwd_ms"] = round(ev_ms(lambda: ops.forward(g, x, P, dts, "euler", rows, want_sol=True, persist=pf)), 4)
        sol = ops.forward(g, x, P, dts, "euler", rows, want_sol=True, persist=pf)[3]
        out[tag + "_bwd_ms"] = round(ev_ms(lambda: ops.backward(g, x, P, dts, "euler", rows, sol, *gs, persist=pf)), 4)
    if "--prof" in sys.argv and out["path"] == 3:          # library built with GNODE_EXTRA_FLAGS=-DGN_PERS_PROF
        import ctypes as C
        ops.forward(g, x, P, dts, "euler", rows, persist=True)
        torch.cuda.synchronize()
        ws = ops.forward.last_workspace[3]
        tk = (C.c_uint64 * 8)()
        ops._lib.load().gnode_forward_phase_ticks(C.c_int64(n), H, 0, C.c_void_p(ws.data_ptr()), tk)
        names = ["Z_S mlp", "wait", "hub segments", "gather", "update + Z_I mlp + store", "drain + barrier + flag", "outputs"]
        out["phase_us_per_step (workgroup 0)"] = {nm: round(tk[i] / 100.0 / len(dts), 3) for i, nm in enumerate(names)}
    if "--prof" in sys.argv and out["path"] == 3:
        import ctypes as C
        sol = ops.forward(g, x, P, dts, "euler", rows, want_sol=True, persist=True)[3]
        ops.backward(g, x, P, dts, "euler", rows, sol, *gs, persist=True)
        torch.cuda.synchronize()
        ws = ops.backward.last_workspace[2]
        tk = (C.c_uint64 * 8)()
        ops._lib.load().gnode_backward_phase_ticks(C.c_int64(n), H, C.c_void_p(ws.data_ptr()), tk)
        names = ["trajectory rows requested", "wait", "hub segments", "gather (2 tables)", "dpre, g_Y, head, next Z / q, store", "drain + barrier + flag", "gW, gb accumulation"]
        out["bwd_phase_us_per_interval (workgroup 0)"] = {nm: round(tk[i] / 100.0 / len(dts), 3) for i, nm in enumerate(names)}
    out["status"] = ops.forward_status()
    print(json.dumps(out), flush=True)
