#!/usr/bin/env python3
"""GN-ODE forward benchmark: node-timesteps/s on the 75k-node Erdos-Renyi workload.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one ODEBlock.forward (encoder + 59 Euler steps + read-out at every
grid point) over this rank's batch of 8 (beta, gamma, seed-set) samples on the
75 000-node / 500 000-edge graph, H = 64, maxTime = 30, deltaT = 0.5
(BASELINE.json configs[3]).  Samples shard across ranks with no
data-path collective (weak scaling: 8 samples per GPU); the only collectives are
the barrier and the MAX of the elapsed time.  Inputs are resident in HBM before
the timed region starts.

value = ranks * samples_per_rank * N * n_euler_steps * K / max-over-ranks seconds.

Besides the contract's keys the JSON line carries (rank 0, N = 1 only; none of it is in `value`):
  roofline      the step kernel against the memory system, several honest views (see `roofline_views`)
  cpu_baseline  the reference's op sequence in PyTorch-CPU + the C/OpenMP restatement, timed on the host cores,
                and the check of the TIMED GPU outputs against that C restatement (`outputs_checked`)
  train         forward + adjoint backward + Adam, ms per step, on configs[1]'s shape and on the 75k graph x 4, each with a
                gradient check (default path vs the recomputing per-interval path; configs[1]: vs the reference-class fixture)
  tiny          configs[0] (karate size, batch 1: the reference's shipped experiment): forward, trainer step eager / HIP graph
  mid           the reference's REAL regime (monitorer-sim.py:10: batch_size 1; fb-social / wiki-vote sizes) on graphs with
                those datasets' degree tails: forward and training step at B = 1, persistent one-launch path vs one launch
                per Euler step
  h8            configs[4]'s shape (monitorer-ngraphs.py:20: hidden 8): training step on a concatenated batch of 8 graphs
                of the five training sizes, evaluation forward on 8 x 75k nodes, kernel averages and byte fractions
  sir           the Monte-Carlo label generator on configs[2]'s shape (10 000 sims x T = 20)
With --gpus N > 1 the line keeps `roofline` (rank 0's kernel) and adds every rank's elapsed time (`rank_elapsed_s`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6290.0    # measured float4 copy, same guide
GATHER_CEILING_GBS = (7400.0, 7900.0)   # same guide, "Indexed rows": uniformly random rows of a 151 MB table, chip-wide


def step_bytes(n, nnz, H, projected_R=True):
    """Byte counts of ONE sample in ONE launch of the fused Euler-step kernel (fp32, int32 CSR):
    algorithmic  every byte the algorithm names: col nnz*4 + rowptr (n+1)*4 + neighbour rows nnz*H*4 (the edge-gather
                 step) + own Z_I row + Y_S, Y_I read + write + next-step Z_I write = 6*n*H*4, + the R compartment
                 (2*n*16 projected, 2*n*H*4 full).  Counts a table row once per EDGE that reads it.
    compulsory   what must cross HBM if every table row were fetched ONCE per launch: the same without the nnz*H*4
                 re-reads (the table's n*H*4 is already in the 6 slabs)."""
    r_bytes = 2 * n * 16 if projected_R else 2 * n * H * 4
    csr = nnz * 4 + (n + 1) * 4
    compulsory = csr + 6 * n * H * 4 + r_bytes
    return {"algorithmic": compulsory + nnz * H * 4, "compulsory": compulsory,
            "edge_gather_only": csr + nnz * H * 4 + n * H * 4}


def self_launch(n_gpus: int) -> int:
    """Parent side of `python bench.py --gpus N` (N > 1) without an outer launcher.  No HIP call happens in this
    process: the ranks are children started by torch.distributed.run on a free loopback port."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd).returncode


def prof_read(lib, kind):
    import ctypes as C
    ms, cnt = C.c_double(), C.c_int64()
    lib.gnode_profile_read_kind(kind, C.byref(ms), C.byref(cnt))
    return ms.value, cnt.value


def traffic_record(n, nnz, H, chunk):
    """HBM-side traffic of the step kernel from the committed PMC passes (rocprofv3 --pmc cannot run inside this
    process): profiles/pmc_step64_latest.json, collected with this command line and corrected as MI355X_MICROARCH.md
    prescribes.  Reported only when the workload matches, and always with its provenance."""
    pm = os.path.join(ROOT, "profiles", "pmc_step64_latest.json")
    try:
        if os.path.exists(pm) and (n, nnz, H, chunk) == (75000, 1000000, 64, 8):
            d = json.load(open(pm))
            return float(d["traffic_bytes_per_launch"]), {"file": "profiles/pmc_step64_latest.json", "measured_in_this_run": False,
                                                          "kernel": d.get("kernel"), "collected": d.get("collected"),
                                                          "tree": d.get("tree"), "fabric_read_bytes": d.get("fabric_read_bytes_per_launch")}
    except Exception:
        pass
    return None, None


def bench_train(lib, dev, n, m, B, H, maxTime, deltaT, reps):
    """forward (trajectory kept) + L1 loss + adjoint backward + Adam through the reference's call surface"""
    import scipy.sparse as sp
    import torch
    from gnode import ops, synth
    from gnode.autograd import l1_loss_sum, l1_loss_mean_backward
    from gnode.ode_nn_ngraph_sim import ODEBlock, ODEfunc
    rp, ci = synth.er_csr(n, m, seed=0)
    nnz = int(ci.shape[0])
    A = sp.csr_matrix((np.ones(nnz), ci, rp), shape=(n, n))
    model = ODEBlock(maxTime, deltaT, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
    P = synth.linear_params(H, seed=0)
    model.load_state_dict({**model.state_dict(), **{k: torch.from_numpy(v) for k, v in P.items()}})
    x = torch.from_numpy(synth.samples(n, B, H, seed=7)).to(dev)
    y = torch.from_numpy(np.random.default_rng(0).dirichlet(np.ones(3), size=(B * n, maxTime))).to(dev)
    rows = ops.subsample_rows(maxTime, deltaT)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)

    def step():
        opt.zero_grad()
        S, I, R = model(x, out_rows=rows)
        loss = l1_loss_mean_backward(S, I, R, y, B * n * (maxTime - 1) * 3, 1)   # L1Loss over [:, 1:, :] and its backward, as the trainer runs them
        opt.step()
        return loss

    step(); step()
    torch.cuda.synchronize()
    lib.gnode_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(reps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    fwd_ms, fwd_n = prof_read(lib, 0)
    bwd_ms, bwd_n = prof_read(lib, 2)
    lib.gnode_profile_enable(0)
    n_steps = len(ops.time_grid(maxTime, deltaT)) - 1
    path = ops.forward_path(model.odefunc.graph, B * n, H, n_steps, len(rows), want_sol=True)[0]
    out = {"shape": f"ER n={n} nnz={nnz} B={B} H={H} {n_steps} Euler steps", "ms_per_step": dt * 1e3,
           "node_timesteps_per_s": B * n * n_steps / dt, "loss_finite": bool(torch.isfinite(loss).item())}
    if path == 2:
        # one persistent launch per forward / per adjoint sweep (intervals n_steps-1 .. 1; the last interval is its own launch)
        out["fwd_launch_avg_us (all %d Euler steps)" % n_steps] = fwd_ms / max(fwd_n, 1) * 1e3
        out["bwd_launch_avg_us (%d intervals; 2 launches when the batch exceeds one resident grid)" % (n_steps - 1)] = bwd_ms / max(bwd_n, 1) * 1e3
        out["fwd_us_per_euler_step"] = fwd_ms / max(fwd_n, 1) * 1e3 / n_steps
        if fwd_n:
            sbf = step_bytes(n, nnz, H, False)
            out["fwd_algorithmic_frac_of_hbm_peak"] = sbf["algorithmic"] * B * n_steps / (fwd_ms / fwd_n * 1e-3) / 1e9 / HBM_PEAK_GBS
        out["limiter"] = "latency: one group barrier (~2 us of flag flight) and one gather round trip (~1 us) per step; see DESIGN.md section 4.2"
        bwd_n = 0
    else:
        out["fwd_step_kernel_avg_us"] = fwd_ms / max(fwd_n, 1) * 1e3
        out["bwd_interval_kernel_avg_us"] = bwd_ms / max(bwd_n, 1) * 1e3
    out["path"] = {0: "one launch per Euler step", 1: "one-workgroup launch", 2: "persistent one-launch"}[path]
    # gradient check (not timed): the default path (kept activations; persistent launches where they apply) against the
    # recomputing backward behind one launch per Euler step / interval, same weights, same batch
    try:
        model.load_state_dict({**model.state_dict(), **{k: torch.from_numpy(v) for k, v in P.items()}})
        grads = []
        for keep, persist in ((None, None), (False, False)):
            old = (ops.KEEP_DEFAULT, ops.PERSIST_DEFAULT)
            if keep is not None: ops.KEEP_DEFAULT, ops.PERSIST_DEFAULT = keep, persist
            try:
                model.zero_grad()
                S, I, R = model(x, out_rows=rows)
                (l1_loss_sum(S, I, R, y, 1) / (B * n * (maxTime - 1) * 3)).backward()
                grads.append({k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None})
            finally:
                ops.KEEP_DEFAULT, ops.PERSIST_DEFAULT = old
        rel = {k: float((grads[0][k] - grads[1][k]).abs().max() / (grads[1][k].abs().max() + 1e-30)) for k in grads[0] if k != "linearS2.bias"}
        out["gradient_check"] = {"default_vs_recomputing_per_interval_max_rel": max(rel.values()), "worst": max(rel, key=rel.get),
                                 "pass": bool(max(rel.values()) <= 2e-4), "gW_checksum": float(grads[0]["odefunc.linear.weight"].double().sum().item())}
    except Exception as exc:
        out["gradient_check"] = f"unavailable: {type(exc).__name__}: {exc}"
    del model, x, y
    torch.cuda.empty_cache()
    return out


def reference_fixture_gradient(dev):
    """configs[1]'s shape, B = 1, the full 59-interval horizon: the product's training gradient against the one the
    REFERENCE's classes produced in float64 (tests/golden/adjoint_fb_H64_T30.npz: data only, written by
    tests/golden/make_golden_adjoint.py in the build container)."""
    import scipy.sparse as sp
    import torch
    from gnode import ops, synth
    from gnode.autograd import l1_loss_sum, l1_loss_mean_backward
    from gnode.ode_nn_ngraph_sim import ODEBlock, ODEfunc
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from labels import closed_form_labels
    d = dict(np.load(os.path.join(ROOT, "tests", "golden", "adjoint_fb_H64_T30.npz")))
    n, B, H, maxTime, deltaT = int(d["n"]), int(d["B"]), int(d["H"]), int(d["maxTime"]), float(d["deltaT"])
    rp, ci = synth.er_csr(n, int(d["m"]), seed=int(d["graph_seed"]))
    A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
    P = synth.linear_params(H, seed=int(d["param_seed"]))
    model = ODEBlock(maxTime, deltaT, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
    model.load_state_dict({**model.state_dict(), **{k: torch.from_numpy(v) for k, v in P.items()}})
    x = torch.from_numpy(synth.samples(n, B, H, seed=int(d["sample_seed"]))).to(dev)
    y = torch.from_numpy(closed_form_labels(B, n, maxTime).reshape(B * n, maxTime, 3)).to(dev)
    S, I, R = model(x, out_rows=ops.subsample_rows(maxTime, deltaT))
    loss = l1_loss_sum(S, I, R, y, 1) / (B * n * (maxTime - 1) * 3)
    loss.backward()
    rel = {}
    for k, v in model.named_parameters():
        if v.grad is None or k == "linearS2.bias" or "G:" + k not in d:
            continue
        want = d["G:" + k]
        rel[k] = float(np.abs(v.grad.cpu().numpy().astype(np.float64) - want).max() / (np.abs(want).max() + 1e-30))
    return {"fixture": "tests/golden/adjoint_fb_H64_T30.npz (reference ODEBlock / ODEfunc / loss, float64, 59 intervals)",
            "loss_abs_err": abs(float(loss.item()) - float(d["loss"])), "max_rel_err": max(rel.values()), "worst": max(rel, key=rel.get),
            "reference_fp32_vs_its_float64_max_rel": max(float(np.abs(d["G32:" + k] - d["G:" + k]).max() / (np.abs(d["G:" + k]).max() + 1e-30)) for k in rel),
            "pass": bool(max(rel.values()) <= 2e-4)}


def bench_train_dist(lib, dev, dist, world, rank, backend):
    import scipy.sparse as sp
    import torch
    from gnode import ops, sharding, synth
    from gnode.autograd import l1_loss_sum, l1_loss_mean_backward
    from gnode.ode_nn_ngraph_sim import ODEBlock, ODEfunc
    n, m, B, H, maxTime, deltaT = 1893, 13835, 8, 64, 30, 0.5
    rp, ci = synth.er_csr(n, m, seed=0)
    A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
    model = ODEBlock(maxTime, deltaT, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
    model.load_state_dict({**model.state_dict(), **{k: torch.from_numpy(v) for k, v in synth.linear_params(H, seed=0).items()}})
    x = torch.from_numpy(synth.samples(n, B, H, seed=100 + rank)).to(dev)
    y = torch.from_numpy(np.random.default_rng(rank).dirichlet(np.ones(3), size=(B * n, maxTime))).to(dev)
    rows = ops.subsample_rows(maxTime, deltaT)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    params = list(model.parameters())
    scale = 1.0 / (world * B * n * (maxTime - 1) * 3)            # element mean over the GLOBAL batch (ode_nn_ngraph_sim.py:248-249)
    t_ar = [0.0]

    def step(time_ar):
        opt.zero_grad()
        S, I, R = model(x, out_rows=rows)
        l1_loss_sum(S, I, R, y, 1).backward()
        if time_ar:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        sharding.allreduce_flat_grads(params, scale=scale)
        if time_ar:
            torch.cuda.synchronize(); t_ar[0] += time.perf_counter() - t0
        opt.step()

    for _ in range(3):
        step(False)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        step(False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for _ in range(reps):
        step(True)
    t = torch.tensor([dt / reps, t_ar[0] / reps], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    w0 = params[0].detach().double().sum().reshape(1).to(t.device)
    wmin, wmax = w0.clone(), w0.clone()
    dist.all_reduce(wmin, op=dist.ReduceOp.MIN); dist.all_reduce(wmax, op=dist.ReduceOp.MAX)
    return {"shape": f"ER n={n}, {B} samples per rank x {world} ranks, H={H}, 59 Euler steps", "ms_per_step_max_over_ranks": float(t[0]) * 1e3,
            "gradient_allreduce_ms (4 809 floats, flat, timed with a sync on both sides)": float(t[1]) * 1e3,
            "samples_per_s": world * B / float(t[0]), "weights_identical_on_all_ranks": bool(float(wmin) == float(wmax))}


def bench_train_dist_h8(lib, dev, dist, world, rank, backend):
    """configs[4] under N ranks (monitorer-ngraphs.py:10,20,22: hidden 8, batches of 8 graphs concatenated along the node axis, RCCL
    gradient all-reduce): every rank trains on its own concatenated batch of 8 graphs of the five training sizes (weak scaling:
    8 N graphs per optimiser step), local forward + adjoint sweep, ONE flat all-reduce of the 129-float gradient, identical Adam
    step on every rank."""
    import scipy.sparse as sp
    import torch
    from gnode import ops, sharding, synth
    from gnode import ode_nn_ngraphs as multi
    from gnode.autograd import l1_loss_sum, l1_loss_mean_backward
    H, maxTime, deltaT = 8, 20, 0.5
    sizes = [(62, 159), (620, 2102), (1893, 13835), (2905, 15645), (7066, 100736)]
    csr = [synth.er_csr(n, m, seed=n) for n, m in sizes]
    A_list = [sp.csr_matrix((np.ones(c.shape[0]), c, r), shape=(len(r) - 1, len(r) - 1)) for r, c in csr]
    model = multi.ODEBlock(maxTime, deltaT, H, multi.ODEfunc(A_list, H, dev), dev).to(dev)
    model.load_state_dict({**model.state_dict(), **{k: torch.from_numpy(v) for k, v in synth.linear_params(H, seed=0).items()}})
    picks = [0, 1, 2, 3, 4, 2, 1, 4]
    xs = []
    for j, p in enumerate(picks):
        xi = synth.samples(sizes[p][0], 1, H, seed=1000 * rank + j)[0]
        xi[0, 5] = p + 1
        xs.append(xi)
    x = torch.from_numpy(np.concatenate(xs, 0)).to(dev)
    tot = x.shape[0]
    y = torch.from_numpy(np.random.default_rng(rank).dirichlet(np.ones(3), size=(tot, maxTime))).to(dev)
    rows = ops.subsample_rows(maxTime, deltaT)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    params = list(model.parameters())
    scale = 1.0 / (world * tot * (maxTime - 1) * 3)
    t_ar = [0.0]

    def step(time_ar):
        opt.zero_grad()
        S, I, R = model(x, out_rows=rows, picks=picks)
        l1_loss_sum(S, I, R, y, 1).backward()
        if time_ar:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        sharding.allreduce_flat_grads([p for p in params if p.grad is not None], scale=scale)
        if time_ar:
            torch.cuda.synchronize(); t_ar[0] += time.perf_counter() - t0
        opt.step()

    for _ in range(3):
        step(False)
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        step(False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for _ in range(reps):
        step(True)
    t = torch.tensor([dt / reps, t_ar[0] / reps], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    w0 = model.odefunc.linear.weight.detach().double().sum().reshape(1).to(t.device)
    wmin, wmax = w0.clone(), w0.clone()
    dist.all_reduce(wmin, op=dist.ReduceOp.MIN); dist.all_reduce(wmax, op=dist.ReduceOp.MAX)
    path = ops.forward_path(model.odefunc.graph_for_picks(picks), tot, H, len(ops.time_grid(maxTime, deltaT)) - 1, len(rows), want_sol=True)[0]
    return {"shape": f"8 graphs (62..7066 nodes, {tot} rows) per rank x {world} ranks, H={H}, 39 Euler steps", "ms_per_step_max_over_ranks": float(t[0]) * 1e3,
            "gradient_allreduce_ms (129 floats, flat, timed with a sync on both sides)": float(t[1]) * 1e3,
            "graphs_per_s": world * len(picks) / float(t[0]), "weights_identical_on_all_ranks": bool(float(wmin) == float(wmax)),
            "path": {0: "one launch per Euler step / interval", 3: "persistent one-launch (gnode_persg.hip)"}.get(path, str(path))}


def _ev_ms(fn, reps, warm=2):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def bench_mid(lib, dev):
    """The regime the reference actually runs (monitorer-sim.py:10,17-22: batch_size 1 on fb-social / wiki-vote) on graphs with
    those datasets' degree tails (gnode/synth.py heavy_tail_csr: longest row ~740 / ~1 020): ODEBlock.forward with all 60
    grid points and one training step at B = 1; the persistent one-launch path (default) next to one launch per step."""
    import scipy.sparse as sp
    import torch
    from gnode import ops, synth
    from gnode.autograd import l1_loss_sum, l1_loss_mean_backward
    from gnode.ode_nn_ngraph_sim import ODEBlock, ODEfunc
    out = {}
    H, maxTime, deltaT = 64, 30, 0.5
    n_steps = len(ops.time_grid(maxTime, deltaT)) - 1
    rows = ops.subsample_rows(maxTime, deltaT)
    for name, n, m, tail in (("fb-social size, heavy tail", 1893, 13835, 0.8), ("wiki-vote size, heavy tail", 7066, 100736, 0.5)):
        rp, ci = synth.heavy_tail_csr(n, m, tail, seed=0)
        nnz = int(ci.shape[0])
        A = sp.csr_matrix((np.ones(nnz), ci, rp), shape=(n, n))
        model = ODEBlock(maxTime, deltaT, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
        model.load_state_dict({**model.state_dict(), **{k: torch.from_numpy(v) for k, v in synth.linear_params(H, seed=0).items()}})
        x = torch.from_numpy(synth.samples(n, 1, H, seed=3)).to(dev)
        y = torch.from_numpy(np.random.default_rng(0).dirichlet(np.ones(3), size=(n, maxTime))).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)

        def fwd():
            with torch.no_grad():
                model(x)

        def train():
            opt.zero_grad()
            S, I, R = model(x, out_rows=rows)
            l1_loss_mean_backward(S, I, R, y, n * (maxTime - 1) * 3, 1)      # the trainer's step (gnode/trainer.py): loss kernel writes dloss/d(S,I,R)
            opt.step()

        rec = {"n": n, "nnz": nnz, "longest_row": int(np.diff(rp).max()), "B": 1, "euler_steps": n_steps,
               "plan (tiles per workgroup, workgroups, XCDs per sample, samples per XCD, samples at once)":
                   ops.forward_path(model.odefunc.graph, n, H, n_steps)[1]}
        for label, persist in (("persistent", True), ("per_step", False)):
            old = ops.PERSIST_DEFAULT
            ops.PERSIST_DEFAULT = persist
            try:
                rec[label + "_forward_ms"] = _ev_ms(fwd, 20)
                rec[label + "_train_step_ms"] = _ev_ms(train, 10)
            finally:
                ops.PERSIST_DEFAULT = old
        rec["persistent_us_per_euler_step"] = rec["persistent_forward_ms"] * 1e3 / n_steps
        sb = step_bytes(n, nnz, H, True)
        rec["forward_algorithmic_frac_of_hbm_peak"] = sb["algorithmic"] * n_steps / (rec["persistent_forward_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        rec["limiter"] = "latency: per step one group barrier (~2 us flag flight) + one gather round trip (~1 us) + the hub segments' round trips"
        out[name] = rec
        del model, x, y
    torch.cuda.empty_cache()
    return out


def bench_tiny(lib, dev):
    """configs[0]: karate size (34 nodes, 78 edges), hidden 64, maxTime 20, batch_size 1 -- the reference's shipped default
    experiment (monitorer-sim.py:10-22): inference forward, and the trainer's step (forward + L1 loss + adjoint sweep + Adam) as
    the drop-in scripts run it: eager launches and HIP-graph replay."""
    import scipy.sparse as sp
    import torch
    from gnode import ops, synth
    from gnode.ode_nn_ngraph_sim import ODEBlock, ODEfunc
    from gnode.trainer import Runner
    n, m, H, maxTime, deltaT = 34, 78, 64, 20, 0.5
    rp, ci = synth.er_csr(n, m, seed=1)
    A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
    model = ODEBlock(maxTime, deltaT, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
    x1 = torch.from_numpy(synth.samples(n, 1, H, seed=2)).to(dev)
    n_steps = len(ops.time_grid(maxTime, deltaT)) - 1

    def fwd():
        with torch.no_grad():
            model(x1)

    out = {"shape": f"ER n={n} nnz={int(ci.shape[0])} B=1 H={H} {n_steps} Euler steps", "forward_ms": _ev_ms(fwd, 30),
           "path": {0: "one launch per Euler step", 1: "one-workgroup launch", 2: "persistent one-launch"}[ops.forward_path(model.odefunc.graph, n, H, n_steps)[0]]}
    xs = [x1.cpu()[0]] * 8
    ys = [torch.from_numpy(np.random.default_rng(0).dirichlet(np.ones(3), size=(n, maxTime)))] * 8
    for mode in (True, False):                  # (the capture first: no autograd state of eager steps alive next to it)
        run = Runner(model, 1e-3, maxTime, deltaT, dev, stack=True, use_graphs=mode)
        xp, yp = run.place(xs, ys)
        run.train_epoch(xp, yp, 1, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ep in range(3):
            run.train_epoch(xp, yp, 1, ep)
        torch.cuda.synchronize()
        out["trainer_step_%s_ms" % ("hip_graph" if mode else "eager")] = (time.perf_counter() - t0) / (3 * 8) * 1e3
    del model
    torch.cuda.empty_cache()
    return out


def bench_h8(lib, dev):
    """configs[4]'s shape (monitorer-ngraphs.py:10,20,22: hidden 8, batch_size 8, train on dolphins / fb-food / fb-social /
    openflights / wiki-vote, evaluate on epinions): a training step on a concatenated batch of 8 graphs of the five training
    sizes, and the evaluation forward on 8 x 75k nodes (Erdos-Renyi stand-ins of those node / edge counts)."""
    import scipy.sparse as sp
    import torch
    from gnode import ops, synth
    from gnode import ode_nn_ngraphs as multi
    from gnode.autograd import l1_loss_sum, l1_loss_mean_backward
    H, maxTime, deltaT = 8, 20, 0.5                                   # monitorer-ngraphs.py:14,20: maxTime 20
    n_steps = len(ops.time_grid(maxTime, deltaT)) - 1
    rows = ops.subsample_rows(maxTime, deltaT)
    sizes = [(62, 159), (620, 2102), (1893, 13835), (2905, 15645), (7066, 100736), (75000, 500000)]
    csr = [synth.er_csr(n, m, seed=n) for n, m in sizes]
    A_list = [sp.csr_matrix((np.ones(c.shape[0]), c, r), shape=(len(r) - 1, len(r) - 1)) for r, c in csr]
    model = multi.ODEBlock(maxTime, deltaT, H, multi.ODEfunc(A_list, H, dev), dev).to(dev)
    torch.manual_seed(0)

    def batch(picks):
        xs = []
        for j, p in enumerate(picks):
            xi = synth.samples(sizes[p][0], 1, H, seed=j)[0]
            xi[0, 5] = p + 1                                          # the graph marker (ode_nn_ngraphs.py:333)
            xs.append(xi)
        return torch.from_numpy(np.concatenate(xs, 0)).to(dev)

    out = {}
    # ---- training step: batch of 8 over the five training graphs
    x = batch([0, 1, 2, 3, 4, 2, 1, 4])
    tot = x.shape[0]
    nnz = sum(int(csr[p][1].shape[0]) for p in [0, 1, 2, 3, 4, 2, 1, 4])
    y = torch.from_numpy(np.random.default_rng(0).dirichlet(np.ones(3), size=(tot, maxTime))).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)

    picks_train = [0, 1, 2, 3, 4, 2, 1, 4]
    known = [True]

    def train():
        # as the drop-in trainer runs it (gnode/trainer.py): each sample's graph marker was read once when the fixed batches
        # were formed; known[0] = False: the markers are read off the batch on every forward (one host sync per step)
        opt.zero_grad()
        S, I, R = model(x, out_rows=rows, picks=picks_train if known[0] else None)
        l1_loss_mean_backward(S, I, R, y, tot * (maxTime - 1) * 3, 1)
        opt.step()

    def measure():
        train(); train()
        torch.cuda.synchronize()
        lib.gnode_profile_enable(1)
        ms = _ev_ms(train, 10, warm=0)
        f_ms, f_n = prof_read(lib, 0)
        b_ms, b_n = prof_read(lib, 2)
        lib.gnode_profile_enable(0)
        return ms, f_ms / max(f_n, 1) * 1e3, b_ms / max(b_n, 1) * 1e3

    sb = step_bytes(tot, nnz, H, False)
    g_batch = model.odefunc.graph_for(x[:, 3 + 2])
    path = ops.forward_path(g_batch, tot, H, n_steps, len(rows), want_sol=True)[0]
    ms, f_us, b_us = measure()
    leg = {"sum_nodes": tot, "nnz": nnz, "H": H, "euler_steps": n_steps, "train_step_ms": ms,
           "path": {0: "one launch per Euler step / interval", 3: "persistent one-launch (gnode_persg.hip)"}.get(path, str(path))}
    if path == 3:
        leg["fwd_launch_avg_us (all %d Euler steps, k_persg)" % n_steps] = f_us
        leg["bwd_launch_avg_us (all %d intervals, k_persg_bwd)" % n_steps] = b_us
        leg["fwd_us_per_euler_step"] = f_us / n_steps
        leg["bwd_us_per_interval"] = b_us / n_steps
        leg["fwd_algorithmic_frac_of_hbm_peak"] = sb["algorithmic"] * n_steps / (f_us * 1e-6) / 1e9 / HBM_PEAK_GBS
        leg["limiter"] = "latency: per step one group barrier over 173 workgroups on 8 XCDs (~2-3 us of flag flight) + one gather round trip of 32-byte rows + the store drain; 0.7 MB of state per step"
        # the same step behind one launch per Euler step / interval (GNODE_PERSIST=0's path)
        prev = ops.PERSIST_DEFAULT
        ops.PERSIST_DEFAULT = False
        try:
            ms0, f0, b0 = measure()
        finally:
            ops.PERSIST_DEFAULT = prev
        known[0] = False
        leg["train_step_ms_markers_read_every_forward"] = measure()[0]
        known[0] = True
        leg["per_step_train_step_ms"] = ms0
        leg["per_step_fwd_step_kernel_avg_us (k_step_generic)"] = f0
        leg["per_step_bwd_interval_kernel_avg_us (k_bwd_fused_generic)"] = b0
    else:
        leg["fwd_step_kernel_avg_us (k_step_generic)"] = f_us
        leg["bwd_interval_kernel_avg_us (k_bwd_fused_generic)"] = b_us
        leg["fwd_algorithmic_frac_of_hbm_peak"] = sb["algorithmic"] / (f_us * 1e-6) / 1e9 / HBM_PEAK_GBS
        leg["limiter"] = "launch latency: 23k rows x 32-byte rows is 0.7 MB of state per step"
    out["train batch of 8 graphs (62..7066 nodes)"] = leg
    del y
    # ---- evaluation forward: 8 x 75k nodes
    xe = batch([5] * 8)
    tot_e, nnz_e = xe.shape[0], 8 * int(csr[5][1].shape[0])

    def fwd():
        with torch.no_grad():
            model(xe, out_rows=rows)

    fwd(); fwd()
    torch.cuda.synchronize()
    lib.gnode_profile_enable(1)
    ms_e = _ev_ms(fwd, 5, warm=0)
    f_ms, f_n = prof_read(lib, 0)
    lib.gnode_profile_enable(0)
    sbe = step_bytes(tot_e, nnz_e, H, False)
    t_l = f_ms / max(f_n, 1) * 1e-3
    out["evaluation forward 8 x 75k nodes"] = {
        "sum_nodes": tot_e, "nnz": nnz_e, "H": H, "euler_steps": n_steps, "forward_ms": ms_e, "node_timesteps_per_s": tot_e * n_steps / (ms_e * 1e-3),
        "step_kernel_avg_us (k_step_generic)": t_l * 1e6,
        "algorithmic_bytes_per_launch": sbe["algorithmic"], "compulsory_bytes_per_launch": sbe["compulsory"],
        "algorithmic_frac_of_hbm_peak": sbe["algorithmic"] / t_l / 1e9 / HBM_PEAK_GBS if f_n else None,
        "compulsory_frac_of_hbm_peak": sbe["compulsory"] / t_l / 1e9 / HBM_PEAK_GBS if f_n else None}
    del model, x, xe
    torch.cuda.empty_cache()
    return out


def bwd_interval_bytes(n, nnz, H):
    """One sample, one interval of the adjoint kernel over kept activations (k_bwd_kept64): the A q
    gather plus the slab rows it reads and writes.  Reads a_S, a_I, a_R (3) + y_i S, I rows (2) + the forward's kept
    P_S(y_i) = A Z_I * Z_S (1 - Z_S) and Z_I(y_i) (2) + kept Z_S(y_{i-1}) (1) + y_{i-1} S, I, R rows at output grid points
    (every second interval with the fused subsample: 1.5 on average); writes a_S, a_I (2) + a_R at output points (0.5)
    + the next interval's q table (1)."""
    slab = n * H * 4
    slabs = 3 + 2 + 2 + 1 + 1.5 + 2 + 0.5 + 1
    csr = nnz * 4 + n * 20 * 4                  # column ids + row headers
    return {"algorithmic": csr + nnz * H * 4 + slabs * slab, "compulsory": csr + slabs * slab}


VALU_ISSUE_PEAK = 256 * 4 * 0.5 * 2.4e9      # wave-instructions / s: 1024 SIMDs, one wave64 vector instruction per 2 cycles (MI355X_MICROARCH.md)


def bench_sir(lib, dev, n, m, sims, T):
    """The Monte-Carlo label generator on configs[2]'s shape, two (beta, gamma) points: the bench's historical one (a fast
    burn-through: everybody is infected within a few steps, most steps only draw recovery coins) and a long-lived frontier.
    Work is COUNTED by the kernel's profiling instantiation (coins drawn, CSR entries read), not modelled."""
    import torch
    from gnode import synth
    from gnode.graph import DeviceGraph
    from gnode.ode_nn import sir_counts, sir_counts_counted
    rp, ci = synth.er_csr(n, m, seed=0)
    nnz = int(ci.shape[0])
    g = DeviceGraph(rp, ci)
    seeds = [1, n // 2]
    out = {"shape": f"ER n={n} nnz={nnz}, {sims} sims x T={T}, 2 seeds"}
    pmc = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_sir_latest.json")))
    except Exception:
        pass
    for tag, beta, gamma in (("beta=0.3 gamma=0.2", 0.3, 0.2), ("beta=0.05 gamma=0.1", 0.05, 0.1)):
        sir_counts(g, seeds, beta, gamma, 64, T, rng_seed=1)
        torch.cuda.synchronize()
        lib.gnode_profile_enable(1)
        t0 = time.perf_counter()
        cnt = sir_counts(g, seeds, beta, gamma, sims, T, rng_seed=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        k_ms, k_n = prof_read(lib, 3)
        lib.gnode_profile_enable(0)
        cnt2, st = sir_counts_counted(g, seeds, beta, gamma, sims, T, rng_seed=2)
        tk = (k_ms * 1e-3) if k_n else dt
        rec = {"seconds": dt, "kernel_ms": k_ms if k_n else None, "trajectory_steps_per_s": sims * (T - 1) / dt,
               "counted": st, "counted_run_same_counts": bool(torch.equal(cnt, cnt2)),
               "coins_per_s": (st["infection_coins"] + st["recovery_coins"]) / tk, "csr_entries_per_s": st["csr_entries_read"] / tk,
               "csr_bytes_per_s": 4.0 * st["csr_entries_read"] / tk,
               "philox_int32_mul_frac_of_chip_rate": st["philox_blocks"] * 40 / tk / (VALU_ISSUE_PEAK * 64 / 4),
               "final_attack_rate": float(1.0 - cnt[0, T - 1].double().mean().item() / sims)}
        p = pmc.get(tag)
        if p:
            rec["valu_issue_frac_of_chip_rate"] = p["SQ_INSTS_VALU"] / tk / VALU_ISSUE_PEAK
            rec["wave_cycle_shares"] = p.get("wave_cycle_shares")
            rec["pmc_source"] = {"file": "profiles/pmc_sir_latest.json", "measured_in_this_run": False, "tree": pmc.get("tree")}
        out[tag] = rec
    out["limiter"] = ("latency, then vector issue: half of the waves' lifetime is s_waitcnt / barrier (dependent list -> row extent -> "
                      "column ids -> bitmap loads, LDS atomics, 5 barriers per step), the vector units issue at ~1/3 of their rate; the "
                      "Philox multiplies themselves are a few per cent of the chip's integer rate (DESIGN.md section 4.3)")
    return out, (g, rp, ci, seeds, 0.3, 0.2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nodes", type=int, default=75000)
    ap.add_argument("--edges", type=int, default=500000)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--samples", type=int, default=8, help="samples per GPU")
    ap.add_argument("--chunk", type=int, default=int(os.environ.get("GNODE_CHUNK", "8")),
                    help="samples integrated together per launch sequence")
    ap.add_argument("--maxTime", type=int, default=30)
    ap.add_argument("--deltaT", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the train-step and Monte-Carlo legs")
    ap.add_argument("--cpu-steps", type=int, default=59, help="Euler steps of the bounded CPU-baseline sample")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` typed as is: start one FRESH rank per GPU (torch.distributed.run children of this
        # process, which has not touched the GPU and never will), pass rank 0's JSON line through, exit with their code.
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus})")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the GN-ODE path has no CPU fallback)")
    # one rank per GPU; the modulo only matters for rehearsals with more ranks than GPUs on a dev box
    # (GNODE_DIST_BACKEND=gloo there: RCCL refuses two ranks on one device)
    local_dev = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    backend = os.environ.get("GNODE_DIST_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from gnode import _lib, ops, sharding, synth
    from gnode.graph import DeviceGraph

    if world > max(torch.cuda.device_count(), 1):
        sharding.share_device_guard()            # rehearsal with several ranks on one GPU: no persistent launches (see there)
    lib = _lib.load()
    n, H, B = args.nodes, args.hidden, args.samples
    rp, ci = synth.er_csr(n, args.edges, seed=0)
    nnz = int(ci.shape[0])
    P = synth.linear_params(H, seed=0)
    x_host = synth.samples(n, B, H, seed=1000 + rank)            # each rank its own samples
    grid = ops.time_grid(args.maxTime, args.deltaT)
    dts = ops.step_sizes(grid)
    n_steps = int(dts.shape[0])

    g = DeviceGraph(rp, ci)
    params = {k: torch.from_numpy(v).to(dev) for k, v in P.items()}
    x = torch.from_numpy(x_host).to(dev)
    chunk = max(1, min(args.chunk, B))
    ws = torch.empty(lib.gnode_forward_workspace_bytes(g.handle, chunk * n, H, 0), dtype=torch.uint8, device=dev)

    # the headline run emits all grid points (what ODEBlock.forward returns); GNODE_BENCH_OUT=sub|last is a
    # diagnostic to price the fused read-out (fused get_sir_t_nodes subsample / final point only)
    out_mode = os.environ.get("GNODE_BENCH_OUT", "all")
    out_rows = {"all": None, "sub": ops.subsample_rows(args.maxTime, args.deltaT),
                "last": np.asarray([n_steps], dtype=np.int32)}[out_mode]

    def one_pass():
        outs = []
        for b0 in range(0, B, chunk):
            xb = x[b0:b0 + chunk].reshape(-1, 3 + H)
            outs.append(ops.forward(g, xb, params, dts, "euler", out_rows, False, ws)[:3])
        return outs

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    sync_all()
    lib.gnode_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        outs = one_pass()
    sync_all()
    elapsed = time.perf_counter() - t0
    step_ms, step_cnt = prof_read(lib, 0)
    lib.gnode_profile_enable(0)
    rank_elapsed = [elapsed]
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        t[rank] = elapsed
        dist.all_reduce(t, op=dist.ReduceOp.SUM)                    # every rank's own clock around the same K steps
        rank_elapsed = [float(v) for v in t.tolist()]
        elapsed = max(rank_elapsed)

    # sanity on the timed outputs (not timed): probabilities.  The real check (against the C restatement of the
    # reference) is in the cpu_baseline leg below: `outputs_checked`.
    S, I, R = outs[0]
    tot = (S + I + R)
    ok = bool(torch.isfinite(tot).all().item()) and abs(float(tot.mean().item()) - 1.0) < 1e-4

    units = world * B * n * n_steps * args.steps
    value = units / elapsed
    t_launch = (step_ms / max(step_cnt, 1)) * 1e-3
    sb = step_bytes(n, nnz, H, H == 64)
    per_launch = {k: v * chunk for k, v in sb.items()}
    gbs = {k: (v / t_launch / 1e9 if t_launch > 0 else 0.0) for k, v in per_launch.items()}
    traffic, traffic_src = traffic_record(n, nnz, H, chunk)
    fabric_read = traffic_src.get("fabric_read_bytes") if traffic_src else None

    result = {
        "metric": "node-timesteps/sec (N*T/s) GN-ODE fwd, 75k-node graph, hidden=64",
        "value": value, "unit": "node-timesteps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"ER G(n={n}, m={args.edges}) nnz={nnz}, H={H}, maxTime={args.maxTime}, deltaT={args.deltaT} "
                               f"-> {n_steps} Euler steps + read-out at {n_steps + 1} grid points, {B} samples per GPU "
                               f"(BASELINE configs[3] shape; configs[1..2] are parity cases)",
                   "samples_per_gpu": B, "samples_per_launch": chunk, "grid_points_emitted": out_mode, "euler_steps": n_steps,
                   "parallelism": f"sample-sharded x{world}, no data-path collective", "outputs_valid": ok},
        "node_maxTime_per_s": world * B * n * args.maxTime * args.steps / elapsed,
        # `frac` is the conservative view: bytes that MUST cross HBM once per launch / launch time / HBM spec peak.  The
        # launch is not bounded by HBM proper (see `limiter`): the other views are listed so that none has to be inferred.
        "roofline": {"bound": "fabric", "bound_note": "the memory side of the contract's hbm | mfma choice; what limits the launch is the L2<->fabric "
                     "random-row read rate, not HBM proper (`limiter`, DESIGN.md section 4.1)", "kernel": "k_step64<true> (software-pipelined CSR pull-gather + MFMA node MLPs + SIR update + read-out, one launch per Euler step)",
                     "achieved": gbs["compulsory"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs["compulsory"] / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "avg_launch_us": t_launch * 1e6, "launches": int(step_cnt),
                     "compulsory_bytes_per_launch": per_launch["compulsory"],
                     "algorithmic_bytes_per_launch": per_launch["algorithmic"],
                     "roofline_views": {
                         "compulsory_frac_of_hbm_spec_8000": gbs["compulsory"] / HBM_PEAK_GBS,
                         "compulsory_frac_of_hbm_achievable_6290": gbs["compulsory"] / HBM_ACHIEVABLE_GBS,
                         "algorithmic_frac_of_hbm_spec (every neighbour-row read counted; L2- and Infinity-Cache-served re-reads included, so it can exceed 1)":
                             gbs["algorithmic"] / HBM_PEAK_GBS,
                         "edge_gather_only_frac_of_hbm_spec (north_star's 40 % target: SURVEY 8d edge-gather bytes charged with the WHOLE kernel's time)":
                             gbs["edge_gather_only"] / HBM_PEAK_GBS,
                         "measured_l2_fabric_traffic_frac_of_hbm_spec": (traffic / t_launch / 1e9 / HBM_PEAK_GBS if traffic and t_launch > 0 else None),
                         "measured_fabric_read_gbps": (fabric_read / t_launch / 1e9 if fabric_read and t_launch > 0 else None),
                         "random_row_read_ceiling_gbps (MI355X_MICROARCH.md, 151 MB table)": list(GATHER_CEILING_GBS)},
                     "limiter": "L2<->fabric random-row read rate: the per-XCD 19.2 MB Z_I table is 5x the 4 MB L2, 83 % of neighbour-row "
                                "reads miss it and are served by the Infinity Cache; the vector L1s sit at their outstanding-miss limit "
                                "(TCP_PENDING_STALL ~65 % of cycles), fabric reads run at the guide's random-row ceiling; occupancy 3 vs 4 "
                                "workgroups per CU, +-6 % read bytes and the read-out's VALU do not move the launch time, and even a gather "
                                "working set cut to a quarter (4.8 MB per XCD) leaves 88 % of it (DESIGN.md section 4.1)"},
    }

    result["rank_elapsed_s"] = {"min": min(rank_elapsed), "max": max(rank_elapsed), "per_rank": rank_elapsed}
    if world > 1 and not args.no_secondary:
        # data-parallel training step on configs[1]'s shape, 8 samples per rank: local forward + adjoint backward, ONE flat
        # all-reduce of the 4 809-float gradient, identical Adam step on every rank; the all-reduce is timed on its own
        try:
            result["train_dist"] = bench_train_dist(lib, dev, dist, world, rank, backend)
        except Exception as exc:
            result["train_dist"] = f"unavailable: {type(exc).__name__}: {exc}"
        try:
            result["train_dist_h8"] = bench_train_dist_h8(lib, dev, dist, world, rank, backend)
        except Exception as exc:
            result["train_dist_h8"] = f"unavailable: {type(exc).__name__}: {exc}"
    single = rank == 0 and world == 1
    sir_ctx = None
    if single and not args.no_secondary:
        del ws
        torch.cuda.empty_cache()
        try:
            result["train"] = {"configs[1] shape": bench_train(lib, dev, 1893, 13835, 8, 64, 30, 0.5, 10),
                               "75k graph x 4": bench_train(lib, dev, 75000, 500000, 4, 64, 30, 0.5, 3)}
            result["train"]["configs[1] shape"]["gradient_vs_reference_classes"] = reference_fixture_gradient(dev)
            result["tiny"] = bench_tiny(lib, dev)
            result["mid"] = bench_mid(lib, dev)
            result["h8"] = bench_h8(lib, dev)
            result["sir"], sir_ctx = bench_sir(lib, dev, 7066, 100736, 10000, 20)
        except Exception as exc:                                   # never lose the headline line to a secondary leg
            result["secondary_error"] = f"{type(exc).__name__}: {exc}"

    if single and not args.no_cpu_baseline:
        import gnode_oracle as O                 # the CPU checker: imported for this leg only
        cores = O.usable_cores()                 # the cgroup CPU share, not the 256 visible cores
        torch.set_num_threads(cores)
        cs = min(args.cpu_steps, n_steps)
        _, _, _, secs, done = O.torch_port_forward(x_host[:1], P, rp, ci, args.maxTime, args.deltaT, n_steps=cs, threads=cores)
        # second CPU figure: the optimised C + OpenMP restatement of the same path (a pull CSR gather
        # instead of the reference's repeat / gather / scatter_add_ sequence), same sample, same cores -- and the
        # CHECK of the timed GPU outputs (sample 0 of the last timed pass) against it
        c_port, checked = None, None
        try:
            import oracle_c as OC
            t0 = time.perf_counter()
            Sc, Ic, Rc = OC.forward_euler(rp, ci, n, x_host[:1], P, dts[:cs])
            c_port = n * cs / (time.perf_counter() - t0)
            if out_mode == "all":
                err20 = err_all = 0.0
                for got, want in ((S, Sc), (I, Ic), (R, Rc)):
                    d = np.abs(got[:cs + 1, :n].cpu().numpy().astype(np.float64) - want[:cs + 1, :, 0])
                    err20 = max(err20, float(d[:21].max()))
                    err_all = max(err_all, float(d.max()))
                checked = {"against": "C restatement of the reference path (oracle/gnode_oracle.c), sample 0 of the last timed pass",
                           "max_abs_err_grid_points_0_20": err20, f"max_abs_err_grid_points_0_{cs}": err_all,
                           "pass": bool(err20 <= 1e-5 and err_all <= 1e-4)}
        except Exception as exc:                                  # the C checker is optional for the bench line
            c_port = f"unavailable: {type(exc).__name__}"
        result["config"]["outputs_checked"] = checked
        if checked is not None:
            result["config"]["outputs_valid"] = bool(ok and checked["pass"])
        result["cpu_baseline"] = {"value": n * done / secs, "unit": "node-timesteps/s", "cores": cores, "kind": "port",
                                  "c_openmp_port_value": c_port,
                                  "sample": f"1 sample x {done} Euler steps of the same graph (reference op sequence "
                                            f"in PyTorch-CPU: repeat-index + gather + scatter_add_), {secs:.1f} s"}
        if sir_ctx is not None:
            try:
                import oracle_c as OC
                from gnode.ode_nn import sir_counts
                gs, rps, cis, seeds, beta, gamma = sir_ctx
                t0 = time.perf_counter()
                want = OC.sir_philox(gs.n, rps, cis, seeds, beta, gamma, 64, 20, rng_seed=2)
                c_dt = time.perf_counter() - t0
                got = sir_counts(gs, seeds, beta, gamma, 64, 20, rng_seed=2).cpu().numpy().astype(np.uint32)
                result["sir"]["counts_bit_exact_vs_c_oracle_64_sim_prefix"] = bool(np.array_equal(got, want))
                result["sir"]["cpu_port_trajectory_steps_per_s"] = 64 * 19 / c_dt
            except Exception as exc:
                result["sir"]["oracle_check"] = f"unavailable: {type(exc).__name__}"
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
