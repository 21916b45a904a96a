#!/usr/bin/env python3
"""GN-ODE forward benchmark: node-timesteps/s on the 75k-node Erdos-Renyi workload.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one ODEBlock.forward (encoder + 59 Euler steps + read-out at every
grid point) over this rank's batch of 8 (beta, gamma, seed-set) samples on the
75 000-node / 500 000-edge graph, H = 64, maxTime = 30, deltaT = 0.5
(BASELINE.json configs[3]).  Samples shard across ranks with no
data-path collective (weak scaling: 8 samples per GPU); the only collectives are
the barrier and the MAX of the elapsed time.  Inputs are resident in HBM before
the timed region starts.

value = ranks * samples_per_rank * N * n_euler_steps * K / max-over-ranks seconds.
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


def algorithmic_bytes_per_sample_step(n, nnz, H, projected_R=True):
    """Dominant kernel (fused Euler step), one sample, one step, fp32 / int32 CSR (DESIGN.md
    "Kernels"):  col nnz*4 + rowptr (n+1)*4 + neighbour rows nnz*H*4  (the edge-gather
    step) + own Z_I row + Y_S, Y_I read + write + next-step Z_I write = 6*n*H*4, + the R
    compartment: Y_R read + write 2*n*H*4, or its 4-float read-out projection 2*n*16 in
    inference mode (no trajectory requested)."""
    r_bytes = 2 * n * 16 if projected_R else 2 * n * H * 4
    return nnz * 4 + (n + 1) * 4 + nnz * H * 4 + 6 * n * H * 4 + r_bytes


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

    from gnode import _lib, ops, synth
    from gnode.graph import DeviceGraph

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
    import ctypes as C
    gms, gcnt, mms, mcnt = C.c_double(), C.c_int64(), C.c_double(), C.c_int64()
    lib.gnode_profile_read(C.byref(gms), C.byref(gcnt), C.byref(mms), C.byref(mcnt))
    lib.gnode_profile_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity on the timed outputs (not timed): probabilities
    S, I, R = outs[0]
    tot = (S + I + R)
    ok = bool(torch.isfinite(tot).all().item()) and abs(float(tot.mean().item()) - 1.0) < 1e-4

    units = world * B * n * n_steps * args.steps
    value = units / elapsed
    gather_avg_s = (gms.value / max(gcnt.value, 1)) * 1e-3
    prj = H == 64                               # inference carries the projected R compartment (no trajectory requested)
    alg_bytes = algorithmic_bytes_per_sample_step(n, nnz, H, prj) * chunk
    achieved = alg_bytes / gather_avg_s / 1e9 if gather_avg_s > 0 else 0.0

    # HBM-side traffic of the same kernel from the committed PMC passes (rocprofv3 --pmc cannot run inside
    # this process): profiles/*pmc_step64.json, collected with this command line and corrected as
    # MI355X_MICROARCH.md prescribes; only reported when the workload matches.
    traffic = None
    try:
        import glob
        pm = os.path.join(ROOT, "profiles", "pmc_step64_latest.json")
        if os.path.exists(pm) and (n, nnz, H, chunk) == (75000, 1000000, 64, 8):
            traffic = float(json.load(open(pm))["traffic_bytes_per_launch"])
    except Exception:
        traffic = None

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
        "roofline": {"bound": "hbm", "kernel": "k_step64<true> (CSR pull-gather + MFMA node MLP + SIR update + read-out, one launch per Euler step)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "avg_launch_us": gather_avg_s * 1e6, "launches": int(gcnt.value),
                     "algorithmic_bytes_per_launch": alg_bytes,
                     # `achieved` counts every neighbour-row read of the gather (the algorithm's bytes), a part of which
                     # the L2 serves, so it can exceed the HBM peak; the measured L2<->fabric bytes over the same time:
                     "traffic_gbps": (traffic / gather_avg_s / 1e9 if traffic and gather_avg_s > 0 else None),
                     "traffic_frac": (traffic / gather_avg_s / 1e9 / HBM_PEAK_GBS if traffic and gather_avg_s > 0 else None),
                     # north_star's target metric: the SURVEY's edge-gather bytes alone (col + rowptr + neighbour rows + AI),
                     # charged with the WHOLE fused kernel's time, as a fraction of the same peak
                     "edge_gather_only_frac": ((nnz * 4 + (n + 1) * 4 + nnz * H * 4 + n * H * 4) * chunk / gather_avg_s / 1e9 / HBM_PEAK_GBS
                                               if gather_avg_s > 0 else 0.0),
                     "node_mlp_avg_launch_us": (mms.value / max(mcnt.value, 1)) * 1e3},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import gnode_oracle as O                 # the CPU checker: imported for this leg only
        cores = O.usable_cores()                 # the cgroup CPU share, not the 256 visible cores
        torch.set_num_threads(cores)
        cs = min(args.cpu_steps, n_steps)
        _, _, _, secs, done = O.torch_port_forward(x_host[:1], P, rp, ci, args.maxTime, args.deltaT, n_steps=cs, threads=cores)
        # second CPU figure: the optimised C + OpenMP restatement of the same path (a pull CSR gather
        # instead of the reference's repeat / gather / scatter_add_ sequence), same sample, same cores
        c_port = None
        try:
            import oracle_c as OC
            import time as _t
            t0 = _t.perf_counter()
            OC.forward_euler(rp, ci, n, x_host[:1], P, np.asarray(O.time_grid(args.maxTime, args.deltaT)[1:cs + 1] -
                                                                O.time_grid(args.maxTime, args.deltaT)[:cs], np.float32))
            c_port = n * cs / (_t.perf_counter() - t0)
        except Exception as exc:                                  # the C checker is optional for the bench line
            c_port = f"unavailable: {type(exc).__name__}"
        result["cpu_baseline"] = {"value": n * done / secs, "unit": "node-timesteps/s", "cores": cores, "kind": "port",
                                  "c_openmp_port_value": c_port,
                                  "sample": f"1 sample x {done} Euler steps of the same graph (reference op sequence "
                                            f"in PyTorch-CPU: repeat-index + gather + scatter_add_), {secs:.1f} s"}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
