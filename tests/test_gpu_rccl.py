"""GPU: the `nccl` (= RCCL) backend actually initialised on the box, world of one rank (a one-GPU box cannot host
two RCCL ranks), with the N>1 host path's collectives forced through it -- see tests/rccl_world1_worker.py.
Runs in a child process so the test session itself never joins a process group."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_rccl_world1_collectives():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_world1_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, worker, str(port)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_WORLD1 ")]
    assert line, r.stdout[-2000:]
    out = json.loads(line[-1][len("RCCL_WORLD1 "):])
    assert out["backend"] == "nccl"
    assert out["max_ok"] and out["counts_ok"] and out["grads_ok"] and out["runner_ok"], out
    assert out["counts_dtype"] == "torch.int32"


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` typed as is: the parent (which never touches the GPU) starts two ranks with
    torch.distributed.run, rank 0 prints ONE JSON line with the whole-job aggregate.  Both ranks share the box's one GPU
    here, so the control plane is gloo (GNODE_DIST_BACKEND: RCCL refuses two ranks on one device); the launch path,
    the barrier / MAX-of-elapsed protocol and the aggregation are what is under test."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GNODE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nodes", "3000",
                        "--edges", "12000", "--samples", "2", "--chunk", "2", "--no-cpu-baseline", "--no-secondary"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["outputs_valid"]
    assert abs(d["value"] - 2 * 2 * 3000 * 59 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-6


def test_bench_two_ranks_with_training_legs():
    """the N > 1 line with its secondary legs: data-parallel training at configs[1]'s shape (`train_dist`) and at configs[4]'s
    (`train_dist_h8`: hidden 8, multi-graph batches) -- gradients all-reduced, weights identical on both ranks afterwards.
    Two ranks on the box's one GPU over gloo (a shared device runs the one-launch-per-step forms)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GNODE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--nodes", "3000",
                        "--edges", "12000", "--samples", "2", "--chunk", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["rank_elapsed_s"]["per_rank"]) == 2
    for leg in ("train_dist", "train_dist_h8"):
        assert isinstance(d[leg], dict), d[leg]
        assert d[leg]["weights_identical_on_all_ranks"] and d[leg]["ms_per_step_max_over_ranks"] > 0
