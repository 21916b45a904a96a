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
