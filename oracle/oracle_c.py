"""ctypes wrapper of oracle/_build/liboracle.so (the C restatement) -- TEST INFRASTRUCTURE.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "gnode_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _lib = C.CDLL(_SO)
        try:                                   # respect the cgroup CPU share (gnode_oracle.usable_cores)
            import gnode_oracle
            _lib.oracle_set_threads(C.c_int(gnode_oracle.usable_cores()))
        except Exception:
            pass
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def rhs(rowptr, col, n, x, W, b):
    x = _f(x)
    rows, H = x.shape[0] // 4, x.shape[1]
    dx = np.empty_like(x)
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    W, b = _f(W), _f(b)
    load().oracle_rhs(_p(rp), _p(ci), C.c_int(n), C.c_long(rows), C.c_int(H), _p(x), _p(W), _p(b), _p(dx))
    return dx


def forward_euler(rowptr, col, n, x, P, dts):
    """x [B, n, 3+H] or [rows, 3+H]; returns S, I, R [G, rows, 1]."""
    x = _f(x).reshape(-1, np.shape(x)[-1])
    rows, H = x.shape[0], x.shape[1] - 3
    dts = _f(dts)
    G = dts.shape[0] + 1
    out = [np.empty((G, rows), dtype=np.float32) for _ in range(3)]
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    a = [_f(P[k]) for k in ("odefunc.linear.weight", "odefunc.linear.bias", "linearS1.weight", "linearS1.bias",
                            "linear3.weight", "linear3.bias", "linearS2.weight", "linearS2.bias")]
    load().oracle_forward_euler(_p(rp), _p(ci), C.c_int(n), C.c_long(rows), C.c_int(H), _p(x), *[_p(v) for v in a],
                                _p(dts), C.c_int(dts.shape[0]), _p(out[0]), _p(out[1]), _p(out[2]))
    return out[0][..., None], out[1][..., None], out[2][..., None]


def sir_philox(n, rowptr, col, seeds, beta, gamma, sims, T, rng_seed, sim_offset=0):
    rp, ci = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    sd = np.ascontiguousarray(list(seeds), np.int32)
    counts = np.zeros((3, T, n), dtype=np.uint32)
    load().oracle_sir_philox(_p(rp), _p(ci), C.c_int(n), _p(sd), C.c_int(sd.shape[0]), C.c_double(beta), C.c_double(gamma),
                             C.c_long(sims), C.c_long(sim_offset), C.c_int(T), C.c_uint64(rng_seed), _p(counts))
    return counts
