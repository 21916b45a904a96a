"""Build libgnode_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(PKG), "csrc")
ROOT = os.path.dirname(os.path.dirname(PKG))
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libgnode_hip.so")
SOURCES = ["gnode_ode.hip", "gnode_h64.hip", "gnode_pers64.hip", "gnode_pers64_bwd.hip", "gnode_persg.hip", "gnode_h128.hip", "gnode_bwd.hip", "gnode_bwd_tiny.hip", "gnode_hub.hip", "gnode_sir.hip", "gnode_dmp.hip", "gnode_meanfield.hip", "gnode_loss.hip"]
# -ffp-contract=off: the SIR update must round like the reference's separate torch ops
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + INCLUDE, "-I" + CSRC]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libgnode_hip.so)")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "gnode.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    hipcc = _hipcc()
    from concurrent.futures import ThreadPoolExecutor

    def compile_one(src):
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *os.environ.get("GNODE_EXTRA_FLAGS", "").split(), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:   # one hipcc per translation unit
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
