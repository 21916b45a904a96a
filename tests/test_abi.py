"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/gnode.h declares.  No compute call is made (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from gnode.build import build_lib
    from gnode import _lib
    build_lib()
    return _lib.load()


def _declared():
    src = open(os.path.join(ROOT, "include", "gnode.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gnode_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(lib):
    names = _declared()
    assert len(names) >= 12
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/gnode.h but not exported: {missing}"


def test_binding_table_matches_header(lib):
    from gnode import _lib
    assert sorted(_lib.EXPORTS) == _declared()


def test_version_and_sizes(lib):
    assert lib.gnode_version() >= 210
    # workspace sizes depend on the graph (hub scratch): without a handle they answer 0 instead of guessing
    # (real sizes are exercised on the GPU: tests/test_gpu_abi.py)
    assert lib.gnode_rhs_workspace_bytes(None, 1000, 64) == 0
    assert lib.gnode_forward_workspace_bytes(None, 1000, 64, 0) == 0
    assert lib.gnode_backward_workspace_bytes(None, 1000, 64) == 0
    assert lib.gnode_forward_keep_bytes(None, 1000, 64, 59, 60) == 0        # nothing is kept without a graph either
    assert lib.gnode_l1_loss_workspace_bytes() >= 8                          # graph-independent: one double per workgroup


def test_product_path_has_no_cpu_fallback():
    """Tensors on the CPU are refused, not silently computed somewhere else."""
    import torch
    from gnode import _lib
    with pytest.raises(_lib.GnodeError):
        _lib.ptr(torch.zeros(4))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "gn-ode-sir_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "gnode_oracle" not in txt and "oracle_c" not in txt and "liboracle" not in txt, f
