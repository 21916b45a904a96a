import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _fresh_library(request):
    """GPU sessions run against the CURRENT sources: rebuild libgnode_hip.so if it is stale (hipcc is on
    the GPU box too).  A failed build fails the session loudly; there is no fallback path to fall back to."""
    if request.config.getoption("-m") and "not gpu" in request.config.getoption("-m"):
        return
    try:
        import torch
        if not torch.cuda.is_available():
            return
    except Exception:
        return
    from gnode.build import build_lib
    build_lib()
