"""gnode: the MI355X-native GN-ODE integration path behind the reference's
`model='ode_nn'` call surface (sissykosm/GN-ODE-SIR).

Sub-modules mirror the reference's file names so existing imports keep working:
    gnode.ode_nn_ngraph_sim   ODEfunc, ODEBlock   (single graph, batched samples)
    gnode.ode_nn_ngraphs      ODEfunc, ODEBlock   (multi-graph batches)
    gnode.ode_nn              sir_torch, get_sir_t_nodes_torch, create_graph
The arithmetic lives in libgnode_hip.so (include/gnode.h); there is no CPU fallback.
"""
from . import _lib  # noqa: F401
from ._lib import GnodeError, LIB_PATH  # noqa: F401
from .graph import DeviceGraph, concat_csr, csr_arrays  # noqa: F401

__all__ = ["GnodeError", "LIB_PATH", "DeviceGraph", "concat_csr", "csr_arrays"]
