"""ctypes binding of libgnode_hip.so (the C ABI declared in include/gnode.h).

There is no CPU fallback: if the HIP library is missing or a tensor is not on a
GPU the calls raise.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgnode_hip.so")

EXPORTS = [
    "gnode_last_error", "gnode_version",
    "gnode_graph_create", "gnode_graph_destroy", "gnode_graph_info",
    "gnode_rhs_workspace_bytes", "gnode_rhs_f32",
    "gnode_forward_workspace_bytes", "gnode_forward_f32", "gnode_forward_status", "gnode_backward_status", "gnode_forward_path", "gnode_sol_carries_neighbour_sums", "gnode_forward_keep_bytes",
    "gnode_backward_workspace_bytes", "gnode_backward_f32",
    "gnode_sir_workspace_bytes", "gnode_sir_coins_workspace_bytes",
    "gnode_sir_mc_philox", "gnode_sir_mc_philox_scan", "gnode_sir_mc_philox_counted", "gnode_sir_mc_coins",
    "gnode_dmp_workspace_bytes", "gnode_dmp_f32",
    "gnode_meanfield_workspace_bytes", "gnode_meanfield_f64",
    "gnode_l1_loss_workspace_bytes", "gnode_l1_loss_f32", "gnode_l1_loss_scaled_f32",
    "gnode_profile_enable", "gnode_profile_read", "gnode_profile_read_kind",
]


class GnodeError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "odefunc_linear_weight", "odefunc_linear_bias", "linearS1_weight", "linearS1_bias",
        "linear3_weight", "linear3_bias", "linearS2_weight", "linearS2_bias")]


class Grads(C.Structure):
    _fields_ = Params._fields_


_lib = None


def load():
    """Load the shared library (building nothing: see gnode.build.build_lib)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GnodeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the GN-ODE path.")
    import torch  # noqa: F401  -- first, so that one HIP runtime (torch's libamdhip64.so.7) serves both
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, sz, f32p = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.POINTER(C.c_float)
    lib.gnode_last_error.restype = C.c_char_p
    lib.gnode_version.restype = C.c_int
    lib.gnode_graph_create.argtypes = [vp, vp, i32, i64, C.POINTER(vp)]
    lib.gnode_graph_destroy.argtypes = [vp]
    lib.gnode_graph_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i64), C.POINTER(i32)]
    lib.gnode_rhs_workspace_bytes.argtypes = [vp, i64, i32]
    lib.gnode_rhs_workspace_bytes.restype = sz
    lib.gnode_rhs_f32.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp, sz, vp]
    lib.gnode_forward_workspace_bytes.argtypes = [vp, i64, i32, i32]
    lib.gnode_forward_workspace_bytes.restype = sz
    lib.gnode_sol_carries_neighbour_sums.argtypes = [vp, i64, i32, i32, i32, i32]
    lib.gnode_sol_carries_neighbour_sums.restype = C.c_int
    lib.gnode_forward_keep_bytes.argtypes = [vp, i64, i32, i32, i32]
    lib.gnode_forward_keep_bytes.restype = sz
    lib.gnode_forward_f32.argtypes = [vp, vp, C.POINTER(Params), vp, i32, i32, vp, i32, vp, vp, vp, vp, vp, sz, i64, i32, vp, sz, vp, i32, C.POINTER(i32)]
    lib.gnode_forward_status.argtypes = [i64, i32, i32, vp, vp, C.POINTER(i32)]
    lib.gnode_forward_status.restype = C.c_int
    lib.gnode_backward_status.argtypes = [i64, i32, vp, vp, C.POINTER(i32)]
    lib.gnode_backward_status.restype = C.c_int
    lib.gnode_forward_path.argtypes = [vp, i64, i32, i32, i32, i32, i32, i32, C.POINTER(i32)]
    lib.gnode_forward_path.restype = C.c_int
    lib.gnode_meanfield_workspace_bytes.argtypes = [vp]
    lib.gnode_meanfield_workspace_bytes.restype = sz
    lib.gnode_meanfield_f64.argtypes = [vp, vp, i32, C.c_double, vp, vp, i32, C.c_double, C.c_double, vp, vp, vp,
                                        C.POINTER(C.c_int64), vp, sz, vp]
    lib.gnode_dmp_workspace_bytes.argtypes = [vp]
    lib.gnode_dmp_workspace_bytes.restype = sz
    lib.gnode_dmp_f32.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, sz, vp]
    lib.gnode_l1_loss_workspace_bytes.argtypes = []
    lib.gnode_l1_loss_workspace_bytes.restype = sz
    lib.gnode_l1_loss_f32.argtypes = [vp, vp, vp, vp, i32, i64, i32, i32, vp, vp, vp, sz, vp]
    lib.gnode_l1_loss_f32.restype = C.c_int
    lib.gnode_l1_loss_scaled_f32.argtypes = [vp, vp, vp, vp, i32, i64, i32, i32, vp, vp, C.c_float, vp, sz, vp]
    lib.gnode_l1_loss_scaled_f32.restype = C.c_int
    lib.gnode_backward_workspace_bytes.argtypes = [vp, i64, i32]
    lib.gnode_backward_workspace_bytes.restype = sz
    lib.gnode_backward_f32.argtypes = [vp, vp, C.POINTER(Params), vp, i32, vp, i32, vp, vp, sz, vp, vp, vp,
                                       C.POINTER(Params), i64, i32, vp, sz, vp, i32, i32]
    lib.gnode_backward_f32.restype = C.c_int
    lib.gnode_sir_workspace_bytes.argtypes = [vp, i32]
    lib.gnode_sir_workspace_bytes.restype = sz
    lib.gnode_sir_coins_workspace_bytes.restype = sz
    lib.gnode_sir_mc_philox.argtypes = [vp, vp, i32, C.c_double, C.c_double, i64, i64, i32, C.c_uint64, vp, vp, sz, vp]
    lib.gnode_sir_mc_philox_scan.argtypes = lib.gnode_sir_mc_philox.argtypes
    lib.gnode_sir_mc_philox_scan.restype = C.c_int
    lib.gnode_sir_mc_philox_counted.argtypes = lib.gnode_sir_mc_philox.argtypes + [C.POINTER(C.c_uint64)]
    lib.gnode_sir_mc_philox_counted.restype = C.c_int
    lib.gnode_sir_mc_coins.argtypes = [vp, vp, i64, i32, vp, i32, C.c_double, C.c_double, i64, i32, vp, i64, vp,
                                       C.POINTER(i64), vp, sz, vp]
    lib.gnode_profile_enable.argtypes = [C.c_int]
    lib.gnode_profile_read.argtypes = [C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(i64)]
    lib.gnode_profile_read_kind.argtypes = [i32, C.POINTER(C.c_double), C.POINTER(i64)]
    for fn in ("gnode_graph_create", "gnode_graph_destroy", "gnode_graph_info", "gnode_rhs_f32", "gnode_forward_f32",
               "gnode_sir_mc_philox", "gnode_sir_mc_coins"):
        getattr(lib, fn).restype = C.c_int
    if lib.gnode_version() < 221:
        raise GnodeError(f"{LIB_PATH} is stale (ABI {lib.gnode_version()} < 221): rebuild it (gnode.build.build_lib)")
    _lib = lib
    return lib


def check(status: int):
    if status != 0:
        raise GnodeError(f"libgnode_hip status {status}: {load().gnode_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor that must already live on the GPU, contiguous."""
    if t is None:
        return None
    if not t.is_cuda:
        raise GnodeError("GN-ODE path: tensor is not on a GPU (no CPU fallback exists)")
    if not t.is_contiguous():
        raise GnodeError("GN-ODE path: tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def host_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
