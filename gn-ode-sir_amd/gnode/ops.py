"""Tensor-level entry points over the C ABI (torch is plumbing: memory + streams)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .graph import DeviceGraph

PARAM_KEYS = ("odefunc.linear.weight", "odefunc.linear.bias", "linearS1.weight", "linearS1.bias",
              "linear3.weight", "linear3.bias", "linearS2.weight", "linearS2.bias")

METHODS = {"euler": 0, "rk4": 1}

# Kept activations for the adjoint backward (include/gnode.h: `keep`): on unless GNODE_KEEP=0, read ONCE at import.
# Cost: 3 * (n_steps + 1) * (rows + 1) * 64 floats on top of the trajectory's 4 slabs per grid point (+75 % activation
# memory: 75k nodes x 8 samples x 59 steps = 36.9 GB of trajectory + 27.6 GB kept); when that allocation fails the
# forward falls back to the recomputing backward's layout (no keep buffer) instead of raising.
KEEP_DEFAULT = os.environ.get("GNODE_KEEP", "1") != "0"
# The persistent one-launch integration of mid-size graphs (csrc/gnode_pers64.hip; include/gnode.h: GNODE_FWD_PER_STEP):
# on unless GNODE_PERSIST=0, read once at import.  Same outputs bit for bit either way.
PERSIST_DEFAULT = os.environ.get("GNODE_PERSIST", "1") != "0"
FWD_PER_STEP = 1


def time_grid(maxTime, deltaT) -> np.ndarray:
    """float64 np.arange(0, maxTime, deltaT): reference ode_nn_ngraph_sim.py:110."""
    return np.arange(0, maxTime, deltaT)


def step_sizes(grid: np.ndarray) -> np.ndarray:
    """fp32 dt_k = t[k+1]-t[k] (a 0-dim float64 tensor times an fp32 state stays fp32)."""
    g = np.asarray(grid, dtype=np.float64)
    return np.ascontiguousarray((g[1:] - g[:-1]).astype(np.float32))


def subsample_rows(maxTime, deltaT) -> np.ndarray:
    """Grid rows get_sir_t_nodes_torch keeps: int(i/deltaT), i < maxTime (ode_nn.py:257-259)."""
    return np.asarray([int(i / deltaT) for i in range(int(maxTime))], dtype=np.int32)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise _lib.GnodeError(f"GN-ODE path is fp32 (got {t.dtype})")
    return t if t.is_contiguous() else t.contiguous()


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def pack_params(tensors: dict) -> _lib.Params:
    p = _lib.Params()
    for k in PARAM_KEYS:
        t = tensors[k]
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise _lib.GnodeError(f"parameter {k} must be a contiguous fp32 GPU tensor")
        setattr(p, k.replace(".", "_"), t.data_ptr())
    return p


def rhs(graph: DeviceGraph, x: torch.Tensor, W: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """ODEfunc.forward on x [4*rows, H] (ode_nn_ngraph_sim.py:58-96)."""
    lib = _lib.load()
    x = _f32c(x)
    graph.check_device(x)
    rows4, H = x.shape
    if rows4 % 4:
        raise _lib.GnodeError("state must have 4 slabs")
    rows = rows4 // 4
    dx = torch.empty_like(x)
    ws = _workspace(lib.gnode_rhs_workspace_bytes(graph.handle, rows, H), x.device)
    _lib.check(lib.gnode_rhs_f32(graph.handle, _lib.ptr(x), _lib.ptr(_f32c(W)), _lib.ptr(_f32c(b)), _lib.ptr(dx), rows, H,
                                 _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
    return dx


def forward(graph: DeviceGraph, x2d: torch.Tensor, params: dict, dts: np.ndarray, method: str = "euler",
            out_rows: np.ndarray | None = None, want_sol: bool = False, workspace: torch.Tensor | None = None,
            want_keep: bool | None = None, persist: bool | None = None):
    """ODEBlock.forward on x2d [rows, 3+H]; returns (S, I, R) each [n_out, rows] and sol or None.

    With want_sol (training) and want_keep, the kept activations the adjoint backward reads back (include/gnode.h: `keep`)
    ride along as ``sol.gnode_keep`` (None on paths that keep nothing); `backward` picks them up from there."""
    lib = _lib.load()
    x2d = _f32c(x2d)
    graph.check_device(x2d)
    rows, H = x2d.shape[0], x2d.shape[1] - 3
    dts = np.ascontiguousarray(dts, dtype=np.float32)
    n_steps = int(dts.shape[0])
    m = METHODS[method]
    if out_rows is not None:
        out_rows = np.ascontiguousarray(out_rows, dtype=np.int32)
        n_out = int(out_rows.shape[0])
    else:
        n_out = n_steps + 1
    dev = x2d.device
    out = torch.empty((3, n_out, rows), dtype=torch.float32, device=dev)
    sol = torch.empty((n_steps + 1, 4 * rows, H), dtype=torch.float32, device=dev) if want_sol else None
    need = lib.gnode_forward_workspace_bytes(graph.handle, rows, H, m)
    ws = workspace if (workspace is not None and workspace.numel() >= need) else _workspace(need, dev)
    p = pack_params(params)
    keep = None
    if want_keep is None:
        want_keep = KEEP_DEFAULT
    if sol is not None and want_keep and m == 0:
        kb = lib.gnode_forward_keep_bytes(graph.handle, rows, H, n_steps, n_out)
        if kb:
            try:
                keep = torch.empty(kb // 4, dtype=torch.float32, device=dev)
            except torch.cuda.OutOfMemoryError:
                keep = None                      # the recomputing backward needs only the trajectory
    info = C.c_int32(0)
    _lib.check(lib.gnode_forward_f32(
        graph.handle, _lib.ptr(x2d), C.byref(p), _lib.host_ptr(dts), n_steps, m,
        _lib.host_ptr(out_rows) if out_rows is not None else None, n_out,
        _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]), _lib.ptr(sol) if sol is not None else None,
        _lib.ptr(keep) if keep is not None else None, keep.numel() * 4 if keep is not None else 0,
        rows, H, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(),
        0 if (PERSIST_DEFAULT if persist is None else persist) else FWD_PER_STEP, C.byref(info)))
    if sol is not None:
        sol.gnode_keep = keep
        sol.gnode_info = int(info.value)         # what the call left in sol / keep: the backward checks the pairing
    # (no reference to `graph` is kept: a DeviceGraph released while some stream is capturing would hipFree inside the capture)
    forward.last_workspace = (rows, H, m, ws, forward_path(graph, rows, H, n_steps, n_out, sol is not None, method, persist)[0])
    return out[0], out[1], out[2], sol


def forward_path(graph: DeviceGraph, rows: int, H: int, n_steps: int, n_out: int | None = None, want_sol: bool = False,
                 method: str = "euler", persist: bool | None = None):
    """(path, plan): 0 = one launch per Euler step, 1 = one-workgroup launch (tiny graphs), 2 = persistent launch with
    plan = (tiles per workgroup, workgroups per sample, XCDs per sample, samples per XCD, samples alive at once)."""
    plan = (C.c_int32 * 8)()
    path = _lib.load().gnode_forward_path(graph.handle, rows, H, METHODS[method], n_steps, n_steps + 1 if n_out is None else n_out,
                                          int(want_sol), 0 if (PERSIST_DEFAULT if persist is None else persist) else FWD_PER_STEP, plan)
    return int(path), tuple(int(v) for v in plan[:5])


def forward_status() -> int:
    """0, or the give-up code of the persistent launch behind the LAST `forward` call (synchronises the stream)."""
    last = getattr(forward, "last_workspace", None)
    if last is None:
        return 0
    rows, H, m, ws, path = last
    if path not in (2, 3):
        return 0                                 # (the control block is only written by the persistent launch)
    code = C.c_int32(0)
    _lib.check(_lib.load().gnode_forward_status(rows, H, m, _lib.ptr(ws), _lib.stream_ptr(), C.byref(code)))
    return int(code.value)


def backward(graph: DeviceGraph, x2d: torch.Tensor, params: dict, dts: np.ndarray, method: str, out_rows, sol: torch.Tensor,
             gS: torch.Tensor, gI: torch.Tensor, gR: torch.Tensor, keep="auto", persist: bool | None = None) -> dict:
    """Adjoint-Euler parameter gradients (torchdiffeq odeint_adjoint semantics)
    given the saved trajectory `sol` and the upstream gradients of S, I, R ([n_out, rows]).
    keep: the forward's kept activations ("auto": ``sol.gnode_keep`` when `forward` attached it; None: recompute)."""
    if method != "euler":
        raise _lib.GnodeError("the adjoint backward is implemented for method='euler' (the reference's method)")
    lib = _lib.load()
    x2d = _f32c(x2d)
    rows, H = x2d.shape[0], x2d.shape[1] - 3
    dts = np.ascontiguousarray(dts, dtype=np.float32)
    n_steps = int(dts.shape[0])
    if out_rows is not None:
        out_rows = np.ascontiguousarray(out_rows, dtype=np.int32)
        n_out = int(out_rows.shape[0])
    else:
        n_out = n_steps + 1
    for t in (gS, gI, gR):
        if tuple(t.shape) != (n_out, rows):
            raise _lib.GnodeError(f"upstream gradient shape {tuple(t.shape)} != {(n_out, rows)}")
    grads = {k: torch.empty_like(params[k], memory_format=torch.contiguous_format) for k in PARAM_KEYS}
    ws = _workspace(lib.gnode_backward_workspace_bytes(graph.handle, rows, H), x2d.device)
    p, gp = pack_params({k: v.detach() for k, v in params.items()}), pack_params(grads)
    if isinstance(keep, str):
        keep = getattr(sol, "gnode_keep", None)
    elif keep is None and getattr(sol, "gnode_keep", None) is not None and H == 64:
        # (a trajectory produced WITH kept activations does not carry A Z_I in its 4th slabs: include/gnode.h)
        raise _lib.GnodeError("this trajectory was produced with a keep buffer: pass it (keep='auto'), or run the forward with want_keep=False")
    _lib.check(lib.gnode_backward_f32(
        graph.handle, _lib.ptr(x2d), C.byref(p), _lib.host_ptr(dts), n_steps,
        _lib.host_ptr(out_rows) if out_rows is not None else None, n_out, _lib.ptr(_f32c(sol)),
        _lib.ptr(keep) if keep is not None else None, keep.numel() * 4 if keep is not None else 0,
        _lib.ptr(_f32c(gS)), _lib.ptr(_f32c(gI)), _lib.ptr(_f32c(gR)), C.byref(gp), rows, H,
        _lib.ptr(ws), ws.numel(), _lib.stream_ptr(), 0 if (PERSIST_DEFAULT if persist is None else persist) else FWD_PER_STEP,
        int(getattr(sol, "gnode_info", -1))))
    backward.last_workspace = (rows, H, ws)
    return grads


def backward_status() -> int:
    """0, or the give-up code of the persistent adjoint sweep behind the LAST `backward` call (synchronises the stream)."""
    last = getattr(backward, "last_workspace", None)
    if last is None:
        return 0
    rows, H, ws = last
    code = C.c_int32(0)
    _lib.check(_lib.load().gnode_backward_status(rows, H, _lib.ptr(ws), _lib.stream_ptr(), C.byref(code)))
    return int(code.value)


def l1_loss_sum(S: torch.Tensor, I: torch.Tensor, R: torch.Tensor, y: torch.Tensor, t0: int = 1, want_sign: bool = True, sign_scale: float = 1.0):
    """sum over rows, t >= t0, c of |pred_c[t, row] - y[row, t, c]| as a float64 device scalar, and (want_sign)
    sign(pred - y) * sign_scale as fp32 [3, T, rows] -- the loss of ode_nn_ngraph_sim.py:230-234 and its gradient in one launch
    (sign_scale = 1 / element count gives L1Loss's mean gradient as it stands)."""
    lib = _lib.load()
    S, I, R = (_f32c(t.detach()) for t in (S, I, R))
    T = int(S.shape[0])
    rows = S.numel() // max(T, 1)
    if y.dtype not in (torch.float32, torch.float64):
        y = y.to(torch.float32)
    y = y.detach().contiguous()
    if tuple(y.shape) != (rows, T, 3) or any(t.numel() != T * rows for t in (I, R)):
        raise _lib.GnodeError(f"l1_loss_sum: outputs {tuple(S.shape)} vs labels {tuple(y.shape)}")
    dev = S.device
    total = torch.empty((), dtype=torch.float64, device=dev)
    sgn = torch.empty((3, T, rows), dtype=torch.float32, device=dev) if want_sign else None
    ws = _workspace(lib.gnode_l1_loss_workspace_bytes(), dev)
    _lib.check(lib.gnode_l1_loss_scaled_f32(_lib.ptr(S), _lib.ptr(I), _lib.ptr(R), _lib.ptr(y), int(y.dtype == torch.float64), rows, T,
                                            int(t0), _lib.ptr(total), _lib.ptr(sgn) if sgn is not None else None, float(sign_scale),
                                            _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
    return total, sgn
