"""Device-resident CSR adjacency (replaces the per-RHS scipy block_diag + H2D copy of
reference ode_nn_ngraph_sim.py:68-71) and the graph ingest of ode_nn.py:394-414."""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib


def csr_arrays(A):
    """scipy sparse adjacency -> (rowptr int32, col int32): sorted columns, duplicates
    merged, values dropped (the reference only uses .row/.col)."""
    a = sp.csr_matrix(A)
    a.sum_duplicates()
    a.sort_indices()
    if a.shape[0] != a.shape[1]:
        raise ValueError("adjacency must be square")
    return np.ascontiguousarray(a.indptr, dtype=np.int32), np.ascontiguousarray(a.indices, dtype=np.int32)


def concat_csr(csrs):
    """Block-diagonal CSR of several graphs (ode_nn_ngraphs.py:65-69), built ONCE per
    batch composition instead of once per RHS call."""
    rps, cis, off, nnz = [np.zeros(1, dtype=np.int64)], [], 0, 0
    for rp, ci in csrs:
        rps.append(rp[1:].astype(np.int64) + nnz)
        cis.append(ci.astype(np.int64) + off)
        nnz += ci.shape[0]
        off += rp.shape[0] - 1
    col = np.concatenate(cis) if cis else np.zeros(0, dtype=np.int64)
    return np.concatenate(rps).astype(np.int32), col.astype(np.int32)


class DeviceGraph:
    """Owns a gnode_graph_t (device CSR)."""

    def __init__(self, rowptr: np.ndarray, col: np.ndarray):
        lib = _lib.load()
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.n = int(self.rowptr.shape[0] - 1)
        self.nnz = int(self.col.shape[0])
        h = C.c_void_p()
        import torch
        self.device_index = torch.cuda.current_device() if torch.cuda.is_available() else -1   # the CSR lives on THIS GPU
        _lib.check(lib.gnode_graph_create(_lib.host_ptr(self.rowptr), _lib.host_ptr(self.col), self.n, self.nnz, C.byref(h)))
        self.handle = h

    def check_device(self, tensor):
        """One process drives one GPU: refuse tensors that live on another device than the graph."""
        if tensor.is_cuda and tensor.device.index != self.device_index:
            raise _lib.GnodeError(f"graph lives on cuda:{self.device_index}, tensor on {tensor.device}")

    @classmethod
    def from_scipy(cls, A):
        return cls(*csr_arrays(A))

    # the device CSR is immutable: copies of a model share it, pickles rebuild it from the host arrays
    def __deepcopy__(self, memo):
        return self

    def __reduce__(self):
        return (DeviceGraph, (self.rowptr, self.col))

    def __del__(self):
        try:
            h = getattr(self, "handle", None)
            if h is not None and _lib is not None and _lib._lib is not None:
                _lib._lib.gnode_graph_destroy(h)
                self.handle = None
        except Exception:       # interpreter shutdown: module globals may already be gone
            pass
