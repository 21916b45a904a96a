"""Drop-in for `DMP_SIR` of reference dmp.py:74-170 (the `model='dmp'` comparison column):
same constructor and `run(seed_list, maxTime)`, the message passing runs in libgnode_hip.so.

    DMP_SIR(weight_adj, nodes_gamma)        weight_adj: scipy sparse / dense [n, n] (the reference passes A*beta)
        .run(seed_list, maxTime) -> float32 tensor [maxTime, n, 3] = (Ps, Pi, Pr) on the GPU
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib
from .graph import DeviceGraph


class DMP_SIR:
    def __init__(self, weight_adj, nodes_gamma, device="cuda"):
        A = sp.csr_matrix(weight_adj)
        A.sort_indices()
        self.N = A.shape[0]
        self.E = int(A.nnz)
        self.graph = DeviceGraph(A.indptr.astype(np.int32), A.indices.astype(np.int32))
        self.weights = torch.from_numpy(np.ascontiguousarray(A.data, dtype=np.float32)).to(device)
        self.nodes_gamma = torch.as_tensor(np.asarray(nodes_gamma, dtype=np.float32)).to(device)
        if self.nodes_gamma.numel() != self.N:
            raise _lib.GnodeError(f"DMP_SIR: nodes_gamma has {self.nodes_gamma.numel()} entries for {self.N} nodes")
        self.marginals = None

    def run(self, seed_list, maxTime):
        lib = _lib.load()
        seeds = np.ascontiguousarray(list(seed_list), dtype=np.int32)
        out = torch.empty((int(maxTime), self.N, 3), dtype=torch.float32, device=self.weights.device)
        ws = torch.empty(lib.gnode_dmp_workspace_bytes(self.graph.handle), dtype=torch.uint8, device=out.device)
        _lib.check(lib.gnode_dmp_f32(self.graph.handle, _lib.ptr(self.weights), _lib.ptr(self.nodes_gamma), _lib.host_ptr(seeds),
                                     int(seeds.shape[0]), int(maxTime), _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                     _lib.stream_ptr()))
        self.marginals = out
        return out

    def output(self):
        return self.marginals
