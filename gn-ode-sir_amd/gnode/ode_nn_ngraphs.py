"""Drop-in for the model classes of reference ode_nn_ngraphs.py (the script
`model='ode_nn'` resolves to in monitorer-ngraphs.py:25-30).

    ODEfunc(A_list, hidden1, device)                    reference :37-83
    ODEBlock(maxTime, deltaT, hidden1, odefunc, device) reference :86-152

A batch concatenates samples of different graphs along the node axis
(:179-196); the graph of each sample is named by a marker `graph_idx+1` in
column 2 of the beta-gamma slab at the sample's first node (:55, :333).  The
reference re-reads that marker on the host and rebuilds a scipy block_diag on
EVERY RHS call (:65-71); here it is read once per forward and the concatenated
CSR is cached on the GPU per batch composition (LRU, bounded; the trainer keeps
the reference's fixed batches -- shuffled once, ode_nn_ngraphs.py:359 -- so a
training run touches one composition per batch, not one per batch per epoch).
"""
from __future__ import annotations

import collections

import torch
import torch.nn as nn

from . import ops
from .graph import DeviceGraph, concat_csr, csr_arrays


class ODEfunc(nn.Module):
    def __init__(self, A_list, hidden1, device):
        super().__init__()
        self.A_list = A_list
        self.ln = nn.LayerNorm(hidden1)             # unused in the reference forward; state_dict parity
        self.linear = nn.Linear(hidden1, hidden1)
        self._csr = [csr_arrays(A) for A in A_list]
        self._cache = collections.OrderedDict()      # batch composition -> DeviceGraph, least recently used first
        self.cache_limit = 256                       # evicted handles are destroyed (their HBM goes back) when unreferenced

    def init_weights(self):
        """reference :57-58 (defined, never called)."""
        self.linear.weight.data.normal_(0, 1)

    def graph_for(self, marker: torch.Tensor) -> DeviceGraph:
        """marker = x[3,:,2] (or x[:,5] of the 2-D input): one host sync per forward."""
        nz = torch.nonzero(marker).flatten()
        return self.graph_for_picks(tuple(int(v) - 1 for v in marker[nz].to(torch.int64).tolist()))

    def graph_for_picks(self, picks) -> DeviceGraph:
        """The concatenated graph of a batch whose samples sit on graphs `picks` (indices into A_list), in order: what
        graph_for reads off the markers, for callers that already know the composition (the trainer reads each sample's
        marker ONCE: its batches are fixed, and a host read-back per forward keeps the host from running ahead)."""
        picks = tuple(int(v) for v in picks)
        g = self._cache.get(picks)
        if g is None:
            g = DeviceGraph(*concat_csr([self._csr[p] for p in picks]))
            self._cache[picks] = g
            while len(self._cache) > self.cache_limit:
                self._cache.popitem(last=False)
        else:
            self._cache.move_to_end(picks)
        return g

    def forward(self, t, x):
        """x [4, sumN, H] -> dx [4, sumN, H] (reference :54-83)."""
        g = self.graph_for(x[3, :, 2])
        if g.n != x.size(1):
            raise ValueError(f"markers describe {g.n} nodes but the state has {x.size(1)}")
        with torch.no_grad():
            flat = x.reshape(4 * x.size(1), x.size(2))
            return ops.rhs(g, flat, self.linear.weight, self.linear.bias).view_as(x)


class ODEBlock(nn.Module):
    def __init__(self, maxTime, deltaT, hidden1, odefunc, device, method="euler"):
        super().__init__()
        self.maxTime = maxTime
        self.deltaT = deltaT
        self.device = device
        self.method = method
        self.integration_time = torch.from_numpy(ops.time_grid(maxTime, deltaT))
        self._dts = ops.step_sizes(ops.time_grid(maxTime, deltaT))
        self.odefunc = odefunc
        self.hidden1 = hidden1
        self.linearS1 = nn.Linear(1, hidden1)
        self.ln = nn.LayerNorm(hidden1)
        self.linear3 = nn.Linear(hidden1, 4)
        self.linearS2 = nn.Linear(4, 1)

    def init_weights(self):
        """reference :121-122 (defined, never called)."""
        self.linearS1.weight.data.normal_(0, 1)

    def _params(self):
        return {"odefunc.linear.weight": self.odefunc.linear.weight, "odefunc.linear.bias": self.odefunc.linear.bias,
                "linearS1.weight": self.linearS1.weight, "linearS1.bias": self.linearS1.bias,
                "linear3.weight": self.linear3.weight, "linear3.bias": self.linear3.bias,
                "linearS2.weight": self.linearS2.weight, "linearS2.bias": self.linearS2.bias}

    def forward(self, x, out_rows=None, picks=None):
        """x [sumN, 3+H] -> (S, I, R), each [G, sumN, 1] (reference :124-152).  picks: the batch's graph indices when the
        caller knows them (else they are read off the markers: one host sync)."""
        g = self.odefunc.graph_for_picks(picks) if picks is not None else self.odefunc.graph_for(x[:, 3 + 2])
        if g.n != x.size(0):
            raise ValueError(f"markers describe {g.n} nodes but the batch has {x.size(0)}")
        from .autograd import forward_with_grad
        S, I, R = forward_with_grad(g, x, self._params(), self._dts, self.method, out_rows)
        return S.unsqueeze(-1), I.unsqueeze(-1), R.unsqueeze(-1)
