"""Drop-in for the model classes of reference ode_nn_ngraph_sim.py (the script
`model='ode_nn'` resolves to in monitorer-sim.py:26-33): same constructor
signatures, same state_dict key names, same forward shapes -- the arithmetic runs
in libgnode_hip.so on the MI355X.

    ODEfunc(A, beta, gamma, hidden1, device)            reference :37-96
    ODEBlock(maxTime, deltaT, n_nodes, indices, hidden1, odefunc, device)   :99-188
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .graph import DeviceGraph


class ODEfunc(nn.Module):
    def __init__(self, A, beta, gamma, hidden1, device):
        super().__init__()
        self.A = A
        self.beta = beta          # unused by forward, as in the reference (:42-43)
        self.gamma = gamma
        self.ln = nn.LayerNorm(hidden1)            # never applied in the reference (:94-95); kept for state_dict parity
        self.linear = nn.Linear(hidden1, hidden1)
        self.graph = DeviceGraph.from_scipy(A)     # CSR goes to HBM once, not once per RHS (:68-71)

    def init_weights(self):
        """reference :54-56 (defined, never called: the default nn.Linear init is what trains)."""
        nn.init.xavier_normal_(self.linear.weight)

    def forward(self, t, x):
        """x [4*B*n, H] -> dx (reference :58-96).  t is unused there too."""
        with torch.no_grad():
            return ops.rhs(self.graph, x, self.linear.weight, self.linear.bias)


class ODEBlock(nn.Module):
    def __init__(self, maxTime, deltaT, n_nodes, indices, hidden1, odefunc, device, method="euler"):
        super().__init__()
        self.maxTime = maxTime
        self.deltaT = deltaT
        self.device = device
        self.method = method                        # the reference hard-codes 'euler' (:168)
        self.integration_time = torch.from_numpy(ops.time_grid(maxTime, deltaT))
        self._dts = ops.step_sizes(ops.time_grid(maxTime, deltaT))
        self.odefunc = odefunc
        self.n_nodes = n_nodes
        self.indices = torch.tensor(indices, requires_grad=False)
        self.hidden1 = hidden1
        self.linearS1 = nn.Linear(1, hidden1)
        self.ln = nn.LayerNorm(hidden1)             # unused in the reference forward
        self.linear3 = nn.Linear(hidden1, 4)
        self.linearS2 = nn.Linear(4, 1)

    def init_weights(self):
        """reference :139-146 (defined, never called)."""
        nn.init.kaiming_normal_(self.linearS1.weight, mode="fan_in", nonlinearity="relu")
        nn.init.kaiming_normal_(self.linear3.weight, mode="fan_in", nonlinearity="relu")
        self.linearS2.weight.data.normal_(0, 1)

    def _params(self):
        return {"odefunc.linear.weight": self.odefunc.linear.weight, "odefunc.linear.bias": self.odefunc.linear.bias,
                "linearS1.weight": self.linearS1.weight, "linearS1.bias": self.linearS1.bias,
                "linear3.weight": self.linear3.weight, "linear3.bias": self.linear3.bias,
                "linearS2.weight": self.linearS2.weight, "linearS2.bias": self.linearS2.bias}

    def forward(self, x, out_rows=None):
        """x [B, n, 3+H] -> (S, I, R), each [G, B*n, 1] (reference :148-188).

        out_rows (extension): ascending grid indices to emit instead of all G points;
        `ops.subsample_rows(maxTime, deltaT)` fuses get_sir_t_nodes_torch (ode_nn.py:249-261).
        """
        x2d = x.reshape(-1, x.size(-1))
        from .autograd import forward_with_grad
        S, I, R = forward_with_grad(self.odefunc.graph, x2d, self._params(), self._dts, self.method, out_rows)
        return S.unsqueeze(-1), I.unsqueeze(-1), R.unsqueeze(-1)
