#!/usr/bin/env python3
"""Golden vectors for the mean-field baseline: the reference's own `runge_kutta_order4(sir, ...)`
(ode_nn.py:214-233) executed unchanged in the build container -- scipy (LSODA) is installed, so this result is
pinned by the reference itself.  Import shims as in make_golden.py (ndlib / torchdiffeq, never called here).
No pickle of the reference is loaded."""
import os
import sys

import numpy as np
import networkx as nx

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def main():
    MG._install_import_shims()
    sys.path.insert(0, "/root/reference")
    import ode_nn as REF
    er = nx.gnm_random_graph(150, 700, seed=11)
    cases = [("karate", nx.karate_club_graph(), [0, 33], 0.08, 0.25, 1, 20),
             ("er150", er, [3], 0.05, 0.4, 0.5, 15)]
    for name, G, seeds, beta, gamma, deltaT, maxTime in cases:
        A = np.asarray(nx.adjacency_matrix(G, nodelist=sorted(G.nodes())).todense(), dtype=np.float64)
        A[A != 0] = 1.0
        I_t, S_t, R_t = REF.runge_kutta_order4(REF.sir, A, A.shape[0], list(seeds), beta, gamma, deltaT, maxTime)
        import scipy.sparse as sp
        Ac = sp.csr_matrix(A); Ac.sort_indices()
        np.savez_compressed(os.path.join(HERE, f"meanfield_{name}.npz"), rowptr=Ac.indptr.astype(np.int32),
                            col=Ac.indices.astype(np.int32), seeds=np.asarray(seeds, np.int32), beta=np.float64(beta),
                            gamma=np.float64(gamma), deltaT=np.float64(deltaT), maxTime=np.int32(maxTime),
                            I=np.asarray(I_t), S=np.asarray(S_t), R=np.asarray(R_t))
        print(name, np.asarray(I_t).shape, float(np.asarray(R_t)[-1].mean()))


if __name__ == "__main__":
    main()
