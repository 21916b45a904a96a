"""CPU: the C restatement (oracle/gnode_oracle.c) against the numpy oracle, which is
itself pinned by the reference-produced golden vectors."""
import numpy as np

import gnode_oracle as O
import oracle_c as OC


def _rel(a, b):
    return np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))) / (np.max(np.abs(b)) + 1e-30)


def test_c_rhs_matches_numpy():
    rp, ci, _ = O.er_graph(300, 1500, seed=2)
    rng = np.random.default_rng(0)
    for B, H in [(1, 64), (3, 8), (2, 20)]:
        P = O.init_params(H, seed=H)
        x = rng.uniform(0, 1.5, (4 * B * 300, H)).astype(np.float32)
        x[3 * B * 300:, 0] = 0.3
        x[3 * B * 300:, 1] = 0.2
        a = OC.rhs(rp, ci, 300, x, P["odefunc.linear.weight"], P["odefunc.linear.bias"])
        b = O.rhs_single(x, P["odefunc.linear.weight"], P["odefunc.linear.bias"], rp, ci, 300)
        assert _rel(a, b) < 1e-5


def test_c_forward_matches_numpy():
    rp, ci, _ = O.er_graph(200, 900, seed=4)
    for B, H, maxTime, dT in [(2, 64, 8, 0.5), (1, 8, 5, 0.25)]:
        P = O.init_params(H, seed=1)
        x = O.make_samples(200, B, H, seed=2)
        S, I, R = OC.forward_euler(rp, ci, 200, x, P, O.step_sizes(O.time_grid(maxTime, dT)))
        So, Io, Ro = O.odeblock_forward_single(x, P, rp, ci, maxTime, dT)
        assert S.shape == So.shape
        assert max(_rel(S, So), _rel(I, Io), _rel(R, Ro)) < 1e-5


def test_c_sir_philox_bit_exact_vs_numpy():
    rp, ci, _ = O.er_graph(120, 400, seed=5)
    a = OC.sir_philox(120, rp, ci, [3, 50], 0.35, 0.25, 200, 10, rng_seed=0xDEADBEEF12345, sim_offset=7)
    b = O.sir_philox(120, rp, ci, [3, 50], 0.35, 0.25, 200, 10, rng_seed=0xDEADBEEF12345, sim_offset=7)
    assert np.array_equal(a, b)
