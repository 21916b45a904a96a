"""Child process of tests/test_gpu_rccl.py: initialise the `nccl` backend (= RCCL on ROCm) with a world of ONE
rank on cuda:0 and push the N>1 host path's collectives through it (GNODE_FORCE_COLLECTIVE=1): the int32
Monte-Carlo count all-reduce, the flat gradient all-reduce, the label generation route and one Runner epoch.
Prints one JSON line.  A one-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one device), so
this is the largest world that can run there; the 2-rank logic is covered on gloo."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gn-ode-sir_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29533")
    os.environ["GNODE_FORCE_COLLECTIVE"] = "1"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend()}
    try:
        import gnode_oracle as O
        import scipy.sparse as sp
        from gnode import sharding as sh, synth
        from gnode.graph import DeviceGraph
        from gnode.ode_nn import sir_counts
        from gnode.ode_nn_ngraph_sim import ODEBlock, ODEfunc
        from gnode.trainer import Runner
        assert sh.collectives_on()
        # barrier + MAX of a timing scalar: what bench.py does between ranks
        dist.barrier()
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out["max_ok"] = float(t.item()) == 1.25
        # Monte-Carlo counts: int32 CUDA tensor through ncclAllReduce(sum)
        n, T = 300, 8
        rp, ci, _ = O.er_graph(n, 1200, seed=1)
        g = DeviceGraph(rp, ci)
        cnt = sir_counts(g, [0, 5], 0.3, 0.2, sims=64, T=T, rng_seed=7)
        out["counts_dtype"] = str(cnt.dtype)
        before = cnt.clone()
        sh.allreduce_counts(cnt)
        torch.cuda.synchronize()
        out["counts_ok"] = bool(torch.equal(cnt, before)) and bool(
            np.array_equal(cnt.cpu().numpy().astype(np.uint32), O.sir_philox(n, rp, ci, [0, 5], 0.3, 0.2, 64, T, 7)))
        # flat gradient all-reduce
        w = torch.nn.Linear(64, 64).to(dev)
        for p_ in w.parameters():
            p_.grad = torch.full_like(p_, 3.0)
        sh.allreduce_flat_grads(list(w.parameters()), scale=0.5)
        out["grads_ok"] = all(bool(torch.all(p_.grad == 1.5)) for p_ in w.parameters())
        # one Runner epoch (broadcast of the initial weights, per-step gradient all-reduce, loss all-reduce)
        H, maxTime, deltaT, NS = 64, 6, 0.5, 4
        A = sp.csr_matrix((np.ones(ci.shape[0]), ci, rp), shape=(n, n))
        model = ODEBlock(maxTime, deltaT, n, [0], H, ODEfunc(A, 0.2, 0.1, H, dev), dev).to(dev)
        P = synth.linear_params(H, seed=1)
        model.load_state_dict({**model.state_dict(), **{k: torch.from_numpy(v) for k, v in P.items()}})
        from golden.labels import closed_form_labels
        x = torch.from_numpy(synth.samples(n, NS, H, seed=2))
        y = torch.from_numpy(closed_form_labels(NS, n, maxTime))
        xs, ys = [x[i] for i in range(NS)], [y[i] for i in range(NS)]
        run = Runner(model, 1e-2, maxTime, deltaT, dev, stack=True, use_graphs=False)
        assert run.collective
        l0, _ = run.train_epoch(xs, ys, 2, 0)
        l1, _ = run.train_epoch(xs, ys, 2, 1)
        ev, _ = run.evaluate(xs, ys, 2)
        out["runner_ok"] = bool(np.isfinite([l0, l1, ev]).all() and l1 < l0)
        out["losses"] = [l0, l1, ev]
    finally:
        dist.destroy_process_group()
    print("RCCL_WORLD1 " + json.dumps(out))


if __name__ == "__main__":
    main()
