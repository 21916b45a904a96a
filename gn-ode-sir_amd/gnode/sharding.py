"""Multi-GPU sharding of the path: independent units, one process per GPU.

* forward / inference: (graph, beta, gamma, seed-set) samples are split into contiguous
  per-rank blocks; the graph CSR is replicated; NO data-path collective (results are
  gathered by the caller only if it wants them in one place).
* Monte-Carlo labels: the `sims` range is split; coins are keyed by the GLOBAL sim
  index, so the summed counts equal the single-GPU counts bit for bit.  One
  all-reduce(sum) of the uint32 [3,T,n] counts per sample (RCCL over xGMI on GPUs).
* training: local backward, one flat all-reduce of the 4 809-float gradient.
Pure host logic + torch.distributed; works with the gloo backend on CPU tensors (tests)
and with nccl (= RCCL) on GPU tensors.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int):
    """Contiguous [lo, hi) block of `total` units for `rank`; sizes differ by at most one."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env():
    """One process per GPU under torchrun: bind this rank to cuda:LOCAL_RANK and join the RCCL
    (backend "nccl") process group.  No-op for a plain single-process run.  Returns (rank, world)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
            backend = os.environ.get("GNODE_DIST_BACKEND", "nccl")
            if int(os.environ.get("LOCAL_WORLD_SIZE", world)) > torch.cuda.device_count():
                share_device_guard()
        else:
            backend = "gloo"
        dist.init_process_group(backend)
    return world_info()


def share_device_guard():
    """Rehearsals that put several ranks on ONE GPU: the persistent one-launch kernels (csrc/gnode_pers64*.hip) need all of
    their workgroups resident at once, one per CU, and two processes' launches could each hold half of the CUs while
    waiting for the other half (their spins are bounded: a give-up code, not a hang).  One process per GPU is the
    deployment; a shared device runs the one-launch-per-step forms."""
    from . import ops
    ops.PERSIST_DEFAULT = False


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def collectives_on() -> bool:
    """True when the data-path collectives must run: a process group exists and either it has more than one rank
    or GNODE_FORCE_COLLECTIVE=1 (single-GPU rehearsal of the RCCL calls: a world of one still goes through
    ncclAllReduce, so the backend, the dtypes and the stream handling are exercised on a one-GPU box)."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("GNODE_FORCE_COLLECTIVE", "0") == "1"


def allreduce_counts(counts: torch.Tensor) -> torch.Tensor:
    """Sum Monte-Carlo count shards [3,T,n] over ranks (int32 holding uint32 bit patterns: two's-complement
    addition is the same operation).  Row 0 is ASSIGNED, not accumulated (reference quirk, ode_nn.py:55-56):
    only rank 0 -- which owns trajectory 0 whenever there is any trajectory at all (shard_range hands the
    remainder to the low ranks) -- contributes it, so a rank whose shard is empty cannot disturb it."""
    if not collectives_on():
        return counts
    if dist.get_rank() != 0:
        counts[:, 0] = 0
    work = counts.to(torch.int64) if counts.dtype == torch.int32 and not counts.is_cuda else counts   # gloo has no int32 sum on some builds
    dist.all_reduce(work, op=dist.ReduceOp.SUM)
    if work is not counts:
        counts.copy_(work.to(counts.dtype))
    return counts


def allreduce_flat_grads(params, scale: float = 1.0):
    """One flat-buffer all-reduce(sum) of every parameter gradient (19 KB for H=64), then
    scale (e.g. 1/global element count to keep the reference's element-mean L1 semantics,
    ode_nn_ngraph_sim.py:248-249)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    on = collectives_on()
    if not on and scale == 1.0:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    if on:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if scale != 1.0:
        flat.mul_(scale)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def gather_rows(local: torch.Tensor, sizes):
    """all_gather of per-rank row blocks of unequal size along dim 1 ([G, rows_r] each)."""
    rank, world = world_info()
    if world == 1:
        return local
    mx = max(sizes)
    pad = torch.zeros(local.shape[0], mx, dtype=local.dtype, device=local.device)
    pad[:, :local.shape[1]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:, :s] for b, s in zip(bufs, sizes)], dim=1)


def sharded_forward(fn, x: torch.Tensor, gather: bool = True):
    """BASELINE configs[3]'s pattern (inference over a batch of (beta, gamma, seed-set) samples on one graph): every rank
    holds the whole batch x [B, n, ...] (or at least its own block), runs `fn` on its contiguous block of samples and --
    `gather` -- every rank receives the full [G, B*n] outputs in sample order (ranks may own unequal blocks, or none).
    fn(x_block) -> tuple of [G, rows_block] tensors.  No collective touches the data path itself."""
    rank, world = world_info()
    B, n = int(x.shape[0]), int(x.shape[1])
    lo, hi = shard_range(B, rank, world)
    outs = None
    if hi > lo:
        outs = tuple(fn(x[lo:hi]))
    if world == 1 or not gather:
        return outs
    sizes = [(shard_range(B, r, world)[1] - shard_range(B, r, world)[0]) * n for r in range(world)]
    # a rank with an empty block still takes part in the gathers: it needs the outputs' leading extent and dtype
    meta = torch.tensor([len(outs), outs[0].shape[0]] if outs else [0, 0], dtype=torch.int64, device=x.device if x.is_cuda and dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    k, G = int(meta[0]), int(meta[1])
    if outs is None:
        outs = tuple(torch.zeros(G, 0, dtype=torch.float32, device=x.device) for _ in range(k))
    return tuple(gather_rows(o, sizes) for o in outs)
