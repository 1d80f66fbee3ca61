"""Data parallelism for the pose-refinement path (new in this build).

Poses shard naturally: every pose owns its parameters, Adam state and loss terms; the inner loop
needs NO collective (the MSE normaliser uses the GLOBAL batch, a constant).  The only exchange is
one sum-all-reduce of the shared-parameter gradients at each shared-parameter step: the
J_regressor gradient (17 x 6890 fp32 = 468 520 B) and, at the outer step, the discriminator
gradients.  One process per GPU; backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" for CPU tests.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


def init(backend: Optional[str] = None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1)."""
    import torch.distributed as dist
    rank, local_rank, world = env_rank_world()
    if world == 1:
        return None
    if not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':     # RCCL: one rank per GPU
            torch.cuda.set_device(local_rank)
            kw['device_id'] = torch.device('cuda', local_rank)
        dist.init_process_group(backend, **kw)
    return dist


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous batch shard [lo, hi) of rank `rank` (sizes differ by at most one)."""
    return n * rank // world, n * (rank + 1) // world


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def flat_all_reduce_sum_(tensors) -> None:
    """One collective for several gradient tensors (flat bucket), written back in place."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n


def shared_adam_step(param: torch.Tensor, local_grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int,
                     lr: float, adam_fn) -> None:
    """Replicated shared-parameter step: all-reduce the local gradient (each rank's gradient is
    already normalised by the GLOBAL batch, so the sum equals the single-process gradient), then
    apply the identical Adam update on every rank.  `adam_fn(p, g, m, v, step, lr)` updates in place."""
    all_reduce_sum_(local_grad)
    adam_fn(param, local_grad, m, v, step, lr)
