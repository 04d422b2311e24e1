"""Data-parallel plumbing for the flow path: samples are independent, weights are
replicated, so the batch (or the draws) are sharded over one process per GPU and the only exchange is
the all-reduce of (sum nll, count) -- 16 bytes over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the
CPU tests).  The reference is single-device (no counterpart); the loss it computes on one device is
`sum(nll) / count` (experiments/train_lean_npe.py:108-127), which this reproduces across ranks."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [lo, hi) of n items for `rank` (first n % world ranks get one more)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def global_mean_nll(nll_local: torch.Tensor, group=None) -> torch.Tensor:
    """Mean NLL over all ranks' samples: all_reduce(SUM) of (sum, count) in fp64."""
    red = torch.stack([nll_local.sum(dtype=torch.float64),
                       torch.tensor(float(nll_local.numel()), dtype=torch.float64, device=nll_local.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(red, op=dist.ReduceOp.SUM, group=group)
    return red[0] / red[1]


def rank_generator(seed: int, rank: int, device) -> torch.Generator:
    """Per-rank RNG stream for sharded sampling (config 5: seed = base + rank)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed + rank)
    return g
