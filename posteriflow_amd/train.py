"""Trainer inner step of the reference on the HIP flow path, data-parallel.

Mirrors ``experiments/train_lean_npe.py``: optimiser AdamW(lr 3e-4, weight_decay 1e-5) (:301),
LambdaLR linear warm-up (500 steps) then cosine decay to 1 % (:305-311), per step
``batch_nll -> zero_grad -> backward -> clip_grad_norm_(5.0) -> opt.step -> sched.step`` (:363-368),
checkpoint dictionary keys (:424-427).  Added for one-process-per-GPU data parallelism (the
reference is single-device): the gradients are all-reduced (mean) in a few flat buckets over
RCCL/xGMI before clipping, and the reported loss is the global per-signal mean.

The loss value comes from the HIP kernels; gradients flow through the interim tensor-op backward
(``_flow_autograd.py``)."""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from .npe import LeanNPE, batch_nll

LR, WEIGHT_DECAY, WARMUP_STEPS, GRAD_CLIP = 3e-4, 1e-5, 500, 5.0      # train_lean_npe.py:185-188, 301, 366


def make_optimizer(model: torch.nn.Module, lr: float = LR) -> torch.optim.Optimizer:
    return torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=WEIGHT_DECAY)


def lr_factor(step: int, total_steps: int, warmup_steps: int = WARMUP_STEPS) -> float:
    if step < warmup_steps:
        return step / max(1, warmup_steps)
    t = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return 0.01 + 0.99 * 0.5 * (1.0 + math.cos(math.pi * min(t, 1.0)))


def make_scheduler(opt, total_steps: int, warmup_steps: int = WARMUP_STEPS):
    return torch.optim.lr_scheduler.LambdaLR(opt, lambda s: lr_factor(s, total_steps, warmup_steps))


def allreduce_gradients(params: Iterable[torch.nn.Parameter], bucket_bytes: int = 16 << 20, group=None) -> None:
    """Mean of the gradients over ranks, in flat fp32 buckets.  ~8.9 M parameters = 35 MB fp32:
    3 buckets of <= 16 MB; ring all-reduce moves 2 (N-1)/N x 35 MB per GPU over one xGMI link
    direction (~153 GB/s) ~ 0.4 ms per step."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    grads = [p.grad for p in params if p.grad is not None]
    bucket: List[torch.Tensor] = []
    size = 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
        bucket, size = [], 0

    for g in grads:
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
    flush()


def train_step(model: LeanNPE, opt, sched, strain, params, nsig, asd_bands=None, group=None) -> Dict[str, float]:
    """One optimisation step on this rank's shard of the batch."""
    loss = batch_nll(model, strain, params, nsig, asd_bands)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    allreduce_gradients(model.parameters(), group=group)
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), GRAD_CLIP)
    opt.step()
    if sched is not None:
        sched.step()
    # global per-signal mean of the loss (weights: signals per rank)
    n_sig = nsig.sum().to(torch.float64)
    red = torch.stack([loss.detach().double() * n_sig, n_sig])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(red, group=group)
    return {"loss": (red[0] / red[1]).item(), "grad_norm": float(gn)}


def checkpoint_dict(model: LeanNPE, epoch: int, val_nll: float, diagnostics: Optional[dict] = None,
                    args: Optional[dict] = None) -> dict:
    """Same keys as the reference's best_model.pth (train_lean_npe.py:424-427)."""
    return {"model_state_dict": model.state_dict(), "epoch": epoch, "val_nll": val_nll,
            "diagnostics": diagnostics or {}, "args": args or {}}
