"""Trainer inner step of the reference on the HIP flow path, data-parallel.

Mirrors ``experiments/train_lean_npe.py``: optimiser AdamW(lr 3e-4, weight_decay 1e-5) (:301),
LambdaLR linear warm-up (500 steps) then cosine decay to 1 % (:305-311), per step
``batch_nll -> zero_grad -> backward -> clip_grad_norm_(5.0) -> opt.step -> sched.step`` (:363-368),
checkpoint dictionary keys (:424-427).  Added for one-process-per-GPU data parallelism (the
reference is single-device): every rank back-propagates ``sum nll / N`` with N the GLOBAL number of
(event, rank) pairs (one 16-byte all-reduce queued before the backward), the gradients are all-reduced
(SUM) in a few flat buckets over RCCL/xGMI before clipping -- so the reduced gradient IS the gradient of
the reference's loss ``sum nll / sum nsig`` (train_lean_npe.py:108-127) over the concatenated batch,
whatever the ranks' signal counts are -- and the reported loss is that global per-signal mean.

Loss and gradients both come from the HIP kernels: the flow's backward is ``_flow_autograd.py`` (re-evaluation,
chain kernel, hand-written transposed GEMMs), the encoder's is ``_enc_train.py``."""
from __future__ import annotations

import math
import os
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from .npe import LeanNPE, batch_nll

LR, WEIGHT_DECAY, WARMUP_STEPS, GRAD_CLIP = 3e-4, 1e-5, 500, 5.0      # train_lean_npe.py:185-188, 301, 366


def make_optimizer(model: torch.nn.Module, lr: float = LR, fused: Optional[bool] = None) -> torch.optim.Optimizer:
    """AdamW with the reference's hyper-parameters (train_lean_npe.py:301-311).  ``fused``: torch's single-launch
    implementation (same update rule); default: on when every parameter lives on a GPU."""
    params = list(model.parameters())
    if fused is None:
        fused = len(params) > 0 and all(p.is_cuda for p in params) and os.environ.get("PF_FUSED_ADAMW", "1") != "0"
    return torch.optim.AdamW(params, lr=lr, weight_decay=WEIGHT_DECAY, fused=bool(fused))


def lr_factor(step: int, total_steps: int, warmup_steps: int = WARMUP_STEPS) -> float:
    if step < warmup_steps:
        return step / max(1, warmup_steps)
    t = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return 0.01 + 0.99 * 0.5 * (1.0 + math.cos(math.pi * min(t, 1.0)))


def make_scheduler(opt, total_steps: int, warmup_steps: int = WARMUP_STEPS):
    return torch.optim.lr_scheduler.LambdaLR(opt, lambda s: lr_factor(s, total_steps, warmup_steps))


def allreduce_gradients(params: Iterable[torch.nn.Parameter], bucket_bytes: int = 16 << 20, group=None,
                        average: bool = True) -> None:
    """Mean (``average=False``: sum) of the gradients over ranks, in flat fp32 buckets.  ~8.9 M parameters = 35 MB fp32:
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
        if average:
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


class OverlappedGradReducer:
    """Data-parallel gradient mean that runs UNDER the backward pass (<= 4 buckets, overlapped).

    The gradients of all trainable parameters live in a few pre-allocated flat fp32 buckets (``p.grad`` is a view into
    its bucket, so nothing is concatenated or copied back); buckets are filled in reverse parameter order -- roughly
    the order in which backward produces gradients -- and a post-accumulate hook marks a bucket ready the moment its
    last gradient has been accumulated, while autograd is still working on the earlier layers.  ``finish()`` waits
    for the outstanding reductions and scales by 1 / world (``finish(average=False)``: leaves the sum -- the form
    ``train_step`` uses, whose local losses are already divided by the global signal count).  35 MB of fp32 gradients = 4 buckets of ~9 MB: a ring
    all-reduce of one bucket is ~0.1 ms per link direction, hidden behind the backward of the encoder.

    Contract -- ONE ``backward()`` per ``zero()``:
    * collectives are issued in bucket order on every rank (bucket i goes out only after buckets 0 .. i-1, whatever
      order the hooks fire in; what is not out when backward ends goes out in ``finish()``, in order): ranks whose
      graphs differ (a parameter unused on some of them) still issue the same sequence of all-reduces;
    * a gradient that arrives for a bucket whose all-reduce is already in flight -- a second ``backward()`` before
      ``finish()`` / ``zero()`` (gradient accumulation, several losses) -- would be added to a buffer that is being or has
      been reduced and never be reduced itself: the hook raises instead.  Accumulate the LOSSES and call backward once,
      or use ``allreduce_gradients`` after the last backward.

    Use: ``reducer.zero()`` instead of ``opt.zero_grad()``, ``loss.backward()``, ``reducer.finish()``, clip, step."""

    def __init__(self, params: Iterable[torch.nn.Parameter], n_buckets: int = 4, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        plist = [p for p in params if p.requires_grad]
        plist.reverse()
        total = sum(p.numel() for p in plist)
        target = max(1, -(-total // max(1, n_buckets)))
        self.buckets: List[dict] = []
        cur: List[torch.nn.Parameter] = []
        size = 0
        for p in plist:
            cur.append(p)
            size += p.numel()
            if size >= target and len(self.buckets) < n_buckets - 1:
                self._close(cur, size)
                cur, size = [], 0
        if cur:
            self._close(cur, size)
        self._next = 0                       # index of the first bucket whose all-reduce has not been issued
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _close(self, plist, size):
        p0 = plist[0]
        flat = torch.zeros(size, dtype=p0.dtype, device=p0.device)      # (fp32 in production; the CPU tests use fp64)
        off = 0
        for p in plist:
            if p.dtype != p0.dtype or not p.dtype.is_floating_point:
                raise TypeError("OverlappedGradReducer expects floating-point parameters of one dtype")
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.buckets.append({"flat": flat, "params": plist, "pending": len(plist), "work": None, "launched": False})

    def _make_hook(self, bi):
        def hook(param):
            b = self.buckets[bi]
            if b["launched"]:
                raise RuntimeError(
                    "OverlappedGradReducer: a gradient arrived for a bucket whose all-reduce has already been issued "
                    "(second backward() before finish()/zero()?): it would never be reduced. One backward per zero().")
            b["pending"] -= 1
            if b["pending"] == 0:
                self._launch_ready()
        return hook

    def _launch_ready(self):
        """issue, in bucket order, the all-reduce of every leading bucket that is complete"""
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] <= 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _launch(self, b):
        if b["launched"]:
            return
        b["launched"] = True
        if self.world > 1:
            b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def zero(self):
        """zero the buckets (p.grad stay views into them) and re-arm the hooks' counters"""
        self._next = 0
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"], b["work"], b["launched"] = len(b["params"]), None, False
            off = 0
            for p in b["params"]:                      # someone may have set .grad to None (zero_grad(set_to_none=True))
                if p.grad is None or p.grad.data_ptr() != b["flat"].data_ptr() + b["flat"].element_size() * off:
                    p.grad = b["flat"][off:off + p.numel()].view_as(p)
                off += p.numel()

    def finish(self, average: bool = True):
        """after backward(): reduce -- in bucket order -- the buckets that are not out yet (parameters without a gradient
        this step), wait, take the mean (``average=False``: keep the sum over ranks)"""
        for b in self.buckets[self._next:]:
            self._launch(b)
        self._next = len(self.buckets)
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
            if self.world > 1 and average:
                b["flat"].div_(self.world)

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def _world(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def train_step(model: LeanNPE, opt, sched, strain, params, nsig, asd_bands=None, group=None,
               reducer: Optional[OverlappedGradReducer] = None, row_cap="exact", sync: bool = True) -> Dict[str, float]:
    """One optimisation step on this rank's shard of the batch; on N ranks the parameters move exactly as the reference's
    single-process step on the concatenated batch would move them (loss ``sum nll / sum nsig`` over ALL ranks' events: each
    rank back-propagates its ``sum nll`` divided by the global pair count and the gradients are SUMMED).  With a
    ``reducer`` the gradient all-reduce overlaps the backward pass (one backward per step: the reducer's contract);
    without one (single rank, or the simple path) it runs after it.  ``row_cap`` (``batch_nll``): "exact" (default: the existing (event, rank) pairs, one host sync per step), an int
    (static bound, no sync) or None (all max_signals rows per event).  ``sync=False`` returns the loss and the gradient norm
    as 0-dim device tensors instead of Python floats: the reference reads ``loss.item()`` every step
    (train_lean_npe.py:368), which makes the host wait for the whole step before it queues the next one (~1 ms of idle GPU
    per 13 ms step here); a loop that logs every k-th step converts only then."""
    total, count = batch_nll(model, strain, params, nsig, asd_bands, row_cap=row_cap, reduction="sum")
    # the reference's loss over the GLOBAL batch: sum_ranks(sum nll) / sum_ranks(pairs).  The global count is needed
    # before the backward (it scales every gradient), the global sum only for the log: both in one 16-byte all-reduce
    # queued here, in float64 (sums of ~1e3 terms of ~1e1 nats)
    world = _world(group)
    if world > 1:
        red = torch.stack([total.detach().double(), count.double()])
        dist.all_reduce(red, group=group)
        n_global = red[1].clamp_min(1.0)
        loss = total / n_global.to(total.dtype)
        mean_nll = red[0] / n_global
    else:                                              # one rank: the same arithmetic without the float64 pair (4 launches fewer)
        loss = total / count.clamp_min(1.0)
        mean_nll = loss.detach()
    if reducer is not None:
        reducer.zero()
        loss.backward()
        reducer.finish(average=False)
    else:
        opt.zero_grad(set_to_none=True)
        loss.backward()
        allreduce_gradients(model.parameters(), group=group, average=False)
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), GRAD_CLIP)
    opt.step()
    if sched is not None:
        sched.step()
    if not sync:
        return {"loss": mean_nll, "grad_norm": gn}
    return {"loss": float(mean_nll), "grad_norm": float(gn)}


def checkpoint_dict(model: LeanNPE, epoch: int, val_nll: float, diagnostics: Optional[dict] = None,
                    args: Optional[dict] = None) -> dict:
    """Same keys as the reference's best_model.pth (train_lean_npe.py:424-427)."""
    return {"model_state_dict": model.state_dict(), "epoch": epoch, "val_nll": val_nll,
            "diagnostics": diagnostics or {}, "args": args or {}}
