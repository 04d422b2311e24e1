"""Caller-side sampling loop of the reference's inference pipeline on the HIP flow path
(SURVEY.md 8f row 1): ``ahsd.inference.pipeline`` lines 57-76 (``_log_prob_physical``) and
161-186 (encode once, draw, wrap, railing mask, denormalise, log q, mass sort).

Data fetching, preprocessing, OOD scoring, gating and result objects stay with the reference
(out of scope); this module is the part that touches the flow.  Everything stays on the GPU;
draws are generated in chunks of ``batch_size`` like the reference (``pipeline.py:105,169``) but
the per-event context is passed to ``pf_flow_inverse`` once per chunk un-expanded."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from .npe import LeanNPE


@torch.no_grad()
def log_prob_physical(model: LeanNPE, y_norm: torch.Tensor, full_ctx: torch.Tensor) -> torch.Tensor:
    """log q(theta | d) in PHYSICAL units for normalised samples (pipeline.py:57-76): the flow's
    density in normalised space plus the ParamScaler Jacobian
    sum_j [log 2 - log(hi_j - lo_j)] - sum_{log dims} log theta_j."""
    if full_ctx.shape[0] != y_norm.shape[0]:
        full_ctx = full_ctx.expand(y_norm.shape[0], -1).contiguous()
    neg_logq = model.flow.compute_psd_aware_nll(y_norm, full_ctx, None)
    sc = model.scaler
    theta = sc.denormalize(y_norm)
    jac = (math.log(2.0) - torch.log(sc.hi - sc.lo)).sum()
    log_theta = torch.where(sc.log_mask, theta.clamp_min(1e-6).log(), torch.zeros_like(theta))
    return -neg_logq + jac - log_theta.sum(dim=1)


@torch.no_grad()
def sample_event(model: LeanNPE, strain: torch.Tensor, num_samples: int = 10000, rank: int = 0,
                 asd_bands: Optional[torch.Tensor] = None, seed: Optional[int] = None,
                 batch_size: int = 98304) -> Dict[str, torch.Tensor]:
    """Posterior draws for ONE event (strain [1, n_det, 16384]), pipeline.py:161-186.
    Returns device tensors: samples [n, P] (physical units, fp64, m1 >= m2), logq [n] (fp64),
    railed [n] (bool, a non-circular parameter within 1e-3 of its bound), context [1, C].
    ``batch_size``: draws per inverse call; the reference chunks by 4096 for memory (pipeline.py:105,169), here
    the default is 98 304 = 8 full rounds of 256 CUs x 3 sixteen-draw workgroups (a chunk that ends in a nearly
    empty round of workgroups wastes that round: 12 289 draws cost as much as 24 576)."""
    dev = strain.device
    gen = None
    if seed is not None:
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
    ctx = model.encode(strain, asd_bands)
    r = torch.full((1,), rank, dtype=torch.long, device=dev)
    full_ctx = model._full_context(ctx, r)                                   # [1, C]
    n_par = len(model.param_names)
    samples = torch.empty(num_samples, n_par, dtype=torch.float64, device=dev)
    logq = torch.empty(num_samples, dtype=torch.float64, device=dev)
    railed = torch.zeros(num_samples, dtype=torch.bool, device=dev)
    for i in range(0, num_samples, batch_size):
        k = min(batch_size, num_samples - i)
        z = torch.randn(k, n_par, device=dev, generator=gen)
        y, _ = model.flow.inverse(z, full_ctx)
        y = model.scaler.wrap(y)
        railed[i:i + k] = ((y.abs() > 0.999) & ~model.scaler.circ_mask).any(dim=1)
        samples[i:i + k] = model.scaler.denormalize(y).double()
        logq[i:i + k] = log_prob_physical(model, y, full_ctx).double()
    if "mass_1" in model.param_names and "mass_2" in model.param_names:      # pipeline.py:184-186
        j1, j2 = model.param_names.index("mass_1"), model.param_names.index("mass_2")
        m1, m2 = samples[:, j1].clone(), samples[:, j2].clone()
        samples[:, j1], samples[:, j2] = torch.maximum(m1, m2), torch.minimum(m1, m2)
    return {"samples": samples, "logq": logq, "railed": railed, "context": ctx,
            "boundary_railing_frac": railed.float().mean()}
