"""CPU restatement of the reference's flow wrapper (``src/ahsd/models/flows.py``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  ``NSFPosteriorFlowRef``
follows ``NSFPosteriorFlow`` method by method; line numbers refer to
``/root/reference/src/ahsd/models/flows.py``.  Only the plain-context
conditioner is what LeanNPE uses (``use_masked_context=False``, ``lean_npe.py:294``); the
masked-context variant (flows.py:112-360) is restated too.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from . import nflows_restated as nfr

FLOW_NORM_BOUND = 3.0  # src/ahsd/models/parameter_scalers.py:27


class PSDScaledNormalRef(nn.Module):
    """flows.py:28-109.  log N(z; 0, diag(exp(ls))^2)."""

    def __init__(self, shape):
        super().__init__()
        self.dim = shape[0] if isinstance(shape, (list, tuple)) else shape

    def log_prob(self, z, log_sigma_psd):
        if z.shape != log_sigma_psd.shape:  # flows.py:68-71
            raise ValueError(f"Shape mismatch: z {z.shape} vs log_sigma_psd {log_sigma_psd.shape}")
        quad = ((z / torch.exp(log_sigma_psd)) ** 2).sum(dim=1)
        var = 2 * log_sigma_psd.sum(dim=1)
        return -0.5 * (quad + var + self.dim * math.log(2 * math.pi))

    def sample(self, num_samples, log_sigma_psd=None):  # flows.py:87-109
        b = log_sigma_psd.shape[0] if log_sigma_psd is not None else 1
        dt = log_sigma_psd.dtype if log_sigma_psd is not None else torch.float32
        return torch.randn(b, num_samples, self.dim, dtype=dt)


class NSFPosteriorFlowRef(nn.Module):
    def __init__(self, features, context_features=0, hidden_features=256,
                 num_layers=12, num_bins=16, tail_bound=FLOW_NORM_BOUND,
                 dropout=0.0, temperature_scale=1.5, scale_by_sqrt_hidden=False, use_masked_context=False,
                 full_context=True):
        super().__init__()
        self.features = features
        self.context_features = context_features
        self.num_layers = num_layers
        # flows.py:471/517: anything that is not a python float falls back to 3.0
        tb = tail_bound if isinstance(tail_bound, float) else FLOW_NORM_BOUND
        self.tail_bound = tb
        self.temperature = nn.Parameter(torch.tensor(temperature_scale, dtype=torch.float32))
        self.base_dist = PSDScaledNormalRef([features])
        self.use_masked_context = use_masked_context
        self.n_context_blocks = features if use_masked_context else None
        self.context_block_dim = context_features // features if use_masked_context else None
        ts = []
        for _ in range(num_layers):  # flows.py:449-526
            if not use_masked_context:       # flows.py:459-460: no permutation in masked-context mode
                ts.append(nfr.ReversePermutation(features))
            ts.append(nfr.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(
                features, hidden_features,
                context_features if context_features > 0 else None,
                num_bins=num_bins, tail_bound=tb, num_blocks=2,
                dropout_probability=dropout,
                scale_by_sqrt_hidden=scale_by_sqrt_hidden,
                masked_context_blocks=(features, context_features // features, full_context) if use_masked_context else None))
        self.transform = nfr.CompositeTransform(ts)
        self.register_buffer("_ar_perm", torch.arange(features))
        self.register_buffer("_ar_inv_perm", torch.arange(features))

    # flows.py:550-588
    def set_autoregressive_order(self, order):
        if sorted(order) != list(range(self.features)):
            raise ValueError(f"order must be a permutation of range({self.features}), got {order}")
        self._ar_perm = torch.tensor(order, dtype=torch.long)
        self._ar_inv_perm = torch.argsort(self._ar_perm)

    # flows.py:590-608
    def _permute_context_blocks(self, context):
        if not self.use_masked_context:
            return context
        b = context.shape[0]
        blocks = context.view(b, self.n_context_blocks, self.context_block_dim)[:, self._ar_perm, :]
        return blocks.reshape(b, -1)

    # flows.py:610-618
    def forward(self, x, context=None):
        x = x[:, self._ar_perm]
        if self.context_features > 0 and context is not None:
            return self.transform(x, self._permute_context_blocks(context))
        return self.transform(x)

    # flows.py:620-655
    def inverse(self, z, context=None, n_overlaps=None):
        if context is not None and not torch.isfinite(context).all():
            context = torch.nan_to_num(context, nan=0.0, posinf=1e-3, neginf=-1e-3)
        if self.context_features > 0 and context is not None:
            try:
                x, ld = self.transform.inverse(z, self._permute_context_blocks(context))
            except AssertionError:
                x, ld = z, z.new_zeros(z.shape[0])
        else:
            x, ld = self.transform.inverse(z)
        x = x[:, self._ar_inv_perm]
        if not torch.isfinite(x).all():
            x = torch.nan_to_num(x, nan=0.0, posinf=1.0, neginf=-1.0)
        return torch.clamp(x, -FLOW_NORM_BOUND, FLOW_NORM_BOUND), ld

    def inverse_raw(self, z, context=None):
        """transform.inverse without the wrapper's clamp (for round-trip KATs)."""
        x, ld = self.transform.inverse(z, None if context is None else self._permute_context_blocks(context))
        return x[:, self._ar_inv_perm], ld

    # flows.py:727-779
    def compute_psd_aware_nll(self, x, context, log_sigma_psd):
        z, ld = self.forward(x, context)
        return -(self.base_dist.log_prob(z, log_sigma_psd) + ld)

    # flows.py:657-695, to the documented math with the N(0, I) base
    def log_prob(self, x, context=None, temperature: Optional[float] = None):
        if context is not None and not torch.isfinite(context).all():
            context = torch.nan_to_num(context, nan=0.0, posinf=1e-3, neginf=-1e-3)
        t = torch.clamp(self.temperature, 0.5, 3.0) if temperature is None \
            else torch.as_tensor(temperature, dtype=x.dtype)
        xs = x / t
        z, ld = self.forward(xs, context)
        logp = self.base_dist.log_prob(z, torch.zeros_like(z)) + ld
        out = logp - self.features * torch.log(t)
        out = torch.nan_to_num(out, nan=1000.0) if torch.isnan(out).any() else out
        return -out

    # flows.py:781-842
    @torch.no_grad()
    def sample_psd_aware(self, num_samples, context, log_sigma_psd, z=None):
        b = context.shape[0]
        if z is None:
            z = self.base_dist.sample(num_samples, log_sigma_psd)
        ctx = context.unsqueeze(1).expand(b, num_samples, self.context_features)
        x, _ = self.inverse(z.reshape(b * num_samples, self.features),
                            ctx.reshape(b * num_samples, self.context_features))
        x = x.clamp(-FLOW_NORM_BOUND, FLOW_NORM_BOUND).reshape(b, num_samples, self.features)
        return x * torch.exp(log_sigma_psd).unsqueeze(1)

    # flows.py:910-920
    def compute_bounds_penalty(self, params_norm, bounds=(-FLOW_NORM_BOUND, FLOW_NORM_BOUND)):
        lo, hi = bounds
        return torch.relu(lo - params_norm).mean() + torch.relu(params_norm - hi).mean()

    # flows.py:922-939
    def compute_endpoint_loss(self, params_norm, context):
        b = params_norm.shape[0]
        zmin = torch.full((b, self.features), -FLOW_NORM_BOUND, dtype=params_norm.dtype)
        xmin, _ = self.inverse(zmin, context)
        xmax, _ = self.inverse(-zmin, context)
        return (torch.relu(xmin - params_norm) + torch.relu(params_norm - xmax)).mean()


def scale_final_layers(flow: NSFPosteriorFlowRef, factor: float = 30.0):
    """BASELINE.md section 3 / SURVEY 8d: default init is near-identity, so the
    measurement configs rescale the final MADE layer to exercise the bin search."""
    with torch.no_grad():
        for t in flow.transform._transforms:
            if hasattr(t, "autoregressive_net"):
                t.autoregressive_net.final_layer.weight.mul_(factor)
                t.autoregressive_net.final_layer.bias.mul_(factor)
    return flow


def flops_per_sample(D, C, H, K, L, num_blocks=2):
    """SURVEY 8d dense-GEMM count: 2 L (D H + C H + nb (2 H^2 + C H) + H D (3K-1))."""
    return 2 * L * (D * H + C * H + num_blocks * (2 * H * H + C * H) + H * D * (3 * K - 1))
