"""Deterministic weight / input recipes shared by ``make_golden.py`` (which runs
the reference's own classes in the build container) and by the tests (which
feed the same tensors to the oracle and to the HIP path).  Data only: nothing
here comes from the reference's sources."""
from __future__ import annotations

import math

import torch


def fill_state_dict(shapes: dict, seed: int) -> dict:
    """shapes: {key: torch.Size}.  Returns {key: float32 tensor}, filled in sorted
    key order from one generator: weights ~ N(0, 1/fan_in), 1-D ~ N(0, 0.02^2),
    LayerNorm gains ~ 1 + N(0, 0.1^2)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        t = torch.randn(shp, generator=g, dtype=torch.float32)
        if len(shp) >= 2:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            t = t / math.sqrt(fan_in)
        elif "norm" in k and k.endswith("weight"):
            t = 1.0 + 0.1 * t
        else:
            t = 0.02 * t
        out[k] = t
    return out


def strain_batch(batch: int, n_det: int, seed: int, t_len: int = 16384) -> torch.Tensor:
    """Whitened-noise-like strain N(0,1) with a loud linear chirp in event 0 and
    non-finite / out-of-range samples in event 1 (exercises lean_npe.py:207)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, n_det, t_len, generator=g, dtype=torch.float32)
    t = torch.arange(t_len, dtype=torch.float32) / 4096.0
    chirp = torch.sin(2 * math.pi * (30.0 * t + 20.0 * t * t)) * torch.exp(-((t - 2.5) / 0.6) ** 2)
    for d in range(n_det):
        x[0, d] += (6.0 - d) * torch.roll(chirp, 7 * d)
    if batch > 1:
        x[1, 0, 5] = float("nan")
        x[1, 0, 77] = float("inf")
        x[1, n_det - 1, 1000] = float("-inf")
        x[1, 0, 2000] = 250.0
        x[1, n_det - 1, 3000] = -1e4
    return x


def physical_params(batch: int, seed: int) -> torch.Tensor:
    """[batch, 11] physical parameters in PARAM_NAMES order, incl. out-of-range,
    zero and negative entries for the log dims and beyond-period angles."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(batch, 11, generator=g)
    lo = torch.tensor([1.0, 1.0, 40.0, 0.0, -math.pi / 2, 0.0, 0.0, 0.0, -1.6, 0.0, 0.0])
    hi = torch.tensor([105.0, 105.0, 2200.0, 2 * math.pi, math.pi / 2, math.pi, math.pi,
                       2 * math.pi, 1.6, 1.0, 1.0])
    p = lo + (hi - lo) * u
    p[0, 0] = 0.0          # log of clamp_min(1e-6)
    p[1, 1] = -3.0
    p[2, 2] = 5000.0       # above range -> clamp
    p[3, 3] = 7.5          # beyond 2 pi
    p[4, 8] = 4.0          # geocent_time past +1.6 (inside premerger range)
    p[5, 9] = 1.5
    p[6, 2] = 10.0         # below range
    return p
