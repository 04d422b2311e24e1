"""CPU restatement of the reference's LeanNPE pieces around the flow.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Pinned by
``tests/golden/{own_code_small,encoder}.npz`` (made by running the reference's
own classes, ``tests/golden/make_golden.py``).

Written as explicit arithmetic over a weight dictionary whose keys are the
reference's ``state_dict`` names, so it also serves as the specification the
HIP embedding kernels are checked against.  Line numbers refer to
``/root/reference/src/ahsd/models/lean_npe.py`` (LN) and
``.../coherent_encoder.py`` (CE).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

PARAM_NAMES = ["mass_1", "mass_2", "luminosity_distance", "ra", "dec", "theta_jn",
               "psi", "phase", "geocent_time", "a1", "a2"]          # LN:41-45

_TWO_PI = 2.0 * math.pi
# (lo, hi, log-space?)  LN:54-66
_RANGES = {
    "mass_1": (1.0, 105.0, True), "mass_2": (1.0, 105.0, True),
    "luminosity_distance": (40.0, 2200.0, True),
    "ra": (0.0, _TWO_PI, False), "dec": (-math.pi / 2, math.pi / 2, False),
    "theta_jn": (0.0, math.pi, False), "psi": (0.0, math.pi, False),
    "phase": (0.0, _TWO_PI, False), "geocent_time": (-1.6, 1.6, False),
    "a1": (0.0, 1.0, False), "a2": (0.0, 1.0, False),
}
_CIRCULAR = ("ra", "phase", "psi")                                    # LN:71


class ParamScalerRef:
    """LN:48-114: fixed affine (log for masses/distance) map to [-1, 1]."""

    def __init__(self, names=PARAM_NAMES, premerger=False):
        lo, hi, lg = [], [], []
        for n in names:
            a, b, is_log = _RANGES[n]
            if n == "geocent_time" and premerger:                     # LN:82-83
                a, b = -1.6, 5.2
            lo.append(math.log(a) if is_log else a)
            hi.append(math.log(b) if is_log else b)
            lg.append(is_log)
        self.lo = torch.tensor(lo, dtype=torch.float32)
        self.hi = torch.tensor(hi, dtype=torch.float32)
        self.log_mask = torch.tensor(lg)
        self.circ_mask = torch.tensor([n in _CIRCULAR for n in names])

    def normalize(self, x):                                            # LN:106-109
        x = torch.where(self.log_mask, torch.log(x.clamp_min(1e-6)), x)
        return (2 * (x - self.lo) / (self.hi - self.lo) - 1).clamp(-1.0, 1.0)

    def denormalize(self, y):                                          # LN:111-114
        x = (y.clamp(-1.0, 1.0) + 1) / 2 * (self.hi - self.lo) + self.lo
        return torch.where(self.log_mask, torch.exp(x), x)

    def wrap(self, y):                                                 # LN:100-104
        w = torch.remainder(y + 1.0, 2.0) - 1.0
        return torch.where(self.circ_mask, w, y.clamp(-1.0, 1.0))


# --------------------------------------------------------------------------- #
# strain embedding  (LeanStrainEncoder, LN:131-252)
# --------------------------------------------------------------------------- #
STEM = ((1, 32, 64, 8), (32, 64, 16, 4), (64, 128, 8, 4), (128, 192, 4, 2))  # LN:158-163


def sinusoidal_positions(n: int, d_model: int) -> torch.Tensor:          # LN:117-128
    pe = torch.zeros(n, d_model)
    pos = torch.arange(n, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def sanitize_strain(strain):                                             # LN:207
    return torch.nan_to_num(strain, nan=0.0, posinf=100.0, neginf=-100.0).clamp(-100.0, 100.0)


def window_log_energy(clean, n_windows=16):                              # LN:210-212
    B, D, T = clean.shape
    win = clean[:, :, : (T // n_windows) * n_windows].reshape(B, D, n_windows, -1)
    return torch.log((win ** 2).mean(dim=-1) + 1e-8)


def stem_forward(w: Dict[str, torch.Tensor], x, return_stages=False):
    """x [N,1,T] -> [N,192,61]; four strided valid convs each followed by exact GELU."""
    stages = []
    for i, (_, _, _, stride) in enumerate(STEM):
        x = F.gelu(F.conv1d(x, w[f"stem.{2 * i}.weight"], w[f"stem.{2 * i}.bias"], stride=stride))
        stages.append(x)
    return (x, stages) if return_stages else x


def _same(t):
    return t


def _mha(q_in, kv_in, in_w, in_b, out_w, out_b, n_heads, rnd=_same, attn_drop=None):
    """torch.nn.MultiheadAttention, batch_first, no mask.  ``rnd`` is applied to every matrix-product
    operand (identity by default; the GPU tests pass a bf16 round trip to mirror the kernel's operand
    rounding -- products and sums stay fp32 either way).  ``attn_drop`` [B, heads, Lq, Lk] (or None = eval): the
    factors nn.MultiheadAttention's dropout multiplies the attention probabilities with (0 or 1 / (1 - p))."""
    B, Lq, E = q_in.shape
    Lk = kv_in.shape[1]
    q = F.linear(rnd(q_in), rnd(in_w[:E]), in_b[:E])
    k = F.linear(rnd(kv_in), rnd(in_w[E:2 * E]), in_b[E:2 * E])
    v = F.linear(rnd(kv_in), rnd(in_w[2 * E:]), in_b[2 * E:])
    hd = E // n_heads
    q = q.reshape(B, Lq, n_heads, hd).transpose(1, 2) / math.sqrt(hd)
    k = k.reshape(B, Lk, n_heads, hd).transpose(1, 2)
    v = v.reshape(B, Lk, n_heads, hd).transpose(1, 2)
    o = _softmax_weighted(rnd(q) @ rnd(k).transpose(-1, -2), v, rnd, attn_drop).transpose(1, 2).reshape(B, Lq, E)
    return F.linear(rnd(o), rnd(out_w), out_b)


def _softmax_weighted(scores, v, rnd=_same, drop=None):
    """softmax(scores) @ v written as (exp(s - max) @ v) / sum(exp(s - max)): the same value, with the
    matrix-product operand being the UN-normalised weights (where a reduced-precision kernel rounds)."""
    e = torch.exp(scores - scores.amax(dim=-1, keepdim=True))
    if drop is not None:                      # dropout acts on the normalised probabilities
        return rnd(e / e.sum(dim=-1, keepdim=True) * drop) @ rnd(v)
    return (rnd(e) @ rnd(v)) / e.sum(dim=-1, keepdim=True)


def fusion_forward(w, tokens, n_layers=3, n_heads=6, rnd=_same, drop=None):
    """pre-norm TransformerEncoderLayer x n_layers, GELU FFN (LN:167-172).  ``drop`` = None: eval.  Train mode with the
    nn.Dropout modules replaced by given factors: drop[(layer, site)] for site in "attn" [B, heads, T, T], "res1" [B, T, E]
    (dropout1, on the attention branch), "ffn" [B, T, 4E] (after the activation), "res2" [B, T, E] (dropout2)."""
    x = tokens
    E = x.shape[-1]
    one = lambda l, k: 1.0 if drop is None else drop[(l, k)]
    for l in range(n_layers):
        p = f"fusion.layers.{l}."
        y = F.layer_norm(x, (E,), w[p + "norm1.weight"], w[p + "norm1.bias"], 1e-5)
        x = x + one(l, "res1") * _mha(y, y, w[p + "self_attn.in_proj_weight"], w[p + "self_attn.in_proj_bias"],
                                      w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"], n_heads, rnd,
                                      None if drop is None else drop[(l, "attn")])
        y = F.layer_norm(x, (E,), w[p + "norm2.weight"], w[p + "norm2.bias"], 1e-5)
        h = one(l, "ffn") * F.gelu(F.linear(rnd(y), rnd(w[p + "linear1.weight"]), w[p + "linear1.bias"]))
        y = F.linear(rnd(h), rnd(w[p + "linear2.weight"]), w[p + "linear2.bias"])
        x = x + one(l, "res2") * y
    return x


def _mlp2(w, prefix, x):
    x = F.gelu(F.linear(x, w[prefix + ".0.weight"], w[prefix + ".0.bias"]))
    return F.gelu(F.linear(x, w[prefix + ".2.weight"], w[prefix + ".2.bias"]))


def encoder_features(w, strain, asd_bands=None, extra_tokens=None, n_heads=6,
                     n_layers=3, n_energy_windows=16, psd_bands=0, drop=None):
    """LN:199-243 -> (feats [B, 8*192 + 64 (+32)], sanitized strain)."""
    B, D, T = strain.shape
    clean = sanitize_strain(strain)
    energy_feat = _mlp2(w, "energy_mlp", window_log_energy(clean, n_energy_windows).reshape(B, -1))
    tok = stem_forward(w, torch.asinh(clean).reshape(B * D, 1, T)).transpose(1, 2)   # [B*D, L, E]
    L, E = tok.shape[1], tok.shape[2]
    tok = tok + sinusoidal_positions(L, E).unsqueeze(0)
    tok = tok.reshape(B, D, L, E) + w["detector_embed.weight"][None, :D, None, :]
    tok = tok.reshape(B, D * L, E)
    if extra_tokens is not None:
        tok = torch.cat([extra_tokens, tok], dim=1)
    tok = fusion_forward(w, tok, n_layers, n_heads, drop=drop)
    q = w["pool_queries"].unsqueeze(0).expand(B, -1, -1)
    pooled = _mha(q, tok, w["pool_attn.in_proj_weight"], w["pool_attn.in_proj_bias"],
                  w["pool_attn.out_proj.weight"], w["pool_attn.out_proj.bias"], n_heads)
    feats = [pooled.reshape(B, -1), energy_feat]
    if psd_bands > 0:
        if asd_bands is None:
            asd_bands = strain.new_zeros(B, D, psd_bands)
        feats.append(_mlp2(w, "noise_mlp", asd_bands.reshape(B, -1)))
    return torch.cat(feats, dim=1), clean


def out_proj(w, feats):                                                   # LN:194-197
    return F.linear(F.gelu(F.linear(feats, w["out_proj.0.weight"], w["out_proj.0.bias"])),
                    w["out_proj.2.weight"], w["out_proj.2.bias"])


def lean_encoder_forward(w, strain, asd_bands=None, psd_bands=0):         # LN:245-252
    feats, _ = encoder_features(w, strain, asd_bands, psd_bands=psd_bands)
    return out_proj(w, feats)


# --------------------------------------------------------------------------- #
# CoherentEncoder geometry front-end  (CE:42-123)
# --------------------------------------------------------------------------- #
SR, T_LEN, F_LO, F_HI = 4096, 16384, 20.0, 1024.0


class CoherentGeometry:
    def __init__(self, n_det=3, bands=16, tau_max_ms=30.0):
        freqs = np.fft.rfftfreq(T_LEN, 1.0 / SR)
        sel = (freqs >= F_LO) & (freqs < F_HI)
        self.band_lo = int(np.argmax(sel))
        self.Nf = int(sel.sum())
        fb = freqs[sel]
        edges = np.geomspace(F_LO, F_HI, bands + 1)
        member = np.stack([((fb >= edges[k]) & (fb < edges[k + 1])) for k in range(bands)])
        self.Bsum = torch.from_numpy(member.astype(np.float32))           # [K, Nf]
        self.bcount = self.Bsum.sum(1).clamp_min(1.0)
        self.maxlag = int(tau_max_ms * 1e-3 * SR)
        self.lags_norm = torch.arange(-self.maxlag, self.maxlag + 1).float() / self.maxlag
        self.pairs = [(i, j) for i in range(n_det) for j in range(i + 1, n_det)]
        self.n_rfft = T_LEN // 2 + 1

    def rel(self, clean):                                                 # CE:93-116
        B = clean.shape[0]
        fd = torch.fft.rfft(clean.float().contiguous(), norm="ortho", dim=-1)
        d = fd[..., self.band_lo: self.band_lo + self.Nf]
        dr, di = d.real, d.imag
        P = dr ** 2 + di ** 2
        amp = torch.sqrt(P + 1e-12)
        feats = [torch.log((P @ self.Bsum.T) / self.bcount + 1e-8).reshape(B, -1)]
        for i, j in self.pairs:
            xr = dr[:, i] * dr[:, j] + di[:, i] * di[:, j]
            xi = di[:, i] * dr[:, j] - dr[:, i] * di[:, j]
            den = (amp[:, i] * amp[:, j]) @ self.Bsum.T + 1e-8
            gr, gi = (xr @ self.Bsum.T) / den, (xi @ self.Bsum.T) / den
            gm = torch.sqrt(gr ** 2 + gi ** 2) + 1e-8
            feats += [gm, gr / gm, gi / gm]
            # GCC delay + peak sharpness (CE:79-91)
            full = torch.zeros(B, self.n_rfft, dtype=torch.complex64)
            full[:, self.band_lo: self.band_lo + self.Nf] = torch.complex(xr, xi)
            cc = torch.fft.irfft(full, n=T_LEN, dim=-1)
            a = torch.cat([cc[:, -self.maxlag:], cc[:, : self.maxlag + 1]], dim=1).abs()
            feats += [self.lags_norm[a.argmax(-1)].unsqueeze(-1),
                      (a.max(-1).values / (a.mean(-1) + 1e-8)).unsqueeze(-1)]
            ei, ej = P[:, i].sum(-1), P[:, j].sum(-1)
            feats.append((torch.log(ei + 1e-8) - torch.log(ej + 1e-8)).unsqueeze(-1))
        return torch.cat(feats, dim=-1)


def coherent_encoder_forward(w, strain, asd_bands, geom: Optional[CoherentGeometry] = None,
                             n_geom_tokens=4, psd_bands=16):              # CE:118-123
    geom = geom or CoherentGeometry()
    clean = sanitize_strain(strain)
    g = _mlp2(w, "geom_mlp", geom.rel(clean))
    E = w["detector_embed.weight"].shape[1]
    gtok = F.linear(g, w["geom_to_tokens.weight"], w["geom_to_tokens.bias"]).reshape(-1, n_geom_tokens, E)
    feats, _ = encoder_features(w, clean, asd_bands, extra_tokens=gtok, psd_bands=psd_bands)
    return out_proj(w, feats)


# --------------------------------------------------------------------------- #
# caller-side algebra  
# --------------------------------------------------------------------------- #
def log_prob_physical(neg_logq_norm, y_norm, scaler: ParamScalerRef):
    """inference/pipeline.py:57-76: add the ParamScaler Jacobian."""
    theta = scaler.denormalize(y_norm)
    jac = (math.log(2.0) - torch.log(scaler.hi - scaler.lo)).sum()
    log_theta = torch.where(scaler.log_mask, torch.log(theta.clamp_min(1e-6)), torch.zeros_like(theta))
    return -neg_logq_norm + jac - log_theta.sum(dim=1)


def batch_nll_ref(nll_fn, context, params, nsig):
    """experiments/train_lean_npe.py:108-127: per-rank loop, mean per-signal NLL.
    nll_fn(params[n,11], rank[n], context[n,C]) -> nll[n]."""
    total, count = 0.0, 0
    for r in range(int(nsig.max().item())):
        idx = (nsig > r).nonzero(as_tuple=True)[0]
        if idx.numel() == 0:
            continue
        rank = torch.full((idx.numel(),), r, dtype=torch.long)
        total = total + nll_fn(params[idx, r, :], rank, context[idx]).sum()
        count += idx.numel()
    return total / count
