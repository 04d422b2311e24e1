"""LeanNPE on MI355X: strain embedding -> context -> flow, with the reference's module API.

Counterpart of the reference's ``src/ahsd/models/lean_npe.py`` (LN) and
``src/ahsd/models/coherent_encoder.py`` (CE): same class names, constructor arguments,
attributes and ``state_dict`` keys, so checkpoints written by
``experiments/train_lean_npe.py:421-427`` load unchanged and ``inference/pipeline.py`` can use it
as is.  The flow is :class:`posteriflow_amd.flows.NSFPosteriorFlow` (HIP, ``libpfhip.so``).

Embedding status: the submodules below hold the parameters under the reference's
names.  The convolutional stem and the energy windows (the part that reads the raw strain from
HBM: sanitise, window log-energy, asinh, 4 strided convolutions + GELU) run in hand-written HIP
kernels (``csrc/pf_embed.hip`` via ``pf_embed_stem_forward``); the small fusion transformer,
attention pooling and MLPs run on device tensor ops this round.  Pinned against golden vectors
produced by the reference's own classes (tests/golden/encoder.npz).
"""
from __future__ import annotations

import math
import warnings
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .flows import NSFPosteriorFlow

PARAM_NAMES = ["mass_1", "mass_2", "luminosity_distance", "ra", "dec", "theta_jn", "psi",
               "phase", "geocent_time", "a1", "a2"]                     # LN:41-45

# name -> (low, high, log-space)                                          LN:54-66
_PRIOR_BOX = {
    "mass_1": (1.0, 105.0, True), "mass_2": (1.0, 105.0, True),
    "luminosity_distance": (40.0, 2200.0, True),
    "ra": (0.0, 2.0 * math.pi, False), "dec": (-0.5 * math.pi, 0.5 * math.pi, False),
    "theta_jn": (0.0, math.pi, False), "psi": (0.0, math.pi, False),
    "phase": (0.0, 2.0 * math.pi, False), "geocent_time": (-1.6, 1.6, False),
    "a1": (0.0, 1.0, False), "a2": (0.0, 1.0, False),
}
_PERIODIC = frozenset(("ra", "phase", "psi"))                            # LN:71
_PREMERGER_TIME = (-1.6, 5.2)                                            # LN:82-83


class ParamScaler:
    """Fixed invertible map physical parameters <-> [-1, 1] (LN:48-114).  [B, 11] elementwise
    work, evaluated with device tensor ops; ``RANGES`` / ``CIRCULAR`` kept as in the reference."""

    RANGES = _PRIOR_BOX
    CIRCULAR = tuple(sorted(_PERIODIC))

    def __init__(self, param_names: List[str] = PARAM_NAMES, premerger: bool = False):
        self.param_names = list(param_names)
        self.premerger = premerger
        lows, highs, logs = [], [], []
        for name in self.param_names:
            lo, hi, is_log = _PRIOR_BOX[name]
            if premerger and name == "geocent_time":
                lo, hi = _PREMERGER_TIME
            lows.append(math.log(lo) if is_log else lo)
            highs.append(math.log(hi) if is_log else hi)
            logs.append(is_log)
        self.lo = torch.tensor(lows, dtype=torch.float32)
        self.hi = torch.tensor(highs, dtype=torch.float32)
        self.log_mask = torch.tensor(logs, dtype=torch.bool)
        self.circ_mask = torch.tensor([n in _PERIODIC for n in self.param_names], dtype=torch.bool)

    def to(self, device):
        for attr in ("lo", "hi", "log_mask", "circ_mask"):
            setattr(self, attr, getattr(self, attr).to(device))
        return self

    def normalize(self, x: torch.Tensor) -> torch.Tensor:
        u = torch.where(self.log_mask, x.clamp_min(1e-6).log(), x)
        return (2.0 * (u - self.lo) / (self.hi - self.lo) - 1.0).clamp(-1.0, 1.0)

    def denormalize(self, y: torch.Tensor) -> torch.Tensor:
        u = (y.clamp(-1.0, 1.0) + 1.0) / 2.0 * (self.hi - self.lo) + self.lo
        return torch.where(self.log_mask, u.exp(), u)

    def wrap(self, y: torch.Tensor) -> torch.Tensor:
        """periodic parameters wrap exactly, bounded ones clamp (LN:100-104)."""
        return torch.where(self.circ_mask, torch.remainder(y + 1.0, 2.0) - 1.0, y.clamp(-1.0, 1.0))


class SinusoidalPositions(nn.Module):
    def __init__(self, d_model: int, max_len: int = 512):
        super().__init__()
        pos = torch.arange(max_len, dtype=torch.float32)[:, None]
        freq = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
        table = torch.zeros(max_len, d_model)
        table[:, 0::2] = torch.sin(pos * freq)
        table[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer("pe", table)

    def forward(self, n: int) -> torch.Tensor:
        return self.pe[:n]


_STEM_SPEC = ((1, 32, 64, 8), (32, 64, 16, 4), (64, 128, 8, 4))          # + (128, d_model, 4, 2)  LN:158-163


class LeanStrainEncoder(nn.Module):
    """Whitened strain [B, n_det, 16384] -> context [B, context_dim] (LN:131-252)."""

    def __init__(self, n_detectors: int = 3, d_model: int = 192, n_layers: int = 3, n_heads: int = 6,
                 n_pool_queries: int = 8, n_energy_windows: int = 16, context_dim: int = 256,
                 dropout: float = 0.05, psd_bands: int = 0):
        super().__init__()
        self.__dict__["_theta"] = None          # flat mode (flatten_parameters): the one leaf, registered as a Parameter
        self.n_detectors, self.n_energy_windows = n_detectors, n_energy_windows
        self.context_dim, self.psd_bands = context_dim, psd_bands
        convs = []
        for cin, cout, k, s in _STEM_SPEC + ((128, d_model, 4, 2),):
            convs += [nn.Conv1d(cin, cout, kernel_size=k, stride=s), nn.GELU()]
        self.stem = nn.Sequential(*convs)
        self.detector_embed = nn.Embedding(n_detectors, d_model)
        self.pos = SinusoidalPositions(d_model)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            block = nn.TransformerEncoderLayer(d_model=d_model, nhead=n_heads, dim_feedforward=4 * d_model,
                                               dropout=dropout, activation="gelu", batch_first=True,
                                               norm_first=True)
            self.fusion = nn.TransformerEncoder(block, num_layers=n_layers)
        self.pool_queries = nn.Parameter(torch.randn(n_pool_queries, d_model) / math.sqrt(d_model))
        self.pool_attn = nn.MultiheadAttention(d_model, n_heads, batch_first=True)
        self.energy_mlp = nn.Sequential(nn.Linear(n_detectors * n_energy_windows, 64), nn.GELU(),
                                        nn.Linear(64, 64), nn.GELU())
        extra = 0
        if psd_bands > 0:
            self.noise_mlp = nn.Sequential(nn.Linear(n_detectors * psd_bands, 64), nn.GELU(),
                                           nn.Linear(64, 32), nn.GELU())
            extra = 32
        self.out_proj = nn.Sequential(nn.Linear(n_pool_queries * d_model + 64 + extra, 512), nn.GELU(),
                                      nn.Linear(512, context_dim))

    # ---- flat parameter storage (opt-in, like NSFPosteriorFlow.flatten_parameters) -------------------------------------
    def _flat_slots(self):
        """(module, attribute, state_dict name) of the parameters of the HIP training path, in its raw layout
        (``_enc_train.train_parameters`` / pf_embed_train_* of include/pf_hip.h)"""
        out = [(self.stem[i], n, f"stem.{i}.{n}") for i in (0, 2, 4, 6) for n in ("weight", "bias")]
        for l, layer in enumerate(self.fusion.layers):
            pre = f"fusion.layers.{l}."
            out += [(layer.norm1, "weight", pre + "norm1.weight"), (layer.norm1, "bias", pre + "norm1.bias"),
                    (layer.self_attn, "in_proj_weight", pre + "self_attn.in_proj_weight"),
                    (layer.self_attn, "in_proj_bias", pre + "self_attn.in_proj_bias"),
                    (layer.self_attn.out_proj, "weight", pre + "self_attn.out_proj.weight"),
                    (layer.self_attn.out_proj, "bias", pre + "self_attn.out_proj.bias"),
                    (layer.norm2, "weight", pre + "norm2.weight"), (layer.norm2, "bias", pre + "norm2.bias"),
                    (layer.linear1, "weight", pre + "linear1.weight"), (layer.linear1, "bias", pre + "linear1.bias"),
                    (layer.linear2, "weight", pre + "linear2.weight"), (layer.linear2, "bias", pre + "linear2.bias")]
        return out + [(self.pool_attn, "in_proj_weight", "pool_attn.in_proj_weight"),
                      (self.pool_attn, "in_proj_bias", "pool_attn.in_proj_bias")]

    def flatten_parameters(self) -> "LeanStrainEncoder":
        """Re-home the 58 parameters of the stem, the Transformer layers and the pool's input projection into ONE flat leaf
        ``_theta`` (the raw layout pf_embed_train_pack reads and pf_embed_train_backward writes); ``weight`` / ``bias`` of
        the sub-modules become views of it.  A training step then hands autograd one tensor instead of 58, skips the
        per-step concatenation, and its flat gradient IS ``_theta.grad``; the optimiser steps one tensor.
        ``state_dict()`` / ``load_state_dict()`` keep the reference's key names.  Call it before building the optimiser."""
        if self.__dict__.get("_flat_shapes") is not None:
            return self
        slots = self._flat_slots()
        params = [getattr(m, n) for m, n, _ in slots]
        with torch.no_grad():
            flat = torch.cat([q.detach().reshape(-1).float() for q in params])
        self._flat_shapes = [tuple(q.shape) for q in params]
        theta = nn.Parameter(flat, requires_grad=any(q.requires_grad for q in params))
        for m, n, _ in slots:
            del m._parameters[n]
        self.__dict__.pop("_theta", None)
        self.register_parameter("_theta", theta)
        self._refresh_flat_views()
        return self

    def _refresh_flat_views(self):
        off = 0
        for (m, n, _), shp in zip(self._flat_slots(), self._flat_shapes):
            k = math.prod(shp)
            m.__dict__[n] = self._theta[off:off + k].view(shp)
            off += k
        for k in ("_train_state", "_mixer_state", "_stem_state", "_dense_state"):
            self.__dict__.pop(k, None)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if self._theta is not None:
            self._refresh_flat_views()
        return out

    _DEVICE_CACHES = ("_train_state", "_mixer_state", "_stem_state", "_dense_state", "_geom_twiddle")

    def __deepcopy__(self, memo):
        """``copy.deepcopy`` (EMA copies, snapshots): in flat mode the sub-modules' weight views of ``_theta`` (non-leaf
        tensors, which torch refuses to deep-copy) are dropped for the copy and rebuilt on both sides; packed fragments and
        the other device-side caches are not copied"""
        import copy
        flat = self._theta is not None
        for m, n, _ in (self._flat_slots() if flat else []):
            m.__dict__.pop(n, None)
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                if k not in self._DEVICE_CACHES:
                    new.__dict__[k] = copy.deepcopy(v, memo)
        finally:
            if flat:
                self._refresh_flat_views()
        if flat:
            new._refresh_flat_views()
        return new

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self._theta is not None:        # flat mode: the reference's names, not "_theta"
            destination.pop(prefix + "_theta", None)
            for m, n, name in self._flat_slots():
                v = getattr(m, n)
                destination[prefix + name] = v if keep_vars else v.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        if self._theta is not None:        # flat mode: gather the reference-named tensors into the leaf (the child modules
            with torch.no_grad():          # no longer own them and must not see the keys)
                for m, n, name in self._flat_slots():
                    src = state_dict.pop(prefix + name, None)
                    view = getattr(m, n)
                    if src is None:
                        if strict:
                            missing_keys.append(prefix + name)
                        continue
                    if tuple(src.shape) != tuple(view.shape):
                        error_msgs.append(f"size mismatch for {prefix + name}: {tuple(src.shape)} vs {tuple(view.shape)}")
                        continue
                    view.copy_(src)
            state_dict[prefix + "_theta"] = self._theta.detach()
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        if self._theta is not None:
            state_dict.pop(prefix + "_theta", None)

    # --- HIP stem ------------------------------------------------------------------------------
    _allow_tensor_op_stem = False
    precision = "fp32"        # "fp32": f32 MFMA, matches the CPU path to ~1e-6; "bf16": throughput

    def _stem_params(self):
        return [p for i in (0, 2, 4, 6) for p in (self.stem[i].weight, self.stem[i].bias)]

    def _stem_hip(self, strain):
        """(tokens [B*D, 61, E], log_energy [B, D, 16]) from the RAW strain [B, D, 16384]: one
        pf_embed_stem_forward call (sanitise + energy windows + asinh + conv stem + GELU)."""
        b, d, t = strain.shape
        dev = strain.device
        if dev.type != "cuda":
            raise _lib.PfError(f"LeanStrainEncoder stem runs on the MI355X only (input on {dev}); no CPU fallback")
        e = self.stem[6].out_channels
        if t != 16384 or e != 192:
            raise NotImplementedError("pf_embed_stem_forward is built for 4 s @ 4096 Hz segments and d_model = 192")
        L = _lib.lib()
        prec = _lib.PRECISIONS[self.precision]
        params = self._stem_params()
        key = (_lib.param_epoch(), dev, prec, tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
        st = self.__dict__.setdefault("_stem_state", {})
        if st.get("key") != key:
            if st.get("map_prec") != (dev, prec):
                host = torch.empty(L.pf_embed_stem_pack_map_len(prec), dtype=torch.int32)
                _lib.check(L.pf_embed_stem_build_pack_map(prec, host.data_ptr()), "pf_embed_stem_build_pack_map")
                st["map"], st["map_prec"] = host.to(dev), (dev, prec)
                st["packed"] = torch.empty(L.pf_embed_stem_packed_bytes(prec), dtype=torch.uint8, device=dev)
            raw = torch.cat([p.detach().reshape(-1).float() for p in params])
            assert raw.numel() == L.pf_embed_stem_raw_param_count()
            _lib.check(L.pf_embed_stem_pack(prec, raw.data_ptr(), st["map"].data_ptr(), st["packed"].data_ptr(),
                                            torch.cuda.current_stream(dev).cuda_stream), "pf_embed_stem_pack")
            st["key"] = key
        n = b * d
        x = strain.reshape(n, t).contiguous().float()
        need = L.pf_embed_stem_workspace_bytes(prec, n)
        if st.get("ws") is None or st["ws"].numel() < need or st["ws"].device != dev:
            st["ws"] = torch.empty(need, dtype=torch.uint8, device=dev)
        tokens = torch.empty(n, 61, e, dtype=torch.float32, device=dev)
        log_energy = torch.empty(n, 16, dtype=torch.float32, device=dev)
        _lib.check(L.pf_embed_stem_forward(prec, st["packed"].data_ptr(), x.data_ptr(), n, tokens.data_ptr(),
                                           log_energy.data_ptr(), st["ws"].data_ptr(), need,
                                           torch.cuda.current_stream(dev).cuda_stream), "pf_embed_stem_forward")
        return tokens, log_energy.reshape(b, d, 16)

    # --- pieces evaluated with device tensor ops ---------------------------------------------------
    @staticmethod
    def _sanitize(strain):                                                # LN:207
        return torch.nan_to_num(strain, nan=0.0, posinf=100.0, neginf=-100.0).clamp(-100.0, 100.0)

    def _window_log_energy(self, clean):                                  # LN:210-212
        b, d, t = clean.shape
        w = self.n_energy_windows
        return clean[:, :, : (t // w) * w].reshape(b, d, w, -1).square().mean(dim=-1).add(1e-8).log()

    def _stem(self, clean):
        b, d, t = clean.shape
        return self.stem(torch.asinh(clean).reshape(b * d, 1, t)).transpose(1, 2)       # [B*D, 61, E]

    def _fuse(self, tokens):
        return self.fusion(tokens)

    # --- HIP token mixer: fusion transformer + pool attention in one kernel (bf16 mode, eval) ----------
    def _mixer_params(self):
        out = []
        for layer in self.fusion.layers:
            out += [layer.norm1.weight, layer.norm1.bias, layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias,
                    layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias, layer.norm2.weight, layer.norm2.bias,
                    layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias]
        return out + [self.pool_attn.in_proj_weight, self.pool_attn.in_proj_bias]

    def _mixer_supported(self) -> bool:
        a = self.fusion.layers[0]
        return (len(self.fusion.layers) == 3 and a.self_attn.embed_dim == 192 and a.self_attn.num_heads == 6
                and a.linear1.out_features == 768 and self.pool_queries.shape == (8, 192)
                and self.pool_attn.num_heads == 6 and self.fusion.norm is None)

    # --- [B, .]-sized dense layers on the hand-written GEMMs (``_dense``: pf_dense_nt / pf_dense_tn, fp32 operands) ---------
    def _dstate(self) -> dict:
        return self.__dict__.setdefault("_dense_state", {})

    def _seq(self, name: str, x):
        """``getattr(self, name)(x)`` for an nn.Sequential / nn.Linear member, its Linears on the HIP GEMMs (host tensors,
        which only the CPU wiring tests feed, go through the module itself)"""
        mod = getattr(self, name)
        if x.device.type != "cuda":
            return mod(x)
        from . import _dense
        if isinstance(mod, nn.Linear):
            return _dense.linear(x.float(), mod.weight, mod.bias, self._dstate(), name)
        return _dense.sequential(mod, x.float(), self._dstate(), name)

    def _pool_out(self, pooled):
        m = self.pool_attn.out_proj
        if pooled.device.type != "cuda":
            return m(pooled)
        from . import _dense
        return _dense.linear(pooled.float(), m.weight, m.bias, self._dstate(), "pool_attn.out_proj")

    def _pool_queries_projected(self):
        """the 8 learned queries through the pool's query projection, scaled by 1 / sqrt(head dim) (LN:228-233: what
        nn.MultiheadAttention does with ``query = pool_queries``)"""
        from . import _dense
        e = self.pool_queries.shape[1]
        w, bias = self.pool_attn.in_proj_weight, self.pool_attn.in_proj_bias
        q = _dense.linear(self.pool_queries.float(), w[:e], bias[:e], self._dstate(), "pool_attn.q")
        return q * (1.0 / math.sqrt(e // self.pool_attn.num_heads))

    def _mix_hip(self, tok, token_bias=None):
        """[B, 8 * 192] pooled features from tokens [B, T, 192] (T <= 192): one pf_embed_fusion_forward call
        (token_bias add + 3 Transformer layers + pool attention) and the pool's out-projection.  ``tok`` is
        overwritten with the Transformer output."""
        L, dev = _lib.lib(), tok.device
        params = self._mixer_params()
        key = (_lib.param_epoch(), dev, tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
        st = self.__dict__.setdefault("_mixer_state", {})
        e = 192
        if st.get("key") != key:
            raw = torch.cat([p.detach().reshape(-1).float() for p in params[:-2]]
                            + [params[-2].detach()[e:].reshape(-1).float(), params[-1].detach()[e:].float()])
            assert raw.numel() == L.pf_embed_fusion_raw_param_count()
            if st.get("packed") is None or st["packed"].device != dev:
                st["packed"] = torch.empty(L.pf_embed_fusion_packed_bytes(), dtype=torch.uint8, device=dev)
            _lib.check(L.pf_embed_fusion_pack(raw.data_ptr(), st["packed"].data_ptr(),
                                              torch.cuda.current_stream(dev).cuda_stream), "pf_embed_fusion_pack")
            st["key"] = key
        b, t, _ = tok.shape
        assert tok.dtype == torch.float32 and tok.is_contiguous()
        with torch.autocast("cuda", enabled=False):
            q = self._pool_queries_projected().contiguous()
        pooled = torch.empty(b, 8, e, dtype=torch.float32, device=dev)
        tb = None if token_bias is None else token_bias.float().contiguous()
        _lib.check(L.pf_embed_fusion_forward(st["packed"].data_ptr(), tok.data_ptr(), t, 0 if tb is None else tb.data_ptr(),
                                             q.data_ptr(), b, pooled.data_ptr(),
                                             torch.cuda.current_stream(dev).cuda_stream), "pf_embed_fusion_forward")
        return self._pool_out(pooled).reshape(b, -1)

    def _compute_feats(self, strain, asd_bands=None, extra_tokens=None):
        """[B, 8*d_model + 64 (+32)] pre-projection features and the sanitised strain (LN:199-243).

        On the GPU there are two HIP routes: the fused inference kernels (stem + one-launch token mixer; bf16 mode, eval, no
        gradient wanted) and the TRAINING path (``_enc_train``: every differentiable or train()-mode call, and the fp32
        parity mode) -- stem, token assembly, the Transformer layers with their dropout and the pool's K / V side forward and
        backward in hand-written kernels behind one autograd node.  Tensor ops remain only for [B, .]-sized work (the MLPs,
        the pool's query / output projections) and, behind ``_allow_tensor_op_stem``, for CPU wiring tests."""
        from . import _enc_train
        b, d, t_len = strain.shape
        on_gpu = strain.device.type == "cuda"
        wants_grad = torch.is_grad_enabled() and (
            strain.requires_grad or (extra_tokens is not None and extra_tokens.requires_grad)
            or any(p.requires_grad for p in _enc_train.train_parameters(self)) or self.detector_embed.weight.requires_grad
            or self.pool_queries.requires_grad)
        n_extra = 0 if extra_tokens is None else extra_tokens.shape[1]
        n_total = d * 61 + n_extra
        clean = None
        if self._allow_tensor_op_stem and not on_gpu:
            return self._compute_feats_tensor_ops(strain, asd_bands, extra_tokens)
        if not on_gpu:
            raise _lib.PfError(f"LeanStrainEncoder runs on the MI355X only (input on {strain.device}); no CPU fallback")
        fused_ok = (self.precision == "bf16" and not self.training and not wants_grad and n_total <= 192
                    and self._mixer_supported() and self.n_energy_windows == 16)
        if not fused_ok:
            if not _enc_train.supported(self, n_total, t_len):
                raise NotImplementedError("the HIP training path is built for d_model 192, 3 layers x 6 heads, FFN 768, 8 pool "
                                          "queries, 16 energy windows, 16384-sample segments and at most 192 tokens")
            e = self.stem[6].out_channels
            # positional + detector embedding of every conv token of an event (LN:218-222); geometry tokens get none
            tok_bias = (self.pos(61)[None] + self.detector_embed.weight[:d, None, :]).reshape(d * 61, e).float()
            if n_extra:
                tok_bias = torch.cat([tok_bias.new_zeros(n_extra, e), tok_bias], dim=0)
            pooled, log_energy = _enc_train.encode_tokens(self, strain, None if extra_tokens is None else extra_tokens.float(),
                                                            tok_bias, training=self.training)
            pooled = self._pool_out(pooled)
        else:
            tok, log_energy = self._stem_hip(strain)       # sanitises in-kernel
            n_tok, e = tok.shape[1], tok.shape[2]
            tok_bias = (self.pos(n_tok)[None] + self.detector_embed.weight[:d, None, :]).reshape(d * n_tok, e)
            # fused HIP token mixer: embedding add + 3 Transformer layers + pool attention
            tok = tok.float().reshape(b, d * n_tok, e)
            if n_extra:
                tok = torch.cat([extra_tokens.float(), tok], dim=1)
                tok_bias = torch.cat([tok_bias.new_zeros(n_extra, e), tok_bias.float()], dim=0)
            pooled = self._mix_hip(tok.contiguous(), tok_bias)
        energy = self._seq("energy_mlp", log_energy.reshape(b, -1))
        parts = [pooled.reshape(b, -1), energy]
        if self.psd_bands > 0:
            if asd_bands is None:
                asd_bands = strain.new_zeros(b, self.n_detectors, self.psd_bands)
            parts.append(self._seq("noise_mlp", asd_bands.reshape(b, -1)))
        return torch.cat(parts, dim=1), clean

    def _compute_feats_tensor_ops(self, strain, asd_bands=None, extra_tokens=None):
        """the same function on device tensor ops: CPU wiring tests only (``_allow_tensor_op_stem``)"""
        b, d, _ = strain.shape
        clean = self._sanitize(strain)
        tok, log_energy = self._stem(clean), self._window_log_energy(clean)
        energy = self.energy_mlp(log_energy.reshape(b, -1))
        n_tok, e = tok.shape[1], tok.shape[2]
        tok_bias = (self.pos(n_tok)[None] + self.detector_embed.weight[:d, None, :]).reshape(d * n_tok, e)
        tok = (tok.reshape(b, d, n_tok, e) + tok_bias.reshape(1, d, n_tok, e)).reshape(b, d * n_tok, e)
        if extra_tokens is not None:
            tok = torch.cat([extra_tokens, tok], dim=1)
        tok = self._fuse(tok)
        pooled, _ = self.pool_attn(self.pool_queries.unsqueeze(0).expand(b, -1, -1), tok, tok)
        parts = [pooled.reshape(b, -1), energy]
        if self.psd_bands > 0:
            if asd_bands is None:
                asd_bands = strain.new_zeros(b, self.n_detectors, self.psd_bands)
            parts.append(self.noise_mlp(asd_bands.reshape(b, -1)))
        return torch.cat(parts, dim=1), clean

    def _autocast(self, dev):
        """The [B, .]-sized tensor-op remainder (the pool's output projection, the energy / noise MLPs, the context head:
        < 2 GFLOP per 1024 events) runs in fp32 in BOTH precisions: under bf16 autocast its forward + backward were ~70
        launches of which 27 were dtype casts of weights and activations (130 us of a 15 ms training step) for GEMMs of
        5-15 us each.  The precision switch acts on the HIP kernels (stem, token mixer, flow)."""
        return torch.autocast("cuda", enabled=False)

    def _empty(self, strain):
        """no events: [0, context_dim] (the reference's tensor ops give the same), nothing is launched"""
        return strain.new_zeros(0, self._out_features(), dtype=torch.float32)

    def _out_features(self) -> int:
        last = self.out_proj[-1] if isinstance(self.out_proj, nn.Sequential) else self.out_proj
        return last.out_features if hasattr(last, "out_features") else last.normalized_shape[0]

    def forward(self, strain, asd_bands=None):
        if strain.shape[0] == 0:
            return self._empty(strain)
        with self._autocast(strain.device):
            feats, _ = self._compute_feats(strain, asd_bands)
            return self._seq("out_proj", feats).float()


_SR, _T_LEN, _F_LO, _F_HI = 4096, 16384, 20.0, 1024.0


class CoherentEncoder(LeanStrainEncoder):
    """LeanStrainEncoder plus frequency-domain geometry tokens (CE:42-123): band energies,
    power-weighted pair coherence (|g|, cos, sin), GCC delay + sharpness, log-amplitude ratio
    -> MLP -> 4 tokens prepended to the fusion transformer."""

    def __init__(self, geometry_bands: int = 16, geom_hidden: int = 128, n_geom_tokens: int = 4,
                 tau_max_ms: float = 30.0, **kw):
        super().__init__(**kw)
        self.K, self.n_geom_tokens = int(geometry_bands), int(n_geom_tokens)
        self.d_model = self.detector_embed.embedding_dim
        self.n_rfft = _T_LEN // 2 + 1
        freqs = np.fft.rfftfreq(_T_LEN, 1.0 / _SR)
        keep = (freqs >= _F_LO) & (freqs < _F_HI)
        self.band_lo, self.Nf = int(np.argmax(keep)), int(keep.sum())
        fb = freqs[keep]
        edges = np.geomspace(_F_LO, _F_HI, self.K + 1)
        member = np.stack([(fb >= edges[k]) & (fb < edges[k + 1]) for k in range(self.K)]).astype(np.float32)
        self.register_buffer("Bsum", torch.from_numpy(member))
        self.register_buffer("bcount", torch.from_numpy(member).sum(1).clamp_min(1.0))
        self.maxlag = int(tau_max_ms * 1e-3 * _SR)
        self.register_buffer("lags_norm", torch.arange(-self.maxlag, self.maxlag + 1).float() / self.maxlag)
        self.pairs = [(i, j) for i in range(self.n_detectors) for j in range(i + 1, self.n_detectors)]
        rel_dim = self.n_detectors * self.K + len(self.pairs) * (3 * self.K + 3)
        self.geom_mlp = nn.Sequential(nn.Linear(rel_dim, geom_hidden), nn.GELU(),
                                      nn.Linear(geom_hidden, geom_hidden), nn.GELU())
        self.geom_to_tokens = nn.Linear(geom_hidden, self.n_geom_tokens * self.d_model)

    def _geometry_plan(self):
        """band b as a contiguous range of the kept bins (geomspace bands are intervals), or None when they are not"""
        plan = self.__dict__.get("_geom_plan")
        if plan is None:
            m = self.Bsum.detach().cpu().numpy() > 0
            edges, ok = [0], self.K <= 16 and m.sum(0).max() <= 1
            for k in range(self.K):
                idx = np.flatnonzero(m[k])
                lo, hi = (int(idx[0]), int(idx[-1]) + 1) if idx.size else (edges[-1], edges[-1])
                ok = ok and lo == edges[-1] and hi - lo == idx.size
                edges.append(hi)
            ok = (ok and edges[-1] == self.Nf and _T_LEN == 16384 and self.band_lo >= 1 and self.band_lo + self.Nf <= 4096
                  and 1 <= self.maxlag <= 127 and self.n_detectors <= 8)
            plan = self.__dict__["_geom_plan"] = (edges if ok else False)
        return plan or None

    def _geometry_rel_hip(self, clean, edges, sanitize=False):
        """the same features by pf_geom_features (csrc/pf_geom.hip): both transforms in LDS, one launch per stage;
        ``sanitize``: ``clean`` is the raw strain, sanitised inside the kernel (LN:207)"""
        from . import _lib
        b, dev = clean.shape[0], clean.device
        tw = self.__dict__.get("_geom_twiddle")
        if tw is None or tw.device != dev:
            host = torch.empty(8192, 2, dtype=torch.float32)
            _lib.check(_lib.lib().pf_geom_twiddles(host.data_ptr()), "pf_geom_twiddles")
            tw = self.__dict__["_geom_twiddle"] = host.to(dev)
        x = clean.float().contiguous()
        a = _lib.PfGeomArgs()
        a.clean, a.batch, a.n_det = x.data_ptr(), b, self.n_detectors
        a.band_lo, a.nf, a.n_bands, a.maxlag = self.band_lo, self.Nf, self.K, self.maxlag
        for i, e in enumerate(edges):
            a.band_edge[i] = e
        spec = torch.empty(b, self.n_detectors, self.Nf, 2, dtype=torch.float32, device=dev)
        etot = torch.empty(b, self.n_detectors, dtype=torch.float32, device=dev)
        rel = torch.empty(b, self.n_detectors * self.K + len(self.pairs) * (3 * self.K + 3), dtype=torch.float32, device=dev)
        a.twiddle, a.spec, a.etot, a.rel = tw.data_ptr(), spec.data_ptr(), etot.data_ptr(), rel.data_ptr()
        a.sanitize = 1 if sanitize else 0
        _lib.check(_lib.lib().pf_geom_features(a, torch.cuda.current_stream(dev).cuda_stream), "pf_geom_features")
        return rel

    def _geometry_rel(self, clean):
        b = clean.shape[0]
        if clean.is_cuda and not (torch.is_grad_enabled() and clean.requires_grad):
            # (the strain is data: no caller differentiates the features with respect to it; if one does, the tensor ops below)
            edges = self._geometry_plan()
            if edges is not None:
                return self._geometry_rel_hip(clean, edges)
        if clean.is_cuda and not self.__dict__.get("_fft_route_logged"):
            # not the default path: say so once -- taken when the strain requires a gradient or the band
            # membership is not a partition into intervals (not constructible through the reference's constructor)
            self.__dict__["_fft_route_logged"] = True
            import logging
            logging.getLogger(__name__).warning("posteriflow_amd: CoherentEncoder geometry features by torch.fft tensor ops "
                                                "(strain requires grad, or non-interval bands); pf_geom_features not used")
        spec = torch.fft.rfft(clean.float().contiguous(), norm="ortho", dim=-1)
        spec = spec[..., self.band_lo: self.band_lo + self.Nf]
        re, im = spec.real, spec.imag
        power = re * re + im * im
        amp = torch.sqrt(power + 1e-12)
        feats = [torch.log(power @ self.Bsum.T / self.bcount + 1e-8).reshape(b, -1)]
        for i, j in self.pairs:
            xr = re[:, i] * re[:, j] + im[:, i] * im[:, j]
            xi = im[:, i] * re[:, j] - re[:, i] * im[:, j]
            den = (amp[:, i] * amp[:, j]) @ self.Bsum.T + 1e-8
            gr, gi = xr @ self.Bsum.T / den, xi @ self.Bsum.T / den
            gm = torch.sqrt(gr * gr + gi * gi) + 1e-8
            feats += [gm, gr / gm, gi / gm]
            full = torch.zeros(b, self.n_rfft, dtype=torch.complex64, device=clean.device)
            full[:, self.band_lo: self.band_lo + self.Nf] = torch.complex(xr, xi)
            cc = torch.fft.irfft(full, n=_T_LEN, dim=-1)
            a = torch.cat([cc[:, -self.maxlag:], cc[:, : self.maxlag + 1]], dim=1).abs()
            feats += [self.lags_norm[a.argmax(-1)].unsqueeze(-1),
                      (a.max(-1).values / (a.mean(-1) + 1e-8)).unsqueeze(-1)]
            ei, ej = power[:, i].sum(-1), power[:, j].sum(-1)
            feats.append((torch.log(ei + 1e-8) - torch.log(ej + 1e-8)).unsqueeze(-1))
        return torch.cat(feats, dim=-1)

    def forward(self, strain, asd_bands=None):
        if strain.shape[0] == 0:
            return self._empty(strain)
        edges = self._geometry_plan() if (strain.is_cuda and not (torch.is_grad_enabled() and strain.requires_grad)) else None
        if edges is not None:      # GPU: both consumers sanitise the raw strain on load (no extra pass over it)
            clean = strain
            rel = self._geometry_rel_hip(strain, edges, sanitize=True)
        else:
            clean = self._sanitize(strain)
            rel = self._geometry_rel(clean)                  # FFT features stay fp32
        with self._autocast(strain.device):
            g = self._seq("geom_mlp", rel)
            gtok = self._seq("geom_to_tokens", g).reshape(-1, self.n_geom_tokens, self.d_model)
            feats, _ = self._compute_feats(clean, asd_bands, extra_tokens=gtok)
            return self._seq("out_proj", feats).float()


class LeanNPE(nn.Module):
    """Encoder + signal-rank embedding + NSF flow, pure NLL (LN:255-338)."""

    def __init__(self, param_names: List[str] = PARAM_NAMES, context_dim: int = 256, rank_dim: int = 32,
                 max_signals: int = 5, flow_layers: int = 10, flow_hidden: int = 256, flow_bins: int = 16,
                 encoder_kwargs: Optional[dict] = None, premerger: bool = False, psd_cond: bool = False,
                 psd_bands: int = 16, encoder_type: str = "conv"):
        super().__init__()
        self.param_names, self.max_signals = list(param_names), max_signals
        self.context_dim, self.encoder_type = context_dim, encoder_type
        self.psd_cond = psd_cond or encoder_type == "coherent"           # LN:271
        self.scaler = ParamScaler(self.param_names, premerger=premerger)
        kw = dict(encoder_kwargs or {})
        if encoder_type == "coherent":
            self.encoder = CoherentEncoder(context_dim=context_dim, psd_bands=psd_bands, **kw)
        else:
            if psd_cond:
                kw["psd_bands"] = psd_bands
            self.encoder = LeanStrainEncoder(context_dim=context_dim, **kw)
        self.rank_embed = nn.Embedding(max_signals, rank_dim)
        self.flow = NSFPosteriorFlow(features=len(self.param_names), context_features=context_dim + rank_dim,
                                     hidden_features=flow_hidden, num_layers=flow_layers, num_bins=flow_bins,
                                     tail_bound=5.0, dropout=0.0, temperature_scale=1.0,
                                     use_masked_context=False)
        self.flow.temperature.requires_grad_(False)                     # LN:297

    def set_precision(self, precision: str) -> "LeanNPE":
        """"fp32" (default: f32 MFMA + fp32 tensor ops, matches the CPU path to ~1e-5) or "bf16" (throughput:
        bf16 MFMA operands with fp32 accumulation in the stem, the token mixer and the flow)."""
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"precision must be 'fp32' or 'bf16', got {precision!r}")
        self.encoder.precision = precision
        self.flow.precision = precision
        return self

    def flatten_parameters(self) -> "LeanNPE":
        """One flat leaf for the flow and one for the encoder's HIP-trained parameters (``NSFPosteriorFlow.flatten_parameters``,
        ``LeanStrainEncoder.flatten_parameters``): a training step's autograd, gradient clipping and AdamW then handle ~25
        tensors instead of ~260.  ``state_dict`` keys are unchanged.  Call it before building the optimiser."""
        self.flow.flatten_parameters()
        self.encoder.flatten_parameters()
        return self

    def _full_context(self, context: torch.Tensor, rank: torch.Tensor) -> torch.Tensor:
        w = self.rank_embed.weight
        if w.requires_grad and torch.is_grad_enabled():
            # the same rows as rank_embed(rank), as a row gather: its backward is one index_add (float atomics into 5 x 32
            # numbers) instead of the embedding's sort-and-segment-reduce (about 20 launches)
            emb = torch.index_select(w, 0, rank)
        else:
            emb = self.rank_embed(rank)
        return torch.cat([context, emb], dim=1)

    def encode(self, strain, asd_bands=None):
        return self.encoder(strain, asd_bands) if self.psd_cond else self.encoder(strain)

    def nll(self, strain, params_phys, rank, context=None, asd_bands=None):
        """[B] negative log-likelihood of physical parameters (LN:306-316)."""
        if context is None:
            context = self.encode(strain, asd_bands)
        ctx = self._full_context(context, rank)
        y = self.scaler.normalize(params_phys)
        return self.flow.compute_psd_aware_nll(y, ctx, None)            # log_sigma = 0 fused in-kernel

    @torch.no_grad()
    def sample_posterior(self, strain, rank: int = 0, n_samples: int = 256, asd_bands=None):
        """[B, n_samples, 11] posterior draws in physical units (LN:318-332); the per-event
        context is passed once, not replicated n_samples times."""
        context = self.encode(strain, asd_bands)
        b = context.shape[0]
        r = torch.full((b,), rank, dtype=torch.long, device=context.device)
        ctx = self._full_context(context, r)
        z = torch.randn(b * n_samples, len(self.param_names), device=context.device)
        y, _ = self.flow.inverse(z, ctx)
        return self.scaler.denormalize(self.scaler.wrap(y).reshape(b, n_samples, -1))

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .cpu(): keep the (non-module) scaler on the parameters' device (LN:334-338)
        out = super()._apply(fn, *args, **kwargs)
        self.scaler.to(next(self.parameters()).device)
        return out


_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def _side_stream(device) -> "torch.cuda.Stream":
    i = device.index if device.index is not None else torch.cuda.current_device()
    if i not in _SIDE_STREAMS:
        _SIDE_STREAMS[i] = torch.cuda.Stream(device)
    return _SIDE_STREAMS[i]


def batch_nll(model: LeanNPE, strain, params, nsig, asd_bands=None, row_cap=None, reduction: str = "mean"):
    """Mean per-signal NLL of a batch of events with up to ``max_signals`` signals each
    (experiments/train_lean_npe.py:108-127), as ONE flow call over all (event, rank) pairs with a
    0/1 weight for rank < nsig, instead of looping over ranks with a boolean-index host sync per rank
    (SURVEY H6): sum / count is identical, shapes are static (every step launches the same kernels) and nothing
    waits on the host.

    ``row_cap``: a STATIC upper bound on the number of (event, rank) pairs that exist in a batch (e.g. 2 x batch for
    the reference's mix of 1-5 signals, mean ~1.8).  The existing pairs are moved to the front by a stable sort (no
    host sync) and only the first ``row_cap`` rows go through the flow, forward and backward -- instead of
    ``max_signals`` = 5 rows per event whatever ``nsig`` is.  Pairs beyond the cap would be dropped from the mean:
    ``batch_nll.last_overflow`` (a device scalar, read it when convenient) counts them.

    ``row_cap="exact"``: what the reference's loop evaluates -- exactly the existing pairs and nothing else.  Their indices
    come from ONE ``nonzero`` on the [B, max_signals] mask (one host sync on data the remix kernel produced, taken BEFORE
    the encoder is queued so that nothing waits behind it; the reference syncs once per rank); the flow then runs on
    ~1.8 rows per event instead of 5, forward and backward.

    ``reduction``: "mean" (default) is the reference's scalar ``sum nll / sum nsig`` over THIS call's events.  "sum"
    returns the pair ``(sum nll, number of pairs)`` (two 0-dim tensors, the count without a graph) -- what a data-parallel
    step needs: the reference's loss over the GLOBAL batch is ``sum_ranks(sum nll) / sum_ranks(count)``, which is not the
    mean of the ranks' local means when the ranks hold different numbers of signals (``train.train_step``)."""
    if reduction not in ("mean", "sum"):
        raise ValueError(f"reduction must be 'mean' or 'sum', got {reduction!r}")

    def _out(total, count):
        if reduction == "sum":
            return total, count.detach().to(total.dtype)
        return total / count.to(total.dtype).clamp_min(1)

    b, r_max = params.shape[0], params.shape[1]
    ranks = torch.arange(r_max, device=nsig.device)[None, :].expand(b, r_max)
    keep = (ranks < nsig[:, None]).reshape(-1)
    if isinstance(row_cap, str):
        if row_cap != "exact":
            raise ValueError(f"row_cap must be None, an int or 'exact', got {row_cap!r}")
        # The encoder is queued FIRST; the pair indices are resolved on a side stream that waits only for what was queued
        # before this call (the remix kernel), so the host sync of nonzero() returns while the encoder is running and
        # the GPU never idles behind it (in program order "nonzero, then encode" the GPU waited ~0.4 ms for the host).
        if keep.is_cuda:
            main = torch.cuda.current_stream(strain.device)
            entry = torch.cuda.Event()
            entry.record(main)
            context = model.encode(strain, asd_bands)
            side = _side_stream(strain.device)
            side.wait_event(entry)
            with torch.cuda.stream(side):
                idx = torch.nonzero(keep).squeeze(1)             # (host sync on `side`: the number of pairs is a shape)
            keep.record_stream(side)
            main.wait_stream(side)
            idx.record_stream(main)
        else:                                                    # (host tensors: the stand-in models of the CPU tests)
            context = model.encode(strain, asd_bands)
            idx = torch.nonzero(keep).squeeze(1)
        count = torch.full((), float(idx.numel()), device=context.device)
        if idx.numel() == 0:
            return _out(context.sum() * 0.0, count)
        event = torch.div(idx, r_max, rounding_mode="floor")
        nll = model.nll(None, params.reshape(b * r_max, -1)[idx], ranks.reshape(-1)[idx],
                        context=torch.index_select(context, 0, event))  # (backward: one index_add, not a sort)
        return _out(nll.sum(), count) if reduction == "sum" else nll.mean()
    context = model.encode(strain, asd_bands)
    # rows of absent ranks are all-zero labels (remix_data.py:229): give them a valid stand-in (rank 0 of
    # the same event) so that the unused rows stay finite; their weight is 0
    rows = torch.where(keep[:, None], params.reshape(b * r_max, -1), params[:, :1].expand(-1, r_max, -1).reshape(b * r_max, -1))
    rank_flat = ranks.reshape(-1)
    if row_cap is not None and row_cap < b * r_max:
        order = torch.argsort((~keep).to(torch.int8), stable=True)[:row_cap]       # existing pairs first, static length
        batch_nll.last_overflow = (keep.sum() - keep[order].sum()).detach()
        event = order // r_max
        nll = model.nll(None, rows[order], rank_flat[order], context=context[event])
        kept = keep[order]
        return _out(torch.where(kept, nll, torch.zeros_like(nll)).sum(), kept.sum())
    nll = model.nll(None, rows, rank_flat, context=context.repeat_interleave(r_max, dim=0))
    return _out(torch.where(keep, nll, torch.zeros_like(nll)).sum(), keep.sum())
