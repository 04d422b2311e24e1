"""Differentiable strain embedding on the HIP training path (``pf_embed_train_forward`` / ``_backward``).

One ``torch.autograd.Function`` spans the stem, the token assembly, the three Transformer layers and the K / V side of the
attention pool of ``LeanStrainEncoder._compute_feats`` (reference ``src/ahsd/models/lean_npe.py:199-233``): the forward is
one C call that keeps its activations in a workspace, the backward is one C call that returns the gradient of every
parameter in ONE flat buffer (the per-parameter gradients handed to autograd are views into it).  Nothing here computes on
the host; the module raises off the GPU."""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional

import torch

from . import _lib

D_MODEL, N_POOL_Q, TOK_PER_DET = 192, 8, 61
FWD_ONLY_CHUNK = 1024      # events per launch group of a no-grad call (workspace ~5.5 MB per 3-detector event in fp32)


def train_parameters(enc) -> List[torch.nn.Parameter]:
    """the encoder's parameters in the raw layout of pf_embed_train_* (include/pf_hip.h); in flat mode
    (``enc.flatten_parameters()``) these are views of the one leaf ``enc._theta``"""
    out = [p for i in (0, 2, 4, 6) for p in (enc.stem[i].weight, enc.stem[i].bias)]
    for layer in enc.fusion.layers:
        out += [layer.norm1.weight, layer.norm1.bias, layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias,
                layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias, layer.norm2.weight, layer.norm2.bias,
                layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias]
    return out + [enc.pool_attn.in_proj_weight, enc.pool_attn.in_proj_bias]


def supported(enc, n_tokens: int, t_len: int) -> bool:
    a = enc.fusion.layers[0]
    return (len(enc.fusion.layers) == 3 and a.self_attn.embed_dim == D_MODEL and a.self_attn.num_heads == 6
            and a.linear1.out_features == 768 and tuple(enc.pool_queries.shape) == (N_POOL_Q, D_MODEL)
            and enc.pool_attn.num_heads == 6 and enc.fusion.norm is None and enc.n_energy_windows == 16
            and enc.stem[6].out_channels == D_MODEL and t_len == 16384 and 1 <= n_tokens <= 192
            and all(abs(l.dropout.p - a.dropout.p) < 1e-12 and abs(l.dropout1.p - a.dropout.p) < 1e-12
                    and abs(l.dropout2.p - a.dropout.p) < 1e-12 and abs(l.self_attn.dropout - a.dropout.p) < 1e-12
                    for l in enc.fusion.layers))


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _draw_seed() -> int:
    """a fresh 64-bit dropout seed from torch's CPU generator (follows torch.manual_seed, no device sync)"""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


class _EncoderTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, precision, n_det, training, dropout_p, seed, fwd_only, strain, extra, token_bias, pool_q, *params):
        L, dev = _lib.lib(), strain.device
        prec = _lib.PRECISIONS[precision]
        b = strain.shape[0]
        n_extra = 0 if extra is None else extra.shape[1]
        desc = _lib.PfEmbedTrainDesc(prec, n_det, n_extra, 1 if training else 0, float(dropout_p), 1 if fwd_only else 0, int(seed))
        key = (_lib.param_epoch(), dev, prec, tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
        if state.get("key") != key:        # flat fp32 copy of the parameters + their MFMA fragments, once per weight update
            if len(params) == 1:           # flat mode: the leaf IS the raw layout
                raw = params[0].detach()
            else:
                raw = torch.cat([p.detach().reshape(-1).float() for p in params])
            assert raw.numel() == L.pf_embed_train_raw_param_count()
            packed = torch.empty(L.pf_embed_train_packed_bytes(prec), dtype=torch.uint8, device=dev)
            _lib.check(L.pf_embed_train_pack(prec, raw.data_ptr(), packed.data_ptr(), _stream(dev)), "pf_embed_train_pack")
            state["key"], state["raw"], state["packed"] = key, raw, packed
        raw, packed = state["raw"], state["packed"]
        need = L.pf_embed_train_workspace_bytes(C.byref(desc), b)
        if need < 0:
            raise NotImplementedError("pf_embed_train_workspace_bytes: unsupported geometry")
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
        x = strain.reshape(b * n_det, -1).contiguous().float()
        ex = None if extra is None else extra.detach().contiguous().float()
        tb = None if token_bias is None else token_bias.detach().contiguous().float()
        q = pool_q.detach().contiguous().float()
        pooled = torch.empty(b, N_POOL_Q, D_MODEL, dtype=torch.float32, device=dev)
        log_energy = torch.empty(b * n_det, 16, dtype=torch.float32, device=dev)
        _lib.check(L.pf_embed_train_forward(C.byref(desc), packed.data_ptr(), raw.data_ptr(), x.data_ptr(),
                                            0 if ex is None else ex.data_ptr(), 0 if tb is None else tb.data_ptr(), q.data_ptr(), b,
                                            pooled.data_ptr(), log_energy.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)),
                   "pf_embed_train_forward")
        log_energy = log_energy.reshape(b, n_det, 16)
        if fwd_only:                          # no backward will follow: nothing is kept (the workspace dies here)
            ctx.held = None
            return pooled, log_energy
        ctx.desc, ctx.b, ctx.n_extra, ctx.n_det = desc, b, n_extra, n_det
        ctx.shapes = [p.shape for p in params]
        ctx.has_tb = token_bias is not None
        ctx.held = (raw, packed, ws, q)       # plain attributes: none of them is an input or output of this Function
        ctx.mark_non_differentiable(log_energy)
        return pooled, log_energy

    @staticmethod
    def backward(ctx, g_pooled, _g_log_energy):
        L = _lib.lib()
        if ctx.held is None:
            raise RuntimeError("the strain embedding's forward ran forward-only (no input required a gradient when it was called)")
        raw, packed, ws, q = ctx.held
        dev = raw.device
        gp = g_pooled.contiguous().float()
        t = ctx.n_extra + TOK_PER_DET * ctx.n_det
        g_raw = torch.empty(raw.numel(), dtype=torch.float32, device=dev)
        g_extra = torch.empty(ctx.b, ctx.n_extra, D_MODEL, dtype=torch.float32, device=dev) if ctx.n_extra else None
        g_tb = torch.empty(t, D_MODEL, dtype=torch.float32, device=dev) if ctx.has_tb else None
        g_q = torch.empty(N_POOL_Q, D_MODEL, dtype=torch.float32, device=dev)
        _lib.check(L.pf_embed_train_backward(C.byref(ctx.desc), packed.data_ptr(), raw.data_ptr(), q.data_ptr(), gp.data_ptr(), ctx.b,
                                             ws.data_ptr(), ws.numel(), g_raw.data_ptr(), 0 if g_extra is None else g_extra.data_ptr(),
                                             0 if g_tb is None else g_tb.data_ptr(), g_q.data_ptr(), _stream(dev)),
                   "pf_embed_train_backward")
        ctx.held = None
        grads, off = [], 0
        for shp in ctx.shapes:                # views into the flat gradient buffer, no copies
            n = math.prod(shp)
            grads.append(g_raw[off:off + n].view(shp))
            off += n
        return (None, None, None, None, None, None, None, None, g_extra, g_tb, g_q, *grads)


def encode_tokens(enc, strain: torch.Tensor, extra_tokens: Optional[torch.Tensor], token_bias: Optional[torch.Tensor],
                  training: bool, seed: Optional[int] = None):
    """(pooled [B, 8, 192] before pool_attn.out_proj, log_energy [B, D, 16]) through the HIP training path."""
    if strain.device.type != "cuda":
        raise _lib.PfError(f"the strain embedding runs on the MI355X only (input on {strain.device}); no CPU fallback")
    with torch.autocast("cuda", enabled=False):
        q = enc._pool_queries_projected()
    p = float(enc.fusion.layers[0].dropout.p)
    train = bool(training and p > 0.0)
    if train and seed is None:
        seed = _draw_seed()
    state = enc.__dict__.setdefault("_train_state", {})
    state["last_seed"] = seed if train else None      # (tests rebuild the dropout factors from it)
    params = [enc._theta] if getattr(enc, "_theta", None) is not None else train_parameters(enc)
    if strain.requires_grad and torch.is_grad_enabled():
        # the backward has no d/d strain (the strain is data in every reference caller): refuse rather than hand autograd a
        # silent zero through the stem
        raise NotImplementedError("the HIP strain embedding does not differentiate with respect to the strain; detach it")
    diff = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (extra_tokens, token_bias, q, *params))
    if diff:
        return _EncoderTrainFn.apply(state, enc.precision, strain.shape[1], train, p, seed or 0, False, strain, extra_tokens,
                                     token_bias, q, *params)
    # No gradient will be asked for (torch.no_grad(), or frozen parameters): the forward-only workspace (one layer's
    # activations, none of the backward's temporaries), in chunks of FWD_ONLY_CHUNK events so that a 4096-event fp32 call
    # needs ~5.6 GB of workspace instead of ~65 GB.  Dropout indices are per call: a train()-mode no-grad call over several
    # chunks reuses the factors per chunk, which is as random as nn.Dropout's fresh draw.
    outs = []
    for lo in range(0, strain.shape[0], FWD_ONLY_CHUNK):
        hi = min(strain.shape[0], lo + FWD_ONLY_CHUNK)
        with torch.no_grad():
            outs.append(_EncoderTrainFn.apply(state, enc.precision, strain.shape[1], train, p, (seed or 0) + lo, True, strain[lo:hi],
                                              None if extra_tokens is None else extra_tokens[lo:hi], token_bias, q, *params))
    if len(outs) == 1:
        return outs[0]
    return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
