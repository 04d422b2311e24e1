"""NSFPosteriorFlow on MI355X: the reference's flow API over the HIP C-ABI.

Drop-in for ``ahsd.models.flows`` (reference ``src/ahsd/models/flows.py``):
same constructor, methods, attributes and ``state_dict`` key names, but no
tensor arithmetic happens in Python -- parameters live in ``nn.Parameter``s
(for optimisers / checkpoints) and every evaluation is one call into
``libpfhip.so`` (``include/pf_hip.h``).  There is no CPU path: calling the flow
with the module or its inputs off the GPU raises.

Reference line numbers below are for ``src/ahsd/models/flows.py``.
"""
from __future__ import annotations

import logging
import math
import os
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import _lib

FLOW_NORM_BOUND = 3.0       # src/ahsd/models/parameter_scalers.py:27
_MIN_BIN = 1e-3             # nflows DEFAULT_MIN_{BIN_WIDTH,BIN_HEIGHT,DERIVATIVE}

_log = logging.getLogger(__name__)


# ----------------------------------------------------------------------------- #
# parameter containers with nflows' module / buffer names (never called)
# ----------------------------------------------------------------------------- #
def _hidden_degrees(n: int, features: int) -> torch.Tensor:
    return torch.arange(n) % max(1, features - 1) + min(1, features - 1)


class _MaskedLinear(nn.Linear):
    """Holds weight/bias + the ``mask`` / ``degrees`` buffers of nflows MaskedLinear.
    The kernels rebuild the same masks from the degree rule (csrc/pf_pack.hip)."""

    def __init__(self, in_degrees: torch.Tensor, out_features: int, features: int, is_output: bool):
        super().__init__(len(in_degrees), out_features)
        if is_output:
            deg = torch.arange(1, features + 1).repeat_interleave(out_features // features)
            mask = (deg[:, None] > in_degrees[None, :]).float()
        else:
            deg = _hidden_degrees(out_features, features)
            mask = (deg[:, None] >= in_degrees[None, :]).float()
        self.register_buffer("mask", mask)
        self.register_buffer("degrees", deg)


class _MaskedContextLinear(nn.Module):
    """Parameters of the reference's MaskedContextLinear (flows.py:112-183).  full_context=True (what
    MADEWithMaskedContext builds, flows.py:268): the mask buffer is all ones; full_context=False (flows.py:171-174):
    context block i is visible to hidden units of degree >= i.  The mask is folded into the weights when they are
    packed (``masked_weight``)."""

    def __init__(self, n_blocks: int, block_dim: int, hidden_degrees: torch.Tensor, full_context: bool = True):
        super().__init__()
        hidden = hidden_degrees.shape[0]
        self.weight = nn.Parameter(torch.empty(hidden, n_blocks * block_dim))
        self.bias = nn.Parameter(torch.zeros(hidden))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if full_context:
            in_deg = torch.full((n_blocks * block_dim,), -1, dtype=torch.long)
        else:
            in_deg = torch.arange(n_blocks).repeat_interleave(block_dim)
        self.full_context = bool(full_context)
        self.register_buffer("mask", (hidden_degrees[:, None] >= in_deg[None, :]).float())

    def masked_weight(self) -> torch.Tensor:
        return self.weight if self.full_context else self.weight * self.mask


class _ResidualBlock(nn.Module):
    def __init__(self, in_degrees, features, context_features, masked_blocks=None):
        super().__init__()
        h = len(in_degrees)
        if masked_blocks:
            self.context_layer = _MaskedContextLinear(masked_blocks[0], masked_blocks[1], in_degrees, *masked_blocks[2:])
        elif context_features:
            self.context_layer = nn.Linear(context_features, h)
        l0 = _MaskedLinear(in_degrees, h, features, False)
        l1 = _MaskedLinear(l0.degrees, h, features, False)
        self.linear_layers = nn.ModuleList([l0, l1])
        nn.init.uniform_(l1.weight, -1e-3, 1e-3)      # nflows zero_initialization
        nn.init.uniform_(l1.bias, -1e-3, 1e-3)


class _MADE(nn.Module):
    def __init__(self, features, hidden, context_features, multiplier, num_blocks=2, masked_blocks=None):
        super().__init__()
        self.initial_layer = _MaskedLinear(torch.arange(1, features + 1), hidden, features, False)
        if masked_blocks:
            self.context_layer = _MaskedContextLinear(masked_blocks[0], masked_blocks[1], self.initial_layer.degrees,
                                                      *masked_blocks[2:])
        elif context_features:
            self.context_layer = nn.Linear(context_features, hidden)
        self.blocks = nn.ModuleList(
            [_ResidualBlock(self.initial_layer.degrees, features, context_features, masked_blocks)
             for _ in range(num_blocks)])
        self.final_layer = _MaskedLinear(self.initial_layer.degrees, features * multiplier,
                                         features, True)

    def ordered_parameters(self, for_packing: bool = False) -> List[torch.Tensor]:
        """Raw layout of include/pf_hip.h ("raw parameter layout").  for_packing: the tensors pf_flow_pack reads --
        a masked-context linear built with full_context=False contributes weight * mask (the x-side masks are
        rebuilt by the pack map from the degree rule; the context mask of that variant is not)."""
        cw = lambda m: m.masked_weight() if for_packing and isinstance(m, _MaskedContextLinear) else m.weight
        out = [self.initial_layer.weight, self.initial_layer.bias]
        if hasattr(self, "context_layer"):
            out += [cw(self.context_layer), self.context_layer.bias]
        for b in self.blocks:
            if hasattr(b, "context_layer"):
                out += [cw(b.context_layer), b.context_layer.bias]
            for lin in b.linear_layers:
                out += [lin.weight, lin.bias]
        return out + [self.final_layer.weight, self.final_layer.bias]


class _SplineLayer(nn.Module):
    def __init__(self, features, hidden, context_features, num_bins, masked_blocks=None):
        super().__init__()
        self.autoregressive_net = _MADE(features, hidden, context_features, 3 * num_bins - 1,
                                        masked_blocks=masked_blocks)


class _Reverse(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.register_buffer("_permutation", torch.arange(features - 1, -1, -1))


class _Composite(nn.Module):
    def __init__(self, transforms):
        super().__init__()
        self._transforms = nn.ModuleList(transforms)


class _FlowAlias(nn.Module):
    """nflows ``Flow(transform, base)`` registers the same modules a second time
    (flows.py:532), so checkpoints carry ``flow._transform.*`` too."""

    def __init__(self, transform):
        super().__init__()
        self._transform = transform


# ----------------------------------------------------------------------------- #
class PSDScaledNormal(nn.Module):
    """Base density N(0, diag(exp(log_sigma))^2) (flows.py:28-109).  ``log_prob``
    is two fused elementwise device ops; inside the flow's NLL it is fused into
    the kernel epilogue instead (``pf_flow_forward(log_sigma=...)``)."""

    def __init__(self, shape: Union[list, int]):
        super().__init__()
        self.shape = list(shape) if isinstance(shape, (list, tuple)) else [shape]
        self.dim = self.shape[0]

    def log_prob(self, z: torch.Tensor, log_sigma_psd: torch.Tensor) -> torch.Tensor:
        if z.shape != log_sigma_psd.shape:
            raise ValueError(f"Shape mismatch: z {z.shape} vs log_sigma_psd {log_sigma_psd.shape}")
        quad = (z * torch.exp(-log_sigma_psd)).square().sum(dim=1)
        return -0.5 * (quad + 2 * log_sigma_psd.sum(dim=1) + self.dim * math.log(2 * math.pi))

    def sample(self, num_samples: int, log_sigma_psd: Optional[torch.Tensor] = None) -> torch.Tensor:
        b = log_sigma_psd.shape[0] if log_sigma_psd is not None else 1
        dev = log_sigma_psd.device if log_sigma_psd is not None else torch.device("cpu")
        dt = log_sigma_psd.dtype if log_sigma_psd is not None else torch.float32
        return torch.randn(b, num_samples, self.dim, device=dev, dtype=dt)


# ----------------------------------------------------------------------------- #
class _Packed:
    """Device-side packed weights for one precision + the key they were built from."""

    def __init__(self):
        self.buf = None
        self.map = None
        self.key = None


def _dev_ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class NSFPosteriorFlow(nn.Module):
    """Masked-autoregressive rational-quadratic-spline flow (flows.py:363-939).

    ``precision``: ``"fp32"`` (exact f32 MFMA, matches the CPU path to ~1e-6) or
    ``"bf16"`` (bf16 MFMA operands, fp32 accumulate: throughput mode).  Default
    from ``$PF_FLOW_PRECISION`` or ``"fp32"``.

    ``hoist_context``: evaluate the context projections of all layers once per call in one
    GEMM instead of inside every layer / autoregressive pass (same arithmetic, projections kept
    in fp32).  ``None`` (default) = measured
    choice: hoisted for the inverse / sampling (the projections would otherwise be recomputed
    in all D passes), in-layer for the forward / density path (108 us vs 86 + 33 us at B = 4096).
    """

    def __init__(self, features: int, context_features: int = 0, hidden_features: int = 256,
                 num_layers: int = 12, num_bins: int = 16,
                 tail_bound: Union[float, Dict[int, float]] = FLOW_NORM_BOUND,
                 max_overlaps: int = 6, dropout: float = 0.0, temperature_scale: float = 1.5,
                 use_masked_context: Optional[bool] = None, **kwargs):
        super().__init__()
        self.features = features
        self.context_features = context_features
        self.hidden_features = hidden_features
        self.num_layers = num_layers
        self.num_bins = num_bins
        self.max_overlaps = max_overlaps
        self.dropout = dropout
        self.logger = _log
        if use_masked_context is None:                                   # flows.py:409-411
            use_masked_context = context_features > 0 and features > 0 and context_features % features == 0
        self.use_masked_context = bool(use_masked_context)
        if self.use_masked_context:                                      # flows.py:412-421
            if context_features <= 0 or context_features % features != 0:
                raise ValueError("use_masked_context needs context_features to be a positive multiple of features")
            self.n_context_blocks = features
            self.context_block_dim = context_features // features
        else:
            self.n_context_blocks = self.context_block_dim = None
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError(f"dropout must be in [0, 1), got {dropout}")
        # flows.py:425-435 -- per-parameter dict is kept for the API, its patching is a
        # no-op in the reference (flows.py:502), so only the global bound takes effect.
        if isinstance(tail_bound, dict):
            self.per_param_tail_bounds = tail_bound
            self.tail_bounds_list = [tail_bound.get(i, FLOW_NORM_BOUND) for i in range(features)]
        else:
            self.per_param_tail_bounds = None
            self.tail_bounds_list = [tail_bound] * features
        # flows.py:471/517: a non-float bound silently becomes FLOW_NORM_BOUND
        self._tail_bound = tail_bound if isinstance(tail_bound, float) else FLOW_NORM_BOUND

        self.temperature = nn.Parameter(torch.tensor(temperature_scale, dtype=torch.float32))
        self.temperature_scale_init = temperature_scale
        self.base_dist = PSDScaledNormal(shape=[features])

        ctx = context_features if context_features > 0 else None
        transforms, self._ar_transforms = [], []
        # MaskedContextLinear's full_context (flows.py:145): MADEWithMaskedContext never passes it (flows.py:268), so every
        # flow the reference builds has the all-ones context mask; the per-position mask is reachable here as a keyword
        full_context = bool(kwargs.pop("full_context", True))
        self.full_context = full_context
        mblocks = (self.n_context_blocks, self.context_block_dim, full_context) if self.use_masked_context else None
        for _ in range(num_layers):
            if not self.use_masked_context:                              # flows.py:459-460
                transforms.append(_Reverse(features))
            layer = _SplineLayer(features, hidden_features, ctx, num_bins, mblocks)
            transforms.append(layer)
            self._ar_transforms.append(layer)
        self.transform = _Composite(transforms)
        self.flow = _FlowAlias(self.transform)
        self.register_buffer("_ar_perm", torch.arange(features, dtype=torch.long))
        self.register_buffer("_ar_inv_perm", torch.arange(features, dtype=torch.long))

        self.precision = os.environ.get("PF_FLOW_PRECISION", "fp32")
        env = os.environ.get("PF_FLOW_HOIST", "")
        self.hoist_context = None if env == "" else env != "0"
        self._workspace = None
        self._packed: Dict[tuple, _Packed] = {}
        self._perm_i32 = None
        self._frozen = False
        self._theta = None           # flatten_parameters(): the one leaf all transform parameters are views of

    # ---- flat parameter storage (opt-in) ---------------------------------------------------------------------------
    def flatten_parameters(self) -> "NSFPosteriorFlow":
        """Re-home every parameter of the transform into ONE flat leaf ``_theta`` (the raw parameter layout of
        include/pf_hip.h: what ``pf_flow_pack`` reads and the backward writes) and turn ``weight`` / ``bias`` of the
        sub-modules into views of it.  What it buys: a differentiable call hands autograd 1 parameter tensor instead of
        18 per layer (180 ``AccumulateGrad`` nodes were ~0.5 ms of host time per backward at LeanNPE's size, more than
        the kernels took), the packing reads the leaf directly (no concatenation per weight update) and an optimiser
        steps one tensor.  What changes: ``parameters()`` / ``named_parameters()`` yield ``_theta`` (+ ``temperature``)
        instead of the per-tensor names -- ``state_dict()`` / ``load_state_dict()`` keep nflows' names (hooks below),
        ``named_parameter_views()`` lists (name, view) pairs and ``named_gradient_views()`` the matching slices of
        ``_theta.grad``.  Plain conditioner only.  Call it before building the optimiser."""
        if self._theta is not None:
            return self
        if self.use_masked_context:
            raise NotImplementedError("flatten_parameters: plain conditioner only (the masked-context variant differentiates "
                                      "tensor ops over its own parameters)")
        params = self._ordered_parameters()
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1).float() for p in params])
        theta = nn.Parameter(flat, requires_grad=any(p.requires_grad for p in params))
        for mod, name in self._param_slots():
            del mod._parameters[name]
        self._theta = theta
        self.register_parameter("_theta", theta)
        self._refresh_views()
        self._packed.clear()
        self.__dict__.pop("_raw_cache", None)
        return self

    def _param_slots(self):
        """(module, attribute) of every transform parameter in raw-layout order"""
        out = []
        for layer in self._ar_transforms:
            net = layer.autoregressive_net
            out += [(net.initial_layer, "weight"), (net.initial_layer, "bias")]
            if hasattr(net, "context_layer"):
                out += [(net.context_layer, "weight"), (net.context_layer, "bias")]
            for b in net.blocks:
                if hasattr(b, "context_layer"):
                    out += [(b.context_layer, "weight"), (b.context_layer, "bias")]
                for lin in b.linear_layers:
                    out += [(lin, "weight"), (lin, "bias")]
            out += [(net.final_layer, "weight"), (net.final_layer, "bias")]
        return out

    def _refresh_views(self):
        """flat mode: (re)create the sub-modules' weight / bias views of ``_theta`` (after construction, .to(), loading)"""
        lay = self._raw_layout()
        off = 0
        slots = self._param_slots()
        per = len(lay["shapes"])
        for i, (mod, name) in enumerate(slots):
            shp, n = lay["shapes"][i % per]
            mod.__dict__[name] = self._theta[off:off + n].view(shp)
            off += n
        self.__dict__.pop("_ordered_cache", None)

    _DEVICE_CACHES = ("_raw_cache", "_ordered_cache", "_raw_mask_cache", "_ctxT", "_plan_cache", "_inc", "_packed", "_workspace",
                      "_perm_i32")

    def __deepcopy__(self, memo):
        """``copy.deepcopy`` (EMA copies, snapshots).  In flat mode the sub-modules hold non-leaf VIEWS of ``_theta`` in their
        ``__dict__`` and torch refuses to deep-copy tensors with a grad_fn: the views are dropped for the copy and rebuilt on
        both sides.  Packed weights, workspaces and the other device-side caches are not copied (the copy packs its own)."""
        import copy
        flat = self._theta is not None
        for mod, name in (self._param_slots() if flat else []):
            mod.__dict__.pop(name, None)
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                if k not in self._DEVICE_CACHES:
                    new.__dict__[k] = copy.deepcopy(v, memo)
        finally:
            if flat:
                self._refresh_views()
        new.__dict__.update(_packed={}, _workspace=None, _perm_i32=None, _frozen=False)
        if flat:
            new._refresh_views()
        return new

    def _slot_names(self):
        """nflows state_dict names of the transform parameters (relative to this module), raw-layout order"""
        names = []
        step = 1 if self.use_masked_context else 2
        for l in range(self.num_layers):
            pre = f"transform._transforms.{step * l + step - 1}.autoregressive_net."
            names += [pre + "initial_layer.weight", pre + "initial_layer.bias"]
            if self.context_features > 0:
                names += [pre + "context_layer.weight", pre + "context_layer.bias"]
            for b in range(2):
                if self.context_features > 0:
                    names += [pre + f"blocks.{b}.context_layer.weight", pre + f"blocks.{b}.context_layer.bias"]
                for k in range(2):
                    names += [pre + f"blocks.{b}.linear_layers.{k}.weight", pre + f"blocks.{b}.linear_layers.{k}.bias"]
            names += [pre + "final_layer.weight", pre + "final_layer.bias"]
        return names

    def named_parameter_views(self):
        """(nflows name, tensor) for every transform parameter: views of ``_theta`` in flat mode, the Parameters otherwise"""
        return list(zip(self._slot_names(), self._ordered_parameters()))

    def named_gradient_views(self):
        """(nflows name, gradient) pairs: slices of ``_theta.grad`` in flat mode, ``p.grad`` otherwise"""
        if self._theta is None:
            return [(n, p.grad) for n, p in self.named_parameter_views()]
        g = self._theta.grad
        out, off = [], 0
        for n, p in self.named_parameter_views():
            out.append((n, None if g is None else g[off:off + p.numel()].view(p.shape)))
            off += p.numel()
        return out

    def _autograd_parameters(self) -> List[torch.Tensor]:
        """what a differentiable call hands to autograd: the one leaf in flat mode, else every Parameter"""
        return [self._theta] if self._theta is not None else self._ordered_parameters()

    def _raw_layout(self) -> dict:
        """offsets (floats) of one layer's tensors inside the raw parameter layout + their shapes"""
        lay = self.__dict__.get("_raw_layout_cache")
        if lay is None:
            H, D, C = self.hidden_features, self.features, self.context_features
            DM = D * (3 * self.num_bins - 1)
            o, shapes = 0, []
            lay = {"W1": [], "b1": [], "W2": [], "b2": [], "Wc": [], "bc": []}

            def take(key, shp, lst=False):
                nonlocal o
                n = 1
                for v in shp:
                    n *= v
                (lay[key].append(o) if lst else lay.__setitem__(key, o))
                shapes.append((tuple(shp), n))
                o += n
            take("W0", (H, D)); take("b0", (H,))
            if C > 0:
                take("Wc", (H, C), True); take("bc", (H,), True)
            for _ in range(2):
                if C > 0:
                    take("Wc", (H, C), True); take("bc", (H,), True)
                take("W1", (H, H), True); take("b1", (H,), True)
                take("W2", (H, H), True); take("b2", (H,), True)
            take("Wf", (DM, H)); take("bf", (DM,))
            lay["P"], lay["shapes"] = o, shapes
            self.__dict__["_raw_layout_cache"] = lay
        return lay

    def _raw_mask(self, dev) -> torch.Tensor:
        """[P] multiplier of one layer's flat gradient: the autoregressive masks on the masked weights, 1 elsewhere"""
        m = self.__dict__.get("_raw_mask_cache")
        if m is None or m.device != dev:
            net = self._ar_transforms[0].autoregressive_net
            parts = []
            for mod, name in self._param_slots()[:len(self._raw_layout()["shapes"])]:
                t = getattr(mod, name)
                if name == "weight" and (isinstance(mod, _MaskedLinear)
                                         or (isinstance(mod, _MaskedContextLinear) and not mod.full_context)):
                    parts.append(mod.mask.detach().reshape(-1).float())
                else:
                    parts.append(torch.ones(t.numel(), dtype=torch.float32, device=t.device))
            m = torch.cat([p.to(dev) for p in parts])
            self.__dict__["_raw_mask_cache"] = m
            del net
        return m

    def packed_ctx_transposed(self, precision: str):
        """all layers' context weights transposed, as pf_dense_nt fragments (pf_flow_pack_ctx_transposed): the weight
        operand of the context gradient; None where the packed form is not built (C % 16, masked-context, L > 16)"""
        if self.use_masked_context and not self.full_context:
            return None                      # (the packing call reads the raw weights; this variant's context mask is not in them)
        desc = self._desc(precision)
        L = _lib.lib()
        nbytes = L.pf_flow_ctx_transposed_bytes(desc)
        if nbytes < 0:
            return None
        dev = self._device()
        raw, key = self._raw_flat(dev)
        st = self.__dict__.setdefault("_ctxT", {})
        ent = st.get(precision)
        if ent is None or ent[0] != key:
            buf = ent[1] if ent is not None and ent[1].device == dev and ent[1].numel() == nbytes else \
                torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(L.pf_flow_pack_ctx_transposed(desc, raw.data_ptr(), buf.data_ptr(),
                                                     torch.cuda.current_stream(dev).cuda_stream), "pf_flow_pack_ctx_transposed")
            st[precision] = ent = (key, buf)
        return ent[1]

    def _raw_flat(self, dev):
        """(flat fp32 copy of the transform parameters in raw-layout order, key of the weights it was made from); in flat
        mode the leaf itself"""
        if self._theta is not None:
            key = (_lib.param_epoch(), dev, self._theta._version, self._theta.data_ptr())
            return self._theta.detach(), key
        params = self._ordered_parameters()
        key = (_lib.param_epoch(), dev, tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
        rk = self.__dict__.get("_raw_cache")          # one flat copy of the parameters serves every stream of a weight update
        if rk is not None and rk[0] == key:
            return rk[1], key
        with torch.no_grad():
            packing = [t for layer in self._ar_transforms for t in layer.autoregressive_net.ordered_parameters(True)]
            raw = torch.cat([p.detach().reshape(-1).float() for p in packing])
        self.__dict__["_raw_cache"] = (key, raw)
        return raw, key

    def freeze_packed(self, frozen: bool = True) -> "NSFPosteriorFlow":
        """Inference: keep the packed weights as they are and skip the per-call scan for
        parameter updates (call again with False, or load new weights, to re-pack)."""
        if frozen:
            self._frozen = False
            self.packed_weights()
            self.packed_weights(inverse=True)
            if self._use_wide(1 << 30):
                self.packed_weights(wide=True)
        self._frozen = frozen
        return self

    def nll_into(self, x: torch.Tensor, context: Optional[torch.Tensor], out: torch.Tensor,
                 log_sigma: Optional[torch.Tensor] = None, sum_count: Optional[torch.Tensor] = None) -> torch.Tensor:
        """compute_psd_aware_nll into a preallocated fp32 ``out[B]`` with no allocation and no
        host-side checks beyond shapes: the call a serving / benchmark loop (or a HIP-graph
        capture) issues.  Inputs must already be contiguous fp32 on the flow's device.
        ``sum_count`` (fp32 [PF_REDUCE_SLOTS, 2] = [16, 2], zeroed by the caller): the kernel also adds (sum of nll, B) to
        it, spread over the slots -- ``sum_count.sum(0)`` is the pair; the 128-byte vector is what a data-parallel rank
        all-reduces."""
        B = x.shape[0]
        self._eval_only("nll_into")
        if x.shape[1] != self.features or out.shape[0] != B or not x.is_contiguous():
            raise ValueError("nll_into: bad shapes / non-contiguous input")
        if self.context_features > 0 and (context is None or context.shape != (B, self.context_features)
                                          or not context.is_contiguous()):
            raise ValueError("nll_into: bad context")
        dev = x.device
        perm, _ = self._perms(dev)
        wide = self._use_wide(B)
        desc = self._desc(wide=wide)
        ws, ws_bytes = self._ws(desc, B, dev)
        if sum_count is not None:
            if sum_count.dtype != torch.float32 or sum_count.numel() != 2 * _lib.PF_REDUCE_SLOTS or not sum_count.is_contiguous():
                raise ValueError(f"nll_into: sum_count must be a contiguous fp32 tensor of [{_lib.PF_REDUCE_SLOTS}, 2] elements")
            _lib.check(_lib.lib().pf_flow_forward_reduce(
                desc, self.packed_weights(wide=wide).data_ptr(), x.data_ptr(), _dev_ptr(context),
                _dev_ptr(perm), _dev_ptr(log_sigma), B, out.data_ptr(), sum_count.data_ptr(), None,
                _dev_ptr(ws), ws_bytes, torch.cuda.current_stream(dev).cuda_stream), "pf_flow_forward_reduce")
            return out
        _lib.check(_lib.lib().pf_flow_forward(
            desc, self.packed_weights(wide=wide).data_ptr(), x.data_ptr(), _dev_ptr(context),
            _dev_ptr(perm), _dev_ptr(log_sigma), B, None, None, out.data_ptr(),
            _dev_ptr(ws), ws_bytes, torch.cuda.current_stream(dev).cuda_stream), "pf_flow_forward")
        return out

    def bind_nll(self, x: torch.Tensor, context: Optional[torch.Tensor], out: torch.Tensor,
                 sum_count: Optional[List[torch.Tensor]] = None, stream: Optional[torch.cuda.Stream] = None):
        """A launcher with every argument of ``nll_into`` resolved once (descriptor, packed weights,
        workspace, pointers): ``launch(i)`` is one C call, a few microseconds of host time, for loops whose
        step is a single ~100 us kernel.  ``sum_count``: rotating fp32 [16, 2] accumulators (``nll_into``); launch i adds
        (sum nll, B) to ``sum_count[i % n]`` and zeroes ``sum_count[(i + 1) % n]`` for the next launch
        (``n >= 2``; the caller zeroes ``sum_count[0]`` before launch 0).  The tensors must stay alive and
        the weights frozen (``freeze_packed``) while the launcher is used."""
        self.nll_into(x, context, out)                          # validates shapes once
        dev = x.device
        perm, _ = self._perms(dev)
        wide = self._use_wide(x.shape[0])
        desc = self._desc(wide=wide)
        ws, ws_bytes = self._ws(desc, x.shape[0], dev)
        packed = self.packed_weights(wide=wide)
        fn = _lib.lib().pf_flow_forward_reduce if sum_count else _lib.lib().pf_flow_forward
        st = (stream or torch.cuda.current_stream(dev)).cuda_stream
        keep = (x, context, out, perm, ws, packed, desc, sum_count)
        base = (desc, packed.data_ptr(), x.data_ptr(), _dev_ptr(context), _dev_ptr(perm), None, x.shape[0])
        if sum_count:
            n = len(sum_count)
            ptrs = [t.data_ptr() for t in sum_count]
            tail = (_dev_ptr(ws), ws_bytes, st)

            def launch(i: int, _keep=keep):
                rc = fn(*base, out.data_ptr(), ptrs[i % n], ptrs[(i + 1) % n], *tail)
                if rc:
                    _lib.check(rc, "pf_flow_forward_reduce")
        else:
            args = (*base, None, None, out.data_ptr(), _dev_ptr(ws), ws_bytes, st)

            def launch(i: int = 0, _keep=keep):
                rc = fn(*args)
                if rc:
                    _lib.check(rc, "pf_flow_forward")
        return launch

    # ---- plumbing -------------------------------------------------------------
    # ---- the PF_FLAG_WIDE layout: 32-unit x 16-k fragments shared by the mid-batch kernel (64 rows per workgroup, two waves
    # per SIMD: up to 16 384 rows) and the large-batch kernel (128 rows per workgroup, weights through an LDS ring: above);
    # pf_flow_forward picks between the two by rounds x round time ----
    wide_min_batch: int = 8193      # rows from which pf_flow_forward is given the PF_FLAG_WIDE layout (measured: the 16-row
                                    # kernel takes 143 us up to 8192 rows and 232 / 284 us at 12 288 / 16 384; a round of the
                                    # mid-batch kernel 165 / 182 us there (149 at 8 192); the large-batch kernel 325 us up to
                                    # 32 768 rows; which of the two runs: pf_flow_fwd.hip `use_mid`)

    def _use_wide(self, batch: int) -> bool:
        env = os.environ.get("PF_FLOW_WIDE", "")
        if env == "0" or self.precision != "bf16" or self.use_masked_context or self.hoist_context:
            return False
        if self.hidden_features != 256 or self.num_bins != 16 or (self.features, self.context_features) not in ((15, 288), (11, 288)):
            return False
        return env == "1" or batch >= self.wide_min_batch

    def _desc(self, precision: Optional[str] = None, inverse: bool = False, wide: bool = False,
              bwd: bool = False, generic: bool = False) -> _lib.PfFlowDesc:
        prec = _lib.PRECISIONS[precision or self.precision]
        if generic:  # the generic kernel's layout for this shape (fp32-mode conditioner re-evaluation reads it)
            return _lib.PfFlowDesc(self.features, self.context_features, self.hidden_features, self.num_bins,
                                   self.num_layers, 2, float(self._tail_bound), _MIN_BIN, _MIN_BIN, _MIN_BIN,
                                   prec, _lib.PF_FLAG_GENERIC | (_lib.PF_FLAG_MASKED_CONTEXT if self.use_masked_context else 0))
        if bwd:     # packing only: the backward chain's transposed bf16 fragments (PF_FLAG_BWD)
            return _lib.PfFlowDesc(self.features, self.context_features, self.hidden_features, self.num_bins,
                                   self.num_layers, 2, float(self._tail_bound), _MIN_BIN, _MIN_BIN, _MIN_BIN,
                                   _lib.PF_PREC_BF16, _lib.PF_FLAG_BWD)
        if wide:
            return _lib.PfFlowDesc(self.features, self.context_features, self.hidden_features, self.num_bins,
                                   self.num_layers, 2, float(self._tail_bound), _MIN_BIN, _MIN_BIN, _MIN_BIN,
                                   _lib.PF_PREC_BF16, _lib.PF_FLAG_WIDE)
        hoist = inverse if self.hoist_context is None else self.hoist_context
        flags = _lib.PF_FLAG_HOIST_CTX if (hoist and self.context_features > 0) else 0
        if self.use_masked_context:
            flags = _lib.PF_FLAG_HOIST_CTX | _lib.PF_FLAG_MASKED_CONTEXT
        make = lambda fl: _lib.PfFlowDesc(self.features, self.context_features, self.hidden_features,
                                          self.num_bins, self.num_layers, 2, float(self._tail_bound),
                                          _MIN_BIN, _MIN_BIN, _MIN_BIN, prec, fl)
        desc = make(flags)
        if self.hoist_context is None and not flags and self.context_features > 0:
            # auto mode: the in-layer context kernels cover C <= 288 (576 in bf16 at H = 256); wider contexts
            # go through the hoisted projection kernel, which has no such limit
            key = ("inlayer_ok", prec)
            ok = self.__dict__.setdefault("_plan_cache", {}).get(key)
            if ok is None:
                ok = _lib.lib().pf_flow_pack_map_len(desc) >= 0
                self._plan_cache[key] = ok
            if not ok:
                desc = make(_lib.PF_FLAG_HOIST_CTX)
        return desc

    def _ws(self, desc, ctx_rows: int, dev):
        """Caller-owned scratch for the hoisted context projections (grown on demand)."""
        need = _lib.lib().pf_flow_workspace_bytes(desc, ctx_rows)
        if need <= 0:
            return None, 0
        if self._workspace is None or self._workspace.device != dev or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._workspace, need

    def _ordered_parameters(self) -> List[torch.Tensor]:
        out = self.__dict__.get("_ordered_cache")        # the module tree is fixed after __init__ (a list of the same
        if out is None:                                  # Parameter objects .to() / load_state_dict keep)
            out = []
            for layer in self._ar_transforms:
                out += layer.autoregressive_net.ordered_parameters()
            self.__dict__["_ordered_cache"] = out
        return out

    def _device(self) -> torch.device:
        dev = self.temperature.device
        if dev.type != "cuda":
            raise _lib.PfError(
                "NSFPosteriorFlow is on %s: posteriflow_amd evaluates flows on the MI355X only "
                "(no CPU fallback); move the module with .to('cuda')" % dev)
        return dev

    def packed_weights(self, precision: Optional[str] = None, inverse: bool = False, wide: bool = False,
                       bwd: bool = False, generic: bool = False) -> torch.Tensor:
        """Packed (masked, fragment-ordered) weights, rebuilt when a parameter changed."""
        desc = self._desc(precision, inverse, wide, bwd, generic)
        ck = (desc.precision, desc.reserved)
        if self._frozen and ck in self._packed and self._packed[ck].buf is not None:
            return self._packed[ck].buf
        dev = self._device()
        L = _lib.lib()
        raw, key = self._raw_flat(dev)
        pk = self._packed.setdefault((desc.precision, desc.reserved), _Packed())
        if pk.key == key:
            return pk.buf
        if pk.map is None or pk.map.device != dev:
            n = L.pf_flow_pack_map_len(desc)
            if n < 0:
                _lib.check(_lib.PF_ERR_UNSUPPORTED, "pf_flow_pack_map_len")
            host = torch.empty(n, dtype=torch.int32)
            _lib.check(L.pf_flow_build_pack_map(desc, host.data_ptr()), "pf_flow_build_pack_map")
            pk.map = host.to(dev)
            pk.buf = torch.empty(L.pf_flow_packed_bytes(desc), dtype=torch.uint8, device=dev)
        assert raw.numel() == L.pf_flow_raw_param_count(desc)
        _lib.check(L.pf_flow_pack(desc, raw.data_ptr(), pk.map.data_ptr(), pk.buf.data_ptr(),
                                  torch.cuda.current_stream(dev).cuda_stream), "pf_flow_pack")
        pk.key = key
        return pk.buf

    def _perms(self, dev) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        """(ar_perm, ar_inv_perm) as int32 device tensors, or (None, None) for identity.
        Resolved once (one D2H copy) and cached until the order / device / weights change."""
        if self._perm_i32 is None or self._perm_i32[0] != dev:
            perm = self._ar_perm.detach().cpu()
            if torch.equal(perm, torch.arange(self.features)):
                self._perm_i32 = (dev, None, None)
            else:
                self._perm_i32 = (dev, self._ar_perm.to(dev, torch.int32).contiguous(),
                                  self._ar_inv_perm.to(dev, torch.int32).contiguous())
        return self._perm_i32[1], self._perm_i32[2]

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_ordered_cache", None)     # .to() / .cuda() may swap Parameter objects
        out = super()._apply(fn, *args, **kwargs)
        if self._theta is not None:
            self._theta = self._parameters["_theta"]
            self._refresh_views()
        self.__dict__.pop("_raw_mask_cache", None)
        return out

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self._theta is not None:        # flat mode: nflows' names (both registration paths, flows.py:529, 532), not "_theta"
            destination.pop(prefix + "_theta", None)
            for name, view in self.named_parameter_views():
                t = view if keep_vars else view.detach()
                destination[prefix + name] = t
                destination[prefix + "flow._" + name] = t

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        if self._theta is not None:        # flat mode: gather nflows-named tensors into the leaf
            with torch.no_grad():
                for name, view in self.named_parameter_views():
                    src = state_dict.pop(prefix + name, None)
                    alias = state_dict.pop(prefix + "flow._" + name, None)
                    src = alias if src is None else src
                    if src is None:
                        if strict:
                            missing_keys.append(prefix + name)
                        continue
                    if tuple(src.shape) != tuple(view.shape):
                        error_msgs.append(f"size mismatch for {prefix + name}: {tuple(src.shape)} vs {tuple(view.shape)}")
                        continue
                    view.copy_(src)
            state_dict = dict(state_dict)
            state_dict[prefix + "_theta"] = self._theta.detach()
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        self.__dict__.pop("_ordered_cache", None)
        self._perm_i32 = None          # _ar_perm may have come from the checkpoint
        self._frozen = False

    def _check_inputs(self, x, context, what):
        dev = self._device()
        if x.dim() != 2 or x.shape[1] != self.features:
            raise ValueError(f"{what}: expected [batch, {self.features}], got {tuple(x.shape)}")
        if x.device != dev:
            raise _lib.PfError(f"{what}: input on {x.device}, flow on {dev}")
        x = x.contiguous().float()
        if self.context_features > 0 and context is not None:
            if context.dim() != 2 or context.shape[1] != self.context_features:
                raise ValueError(f"{what}: context must be [batch, {self.context_features}], "
                                 f"got {tuple(context.shape)}")
            if context.device != dev:
                raise _lib.PfError(f"{what}: context on {context.device}, flow on {dev}")
            context = self._permute_context_blocks(context).contiguous().float()
        elif self.context_features > 0:
            raise ValueError(f"{what}: this flow was built with context_features="
                             f"{self.context_features}; a context is required")
        else:
            context = None
        return dev, x, context

    def _needs_grad(self, *tensors) -> bool:
        need = torch.is_grad_enabled() and (
            any(t is not None and t.requires_grad for t in tensors)
            or any(p.requires_grad for p in self._autograd_parameters()))
        if need and self._generic_shape() and not self._generic_trainable():
            raise NotImplementedError(
                f"NSFPosteriorFlow(H={self.hidden_features}, D={self.features}, K={self.num_bins}): the generic-shape kernel "
                "evaluates densities and samples for this shape; its backward (fp32 data-gradient chain) needs "
                "H in {64, 128, 192, 256, 384, 512}, D <= 16, K <= 32, a plain conditioner and no conditioner dropout -- call "
                "under torch.no_grad() or freeze the parameters")
        return need

    def _generic_trainable(self) -> bool:
        """generic shapes the fp32 backward chain covers (csrc/pf_flow_bwd_chain.hip: H = 384 / 512 and K <= 32 in fp32)"""
        return (self.hidden_features in (64, 128, 192, 256, 384, 512) and self.features <= 16 and self.num_bins <= 32
                and not (self.dropout and self.dropout > 0.0))

    # ---- conditioner dropout (train mode) ---------------------------------------------
    # nflows drops relu(W0 relu(h) + b0) inside every residual block while the module is in train mode (upstream
    # MaskedResidualBlock.forward, built at flows.py:510-524 with dropout_probability=dropout; the reference's own
    # masked-context block does the same, flows.py:225-234).  The training forward kernel (flow_train_kernel) applies
    # it from a counter hash of a per-call seed; the backward regenerates the factors (pf_flow_dropout_mask).
    def _drop_active(self) -> bool:
        return bool(self.training and self.dropout and self.dropout > 0.0)

    def _eval_only(self, what: str) -> None:
        """the serving entry points and the inverse have no dropout: in train() mode of a flow with dropout > 0 nflows
        would apply a fresh mask in every MADE call (every autoregressive pass of the inverse) -- fail loudly instead
        of silently evaluating a different model"""
        if self._drop_active():
            raise RuntimeError(f"NSFPosteriorFlow.{what}: the flow is in train() mode with dropout={self.dropout:g}; "
                               "call .eval() first (dropout is applied by the differentiable forward only)")

    @staticmethod
    def _draw_dropout_seed() -> int:
        """a fresh 62-bit seed from torch's CPU generator (follows torch.manual_seed; no device sync)"""
        return int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64).item())

    def forward_kernel_name(self, batch: int) -> str:
        """Name of the kernel pf_flow_forward dispatches for this flow at `batch` rows (as rocprofv3 prints it)."""
        name = _lib.lib().pf_flow_forward_kernel_name(self._desc(wide=self._use_wide(int(batch))), int(batch))
        return name.decode() if name else "?"

    def _forward_call(self, x, context, log_sigma, want_z=True, guard=True, layer_inputs=None, dropout_seed=None):
        if guard:
            dev, x, context = self._check_inputs(x, context, "NSFPosteriorFlow.forward")
        else:                       # already validated (and context blocks permuted) by the caller
            dev = x.device
        B = x.shape[0]
        if context is not None and context.shape[0] != B:
            raise ValueError(f"batch mismatch: x {B} vs context {context.shape[0]}")
        z = torch.empty_like(x) if want_z else None
        logdet = torch.empty(B, dtype=torch.float32, device=dev)
        nll = torch.empty(B, dtype=torch.float32, device=dev)
        perm, _ = self._perms(dev)
        drop = self._drop_active()
        wide = self._use_wide(B) and not drop
        packed = self.packed_weights(wide=wide)
        desc = self._desc(wide=wide)
        ws, ws_bytes = self._ws(desc, B, dev)
        # layer_inputs: fp32 [L, B, D] that receives every conditioner's input (training forward)
        head = (desc, packed.data_ptr(), x.data_ptr(), _dev_ptr(context), _dev_ptr(perm), _dev_ptr(log_sigma), B,
                _dev_ptr(z), logdet.data_ptr(), nll.data_ptr(), _dev_ptr(layer_inputs))
        tail = (_dev_ptr(ws), ws_bytes, torch.cuda.current_stream(dev).cuda_stream)
        if drop:                    # train mode: a no-grad call draws its own seed, like nn.Dropout would
            seed = self._draw_dropout_seed() if dropout_seed is None else int(dropout_seed)
            _lib.check(_lib.lib().pf_flow_forward_train_dropout(*head, float(self.dropout), seed, *tail),
                       "pf_flow_forward_train_dropout")
        else:
            _lib.check(_lib.lib().pf_flow_forward_train(*head, *tail), "pf_flow_forward")
        return z, logdet, nll

    def _nll(self, x, context, log_sigma):
        """nll[B]; differentiable: the HIP backward of _flow_autograd.py (re-evaluation, chain, transposed GEMMs)."""
        if self._needs_grad(x, context, log_sigma):
            from ._flow_autograd import FlowNLL
            dev, x, context = self._check_inputs(x, context, "NSFPosteriorFlow.compute_psd_aware_nll")
            return FlowNLL.apply(self, x, context, log_sigma, *self._autograd_parameters())[0]
        return self._forward_call(x, context, log_sigma, want_z=False)[2]

    # ---- reference API ------------------------------------------------------------
    def set_autoregressive_order(self, order: List[int]) -> None:            # flows.py:550-588
        if sorted(order) != list(range(self.features)):
            raise ValueError(f"order must be a permutation of range({self.features}), got {order}")
        perm = torch.tensor(order, dtype=torch.long, device=self._ar_perm.device)
        self._ar_perm = perm
        self._ar_inv_perm = torch.argsort(perm)
        self._perm_i32 = None

    def _permute_context_blocks(self, context: torch.Tensor) -> torch.Tensor:  # flows.py:590-608
        """masked-context mode: context block i follows whatever parameter sits at autoregressive
        position i (identity for the default order and for the plain conditioner)."""
        if not self.use_masked_context:
            return context
        b = context.shape[0]
        blocks = context.view(b, self.n_context_blocks, self.context_block_dim)[:, self._ar_perm, :]
        return blocks.reshape(b, self.n_context_blocks * self.context_block_dim)

    def forward(self, x: torch.Tensor, context: Optional[torch.Tensor] = None):
        """x -> (z, log|det dz/dx|)   (flows.py:610-618)."""
        if self._needs_grad(x, context):
            from ._flow_autograd import FlowForward
            dev, x, context = self._check_inputs(x, context, "NSFPosteriorFlow.forward")
            return FlowForward.apply(self, x, context, *self._autograd_parameters())
        z, logdet, _ = self._forward_call(x, context, None)
        return z, logdet

    def compute_psd_aware_nll(self, x, context, log_sigma_psd):
        """-(log N(z; 0, Sigma_psd) + log|det|)   (flows.py:727-779); one kernel."""
        ls = None
        if log_sigma_psd is not None:
            if log_sigma_psd.shape != x.shape:
                raise ValueError(f"Shape mismatch: z {x.shape} vs log_sigma_psd {log_sigma_psd.shape}")
            ls = log_sigma_psd.contiguous().float()
        return self._nll(x, context, ls)

    def log_prob(self, x, context=None, temperature: Optional[float] = None):
        """Negative log-density with the temperature change of variables
        (flows.py:657-695, to its documented math with the N(0, I) base)."""
        if context is not None:
            # the reference tests `isfinite(context).all()` on the host first (flows.py:664-669); the
            # replacement is the identity on finite input, so it is applied unconditionally: no device->host
            # sync per call (the warning the reference logs is not reproduced)
            context = torch.nan_to_num(context, nan=0.0, posinf=1e-3, neginf=-1e-3)
        t = torch.clamp(self.temperature, 0.5, 3.0) if temperature is None \
            else torch.as_tensor(float(temperature), device=x.device)
        nll = self._nll(x / t, context, None)
        out = -nll - self.features * torch.log(t)
        return -torch.nan_to_num(out, nan=1000.0)

    def compute_nll_loss(self, params_norm, context):                        # flows.py:900-908
        return self.log_prob(params_norm, context).mean()

    def compute_bounds_penalty(self, params_norm, bounds=(-FLOW_NORM_BOUND, FLOW_NORM_BOUND)):
        lo, hi = bounds                                                      # flows.py:910-920
        return torch.relu(lo - params_norm).mean() + torch.relu(params_norm - hi).mean()

    # ---- inverse / sampling ------------------------------------------------------------------
    # ---- incremental inverse: one masked conditioner evaluation per layer instead of D dense ones ----
    incremental_inverse: Optional[bool] = None      # None: on for the plain conditioner (both precisions); False: D-pass kernel

    def _generic_shape(self) -> bool:
        """a shape outside the scheduled kernels' set (e.g. the 12 x 384 x 24 head of experiments/frozen_context_heads.py:
        159-163): evaluated by the generic kernel (csrc/pf_flow_generic.hip) -- forward and the D-pass inverse, no gradients"""
        return not (self.hidden_features in (64, 128, 192, 256) and 1 <= self.features <= self.hidden_features // 16
                    and 2 <= self.num_bins <= 16)

    def _use_incremental(self) -> bool:
        if self.incremental_inverse is False or self.use_masked_context or self._generic_shape():
            return False
        # features == 1: every hidden unit has degree 0 (one pass, nothing incremental about it): D-pass kernel
        return (self.hidden_features % 32 == 0 and 2 <= self.features <= min(16, self.hidden_features // 16))

    def _inc_state(self, dev):
        """Weights of every layer in the layout of pf_flow_inverse_inc (hidden units sorted by degree, masks
        applied, MFMA A-fragments in the current precision) + the stacked context-projection weights; rebuilt when a
        parameter changed."""
        params = self._ordered_parameters()
        f32 = self.precision != "bf16"
        key = (_lib.param_epoch(), dev, f32, tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
        st = self.__dict__.setdefault("_inc", {}).setdefault(self.precision, {})
        if st.get("key") == key:
            return st
        L_, D, H, K = _lib.lib(), self.features, self.hidden_features, self.num_bins
        desc = self._desc(self.precision, inverse=True)
        deg = _hidden_degrees(H, D).to(dev)
        perm = torch.argsort(deg, stable=True)
        deg_sorted = deg[perm]
        u1 = (_lib.C.c_int32 * (D + 1))(*[int((deg_sorted <= i).sum()) for i in range(D + 1)])
        layer_bytes = L_.pf_flow_inc_layer_bytes(desc)
        if layer_bytes <= 0:
            _lib.check(_lib.PF_ERR_UNSUPPORTED, "pf_flow_inc_layer_bytes")
        nl = self.num_layers
        buf = torch.empty(nl * layer_bytes, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        nt, hk = H // 16, H // 32
        wc, bc = [], []
        with torch.no_grad():
            for l, layer in enumerate(self._ar_transforms):
                net = layer.autoregressive_net
                off = l * layer_bytes

                def pack(mat, off):
                    mat = mat.float().contiguous()
                    n, k = mat.shape
                    if f32:   # fragment (tile, q), lane (r, kq): W[16 tile + r][16 q + 4 kq .. + 3], lane = 16 kq + r
                        frag = mat.view(n // 16, 16, k // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous().view(-1)
                        buf[off:off + n * k * 4].copy_(frag.view(torch.uint8))
                        return off + n * k * 4
                    _lib.check(L_.pf_pack_bf16_frags(mat.data_ptr(), n, k, buf.data_ptr() + off, stream),
                               "pf_pack_bf16_frags")
                    return off + n * k * 2

                w0 = (net.initial_layer.weight * net.initial_layer.mask)[perm]
                a0 = torch.zeros(H, 16 if f32 else 32, device=dev)
                a0[:, :D] = w0
                if not f32:                        # the input enters the bf16 kernel as a hi | lo pair
                    a0[:, 16:16 + D] = w0
                off = pack(a0, off)
                biases = [net.initial_layer.bias[perm]]
                for blk in net.blocks:
                    for lin in blk.linear_layers:
                        off = pack((lin.weight * lin.mask)[perm][:, perm], off)
                        biases.append(lin.bias[perm])
                wf = (net.final_layer.weight * net.final_layer.mask)[:, perm].reshape(D, 3 * K - 1, H)
                wf48 = torch.zeros(D, 48, H, device=dev)
                wf48[:, 0:K] = wf[:, 0:K]
                wf48[:, 16:16 + K] = wf[:, K:2 * K]
                wf48[:, 32:32 + K - 1] = wf[:, 2 * K:]
                off = pack(wf48.reshape(D * 48, H), off)
                bf = net.final_layer.bias.reshape(D, 3 * K - 1)
                bf48 = torch.zeros(D, 48, device=dev)
                bf48[:, 0:K] = bf[:, 0:K]
                bf48[:, 16:16 + K] = bf[:, K:2 * K]
                bf48[:, 32:32 + K - 1] = bf[:, 2 * K:]
                vec = torch.cat([b.float() for b in biases] + [bf48.reshape(-1)]).contiguous()
                assert off + vec.numel() * 4 == (l + 1) * layer_bytes
                buf[off:off + vec.numel() * 4].copy_(vec.view(torch.uint8))
                if self.context_features > 0:
                    wc.append(torch.stack([net.context_layer.weight[perm]] + [b.context_layer.weight[perm] for b in net.blocks]))
                    bc.append(torch.stack([net.context_layer.bias[perm]] + [b.context_layer.bias[perm] for b in net.blocks]))
            wfrags = bcat = None
            if wc:       # every layer's three context matrices as ONE packed GEMM operand [L * 3 * H][C] (sorted-unit rows)
                prec = _lib.PRECISIONS[self.precision]
                C_ = self.context_features
                kpad = -(-C_ // (16 if f32 else 32)) * (16 if f32 else 32)
                wcat = torch.zeros(len(wc) * 3 * H, kpad, dtype=torch.float32, device=dev)
                wcat[:, :C_] = torch.stack(wc).reshape(-1, C_).float()
                nbytes = _lib.lib().pf_dense_frag_bytes(prec, wcat.shape[0], kpad)
                wfrags = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                _lib.check(_lib.lib().pf_dense_pack_matrix(prec, wcat.data_ptr(), 0, kpad, wcat.shape[0], kpad, wfrags.data_ptr(),
                                                           torch.cuda.current_stream(dev).cuda_stream), "pf_dense_pack_matrix")
                bcat = torch.stack(bc).reshape(-1).float().contiguous()
                st["_wcat_keep"] = wcat          # (the pack kernel is asynchronous)
            st.update(key=key, buf=buf, u1=u1, wfrags=wfrags, bcat=bcat)
        return st

    def _inverse_call(self, z, context, ctx_rows):
        self._eval_only("inverse / sample")
        dev = self._device()
        B = z.shape[0]
        x = torch.empty_like(z)
        logdet = torch.empty(B, dtype=torch.float32, device=dev)
        flags = torch.zeros(B, dtype=torch.int32, device=dev)
        _, inv_perm = self._perms(dev)
        if self._use_incremental():
            st = self._inc_state(dev)
            proj = None
            if st["wfrags"] is not None:
                # [ctx_rows, L * 3 * H]: one GEMM for every layer's three context projections (pf_flow_ctx_project_rows: the
                # hoisted-projection kernel with a row-major epilogue), with the operand rounding of the bf16 forward kernel
                # (context and weights to bf16, fp32 accumulation) so that forward(inverse(z)) sees the same conditioner
                cx = context.float().contiguous()
                proj = torch.empty(cx.shape[0], st["bcat"].numel(), dtype=torch.float32, device=dev)
                _lib.check(_lib.lib().pf_flow_ctx_project_rows(
                    _lib.PRECISIONS[self.precision], st["wfrags"].data_ptr(), st["bcat"].data_ptr(), cx.data_ptr(), cx.shape[0],
                    self.context_features, st["bcat"].numel(), proj.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                    "pf_flow_ctx_project_rows")
            rc = _lib.lib().pf_flow_inverse_inc(
                self._desc(self.precision, inverse=True), st["u1"], st["buf"].data_ptr(), _dev_ptr(proj), ctx_rows,
                z.data_ptr(), _dev_ptr(inv_perm), B, x.data_ptr(), logdet.data_ptr(), flags.data_ptr(),
                torch.cuda.current_stream(dev).cuda_stream)
            if rc != _lib.PF_ERR_UNSUPPORTED:
                _lib.check(rc, "pf_flow_inverse_inc")
                return x, logdet, flags
            # more than 8 sixteen-unit tiles of one degree (H / (D - 1) > 112: few features, where the D-pass
            # kernel needs only 2-3 passes anyway): the D-pass kernel below handles every shape
            self.incremental_inverse = False
        desc = self._desc(inverse=True)
        ws, ws_bytes = self._ws(desc, ctx_rows, dev)
        _lib.check(_lib.lib().pf_flow_inverse(
            desc, self.packed_weights(inverse=True).data_ptr(), z.data_ptr(), _dev_ptr(context),
            ctx_rows, _dev_ptr(inv_perm), B, x.data_ptr(), logdet.data_ptr(), flags.data_ptr(),
            _dev_ptr(ws), ws_bytes, torch.cuda.current_stream(dev).cuda_stream), "pf_flow_inverse")
        return x, logdet, flags

    def inverse(self, z: torch.Tensor, context: Optional[torch.Tensor] = None,
                n_overlaps: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """z -> (x, log|det dx/dz|), deterministic (flows.py:620-655).

        Same post-processing as the reference: non-finite context entries are replaced
        (nan -> 0, +-inf -> +-1e-3), the result is un-permuted, nan_to_num'ed and clamped to
        +-FLOW_NORM_BOUND.  Where the spline's quadratic has a negative discriminant the
        reference raises inside nflows and falls back to x = z, log_det = 0 for the WHOLE
        batch (flows.py:638-642); here the fallback is applied to the affected rows only
        (per-row flags from the kernel, no host synchronisation).

        ``context`` may have fewer rows than ``z`` if the counts divide: sample i then uses
        context row i // (len(z) // len(context)), the expand().reshape() pattern of
        lean_npe.py:328 / pipeline.py:171 without materialising the copies; a stride-0
        (``expand``-ed) context is recognised and treated as a single row.
        """
        dev = self._device()
        if z.dim() != 2 or z.shape[1] != self.features:
            raise ValueError(f"NSFPosteriorFlow.inverse: expected [batch, {self.features}], got {tuple(z.shape)}")
        if z.device != dev:
            raise _lib.PfError(f"NSFPosteriorFlow.inverse: input on {z.device}, flow on {dev}")
        if torch.is_grad_enabled() and (z.requires_grad or (context is not None and context.requires_grad)):
            raise NotImplementedError("inverse() has no backward (every reference caller runs it under no_grad)")
        z = z.contiguous().float()
        B = z.shape[0]
        if B == 0:                                          # nothing to invert (the reference's tensor ops return empties too)
            return z.clone(), torch.empty(0, dtype=torch.float32, device=dev)
        ctx_rows = B
        if self.context_features > 0:
            if context is None:
                raise ValueError("NSFPosteriorFlow.inverse: this flow requires a context")
            if context.dim() != 2 or context.shape[1] != self.context_features:
                raise ValueError(f"NSFPosteriorFlow.inverse: context must be [rows, {self.context_features}], "
                                 f"got {tuple(context.shape)}")
            if context.device != dev:
                raise _lib.PfError(f"NSFPosteriorFlow.inverse: context on {context.device}, flow on {dev}")
            if context.shape[0] == B and B > 1 and context.stride(0) == 0:
                context = context[:1]                      # expand()-ed single row
            ctx_rows = context.shape[0]
            if ctx_rows < 1 or (B and B % ctx_rows != 0):
                raise ValueError(f"NSFPosteriorFlow.inverse: {ctx_rows} context rows do not divide batch {B}")
            context = torch.nan_to_num(context.float(), nan=0.0, posinf=1e-3, neginf=-1e-3)
            context = self._permute_context_blocks(context).contiguous()
        else:
            context = None
        x, logdet, flags = self._inverse_call(z, context, ctx_rows)
        bad = flags.bool()
        x = torch.where(bad[:, None], z, x)
        logdet = torch.where(bad, torch.zeros_like(logdet), logdet)
        x = torch.nan_to_num(x, nan=0.0, posinf=1.0, neginf=-1.0)
        return torch.clamp(x, -FLOW_NORM_BOUND, FLOW_NORM_BOUND), logdet

    def _temperature(self, temperature, dev):
        if temperature is None:
            return torch.clamp(self.temperature.detach(), 0.5, 3.0)
        return torch.tensor(float(temperature), dtype=torch.float32, device=dev)

    @torch.no_grad()
    def sample(self, num_samples: int, context: Optional[torch.Tensor] = None,
               temperature: Optional[float] = None) -> torch.Tensor:
        """Draws from p(x | context): z ~ N(0, T^2 I) through the inverse (flows.py:697-725).
        Returns [batch, num_samples, D] with a context, [num_samples, D] without.  (The
        reference's version feeds a 3-D z into the 2-D transform, flows.py:714 vs :101-109;
        this follows its documented intent.)"""
        dev = self._device()
        t = self._temperature(temperature, dev)
        if context is not None:
            b = context.shape[0]
            z = torch.randn(b * num_samples, self.features, device=dev) * t
            x, _ = self.inverse(z, context)              # context rows are grouped, not copied
            x = x.reshape(b, num_samples, self.features)
        else:
            z = torch.randn(num_samples, self.features, device=dev) * t
            x, _ = self.inverse(z, None)
        return torch.clamp(x, -FLOW_NORM_BOUND, FLOW_NORM_BOUND)

    @torch.no_grad()
    def sample_psd_aware(self, num_samples: int, context: torch.Tensor,
                         log_sigma_psd: torch.Tensor) -> torch.Tensor:
        """[batch, num_samples, D]: unit-normal base through the inverse, clamp, then the
        factored scale exp(log_sigma_psd) applied outside the flow (flows.py:781-842)."""
        b = context.shape[0]
        z = self.base_dist.sample(num_samples, log_sigma_psd).to(context.device)
        x, _ = self.inverse(z.reshape(b * num_samples, self.features), context)
        x = torch.clamp(x, -FLOW_NORM_BOUND, FLOW_NORM_BOUND).reshape(b, num_samples, self.features)
        return x * torch.exp(log_sigma_psd).unsqueeze(1)

    def sample_with_uncertainty(self, num_samples: int, context: torch.Tensor,
                                n_overlaps: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        samples = self.sample(num_samples, context)                          # flows.py:844-878
        if context.shape[0] == 1 and samples.dim() == 3:
            samples = samples.squeeze(0)
        dim = 1 if samples.dim() == 3 else 0
        return {"samples": samples, "mean": samples.mean(dim=dim), "std": samples.std(dim=dim),
                "n_overlaps": n_overlaps[0] if n_overlaps is not None else None}

    def extract_signals(self, strain_batch: torch.Tensor, context: torch.Tensor,
                        n_samples: int = 50, return_all_samples: bool = False) -> Dict:
        samples = self.sample(n_samples, context)                            # flows.py:880-898
        dim = 0 if samples.dim() == 2 else 1
        out = {"mean": samples.mean(dim=dim), "std": samples.std(dim=dim), "cov": None}
        if return_all_samples:
            out["samples_all"] = samples
        return out

    def compute_endpoint_loss(self, params_norm: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():                                                # flows.py:922-939
            zmin = torch.full_like(params_norm, -FLOW_NORM_BOUND)
            xmin, _ = self.inverse(zmin, context)
            xmax, _ = self.inverse(-zmin, context)
        return (torch.relu(xmin - params_norm) + torch.relu(params_norm - xmax)).mean()


def create_flow_model(flow_type: str, features: int, context_features: int = 0,
                      max_overlaps: int = 6, config: Optional[Any] = None, **kwargs) -> nn.Module:
    """Factory with the reference's signature (flows.py:942-1022).  ``config`` may be a
    mapping (or an object with attributes) holding a ``flow_config`` block with
    num_layers / hidden_features / num_bins / dropout / tail_bound / per_param_tail_bounds."""
    if config is not None:
        if isinstance(config, (str, os.PathLike)):
            import yaml
            with open(config) as fh:
                config = yaml.safe_load(fh)
        fc = config.get("flow_config", {}) if isinstance(config, dict) else getattr(config, "flow_config", {})
        fc = fc or {}
        for name, default, cast in (("num_layers", 12, int), ("hidden_features", 256, int),
                                    ("num_bins", 16, int), ("dropout", 0.15, float)):
            kwargs.setdefault(name, cast(fc.get(name, default)))
        per = fc.get("per_param_tail_bounds", None)
        kwargs.setdefault("tail_bound", per if per else float(fc.get("tail_bound", FLOW_NORM_BOUND)))
    kwargs.setdefault("num_layers", 12)
    kwargs.setdefault("hidden_features", 256)
    kwargs.setdefault("num_bins", 16)
    kwargs.setdefault("tail_bound", FLOW_NORM_BOUND)
    kwargs.setdefault("dropout", 0.15)
    if flow_type.lower() != "nsf":
        raise ValueError(f"Unknown flow_type: {flow_type}. Only 'nsf' is supported.")
    return NSFPosteriorFlow(features=features, context_features=context_features,
                            max_overlaps=max_overlaps, **kwargs)
