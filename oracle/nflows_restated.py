"""CPU restatement of the nflows pieces PosteriFlow's flow is built from.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Pure torch, runs in fp32 or
fp64 (``module.double()``), executes the *reference's* algorithm unchanged:
one dense ``F.linear(x, W * mask, b)`` per masked layer, un-fused spline ops,
boolean-mask tail handling, and the D-pass autoregressive inverse.

The algorithm lives in the third-party package ``nflows`` (absent here; the
reference lists it un-pinned at ``environment.yaml:35``; 0.14 is the latest
release).  Each piece below names the upstream symbol it restates and the
reference call site that fixes how it is used:

  reference construction   src/ahsd/models/flows.py:459-460, 510-525, 529, 532
  reference execution      src/ahsd/models/flows.py:615-617 (forward), 637 (inverse)
  in-tree near-copies of the upstream MADE wiring (degree / mask / init rules)
                           src/ahsd/models/flows.py:203-223, 258-303

Module and parameter names follow upstream so that a ``state_dict`` is
interchangeable (SURVEY.md section 8a "state_dict layout").

PARITY UNPINNED for this file: the reference holds no golden vector, checkpoint or
fixture of the flow transform and nflows itself is not installed here, so nothing
reference-held pins this restatement.  What stands in: the known-answer tests of
tests/test_oracle_kat.py (round trip, fp64 slogdet of the Jacobian, autoregressive
structure, normalisation), and scripts/compare_with_nflows.py, which builds the
reference's exact transform from a real nflows where one is importable and compares
forward / inverse to 1e-6 (exit 77 here).  The own-code parts of the reference
(PSDScaledNormal, masks of the masked-context variant, ParamScaler, encoders) ARE
pinned by reference-generated fixtures under tests/golden/.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

import contextlib

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_MIN_DERIVATIVE = 1e-3


# --------------------------------------------------------------------------- #
# optional emulation of the HIP bf16 mode's operand rounding (NOT part of the
# reference algorithm): operands of every conditioner GEMM are rounded to bf16
# (round-to-nearest-even), products accumulate in fp32, biases / residual state
# stay fp32; the initial layer's x is split hi + lo (two bf16 terms) exactly as
# the kernel does (csrc/pf_flow_fwd.hip).  Lets the bf16 kernel be checked
# tightly against "the same arithmetic on the CPU".
# --------------------------------------------------------------------------- #
_EMULATE = None


@contextlib.contextmanager
def gemm_emulation(mode):
    global _EMULATE
    prev, _EMULATE = _EMULATE, mode
    try:
        yield
    finally:
        _EMULATE = prev


def _bf16(t):
    return t.to(torch.bfloat16).to(t.dtype)


def round_projection(t):
    """hoisted-context bf16 mode stores relu / sigmoid of the context projections as bf16"""
    return _bf16(t) if _EMULATE == "bf16_hoist" else t


def linear(x, w, b, split_input=False):
    if _EMULATE in ("bf16", "bf16_hoist"):
        if split_input:
            hi = _bf16(x)
            x = hi + _bf16(x - hi)
        else:
            x = _bf16(x)
        w = _bf16(w)
    return F.linear(x, w, b)


# --------------------------------------------------------------------------- #
# degrees and masks  (upstream nflows.transforms.made: _get_input_degrees,
# MaskedLinear._get_mask_and_degrees; used verbatim at flows.py:208-215, 260-294)
# --------------------------------------------------------------------------- #
def input_degrees(features: int) -> torch.Tensor:
    return torch.arange(1, features + 1)


def mask_and_degrees(in_degrees: torch.Tensor, out_features: int,
                     autoregressive_features: int, is_output: bool):
    """random_mask=False branch only (flows.py:520 passes random_mask=False)."""
    if is_output:
        # upstream torchutils.tile(degrees, m): each degree repeated m times,
        # i.e. output index = feature * multiplier + j   ([D, M] contiguous)
        mult = out_features // autoregressive_features
        out_degrees = input_degrees(autoregressive_features).repeat_interleave(mult)
        mask = (out_degrees[:, None] > in_degrees[None, :]).float()
    else:
        hi = max(1, autoregressive_features - 1)
        lo = min(1, autoregressive_features - 1)
        out_degrees = torch.arange(out_features) % hi + lo
        mask = (out_degrees[:, None] >= in_degrees[None, :]).float()
    return mask, out_degrees


class MaskedLinear(nn.Linear):
    """upstream made.MaskedLinear: ``F.linear(x, weight * mask, bias)``."""

    split_input = False   # set on MADE.initial_layer (bf16 emulation only)

    def __init__(self, in_degrees, out_features, autoregressive_features, is_output):
        super().__init__(len(in_degrees), out_features, bias=True)
        mask, degrees = mask_and_degrees(in_degrees, out_features,
                                         autoregressive_features, is_output)
        self.register_buffer("mask", mask)
        self.register_buffer("degrees", degrees)

    def forward(self, x):
        return linear(x, self.weight * self.mask, self.bias, self.split_input)


class MaskedResidualBlock(nn.Module):
    """upstream made.MaskedResidualBlock (no batch norm; flows.py:523).

    t = W1m . drop(relu(W0m . relu(h)));  t = glu([t, ctx_layer(ctx)]);  h + t
    """

    def __init__(self, in_degrees, autoregressive_features, context_features,
                 dropout_probability=0.0):
        super().__init__()
        features = len(in_degrees)
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, features)
        l0 = MaskedLinear(in_degrees, features, autoregressive_features, False)
        l1 = MaskedLinear(l0.degrees, features, autoregressive_features, False)
        self.linear_layers = nn.ModuleList([l0, l1])
        self.degrees = l1.degrees
        self.dropout = nn.Dropout(p=dropout_probability)
        # zero_initialization=True upstream (same rule at flows.py:221-223)
        nn.init.uniform_(self.linear_layers[-1].weight, -1e-3, 1e-3)
        nn.init.uniform_(self.linear_layers[-1].bias, -1e-3, 1e-3)

    def forward(self, inputs, context=None):
        t = F.relu(inputs)
        t = self.linear_layers[0](t)
        t = F.relu(t)
        t = self.dropout(t)
        t = self.linear_layers[1](t)
        if context is not None:
            gate = linear(context, self.context_layer.weight, self.context_layer.bias)
            if _EMULATE == "bf16_hoist":
                t = t * round_projection(torch.sigmoid(gate))
            else:
                t = F.glu(torch.cat((t, gate), dim=1), dim=1)
        return inputs + t


class MADE(nn.Module):
    """upstream made.MADE with use_residual_blocks=True, random_mask=False."""

    def __init__(self, features, hidden_features, context_features=None,
                 num_blocks=2, output_multiplier=1, dropout_probability=0.0):
        super().__init__()
        self.initial_layer = MaskedLinear(input_degrees(features), hidden_features,
                                          features, False)
        self.initial_layer.split_input = True
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, hidden_features)
        blocks, prev = [], self.initial_layer.degrees
        for _ in range(num_blocks):
            blocks.append(MaskedResidualBlock(prev, features, context_features,
                                              dropout_probability))
            prev = blocks[-1].degrees
        self.blocks = nn.ModuleList(blocks)
        self.final_layer = MaskedLinear(prev, features * output_multiplier,
                                        features, True)

    def forward(self, inputs, context=None):
        h = self.initial_layer(inputs)
        if context is not None:
            h = h + round_projection(F.relu(linear(context, self.context_layer.weight, self.context_layer.bias)))
        for block in self.blocks:
            h = block(h, context)
        return self.final_layer(h)


# --------------------------------------------------------------------------- #
# masked-context conditioner -- the reference's OWN variant (src/ahsd/models/flows.py:112-303):
# context enters ADDITIVELY between the two masked linears of a block through a masked linear
# over n_blocks x block_dim context columns (mask all-ones unless full_context=False).
# --------------------------------------------------------------------------- #
class MaskedContextLinear(nn.Module):
    """flows.py:112-183"""

    def __init__(self, n_blocks, block_dim, hidden_degrees, full_context=True):
        super().__init__()
        h = hidden_degrees.shape[0]
        self.weight = nn.Parameter(torch.empty(h, n_blocks * block_dim))
        self.bias = nn.Parameter(torch.zeros(h))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if full_context:
            in_deg = torch.full((n_blocks * block_dim,), -1, dtype=torch.long)
        else:
            in_deg = torch.arange(n_blocks).repeat_interleave(block_dim)
        self.register_buffer("mask", (hidden_degrees[:, None] >= in_deg[None, :]).float())

    def forward(self, ctx):
        return linear(ctx, self.weight * self.mask, self.bias)


class MaskedContextResidualBlock(nn.Module):
    """flows.py:186-234: relu, W0, + ctx, relu, dropout, W1, residual add (no gate)."""

    def __init__(self, in_degrees, autoregressive_features, n_blocks, block_dim, dropout_probability=0.0,
                 full_context=True):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout_probability)               # flows.py:219
        features = len(in_degrees)
        self.context_layer = MaskedContextLinear(n_blocks, block_dim, in_degrees, full_context)
        l0 = MaskedLinear(in_degrees, features, autoregressive_features, False)
        l1 = MaskedLinear(l0.degrees, features, autoregressive_features, False)
        self.linear_layers = nn.ModuleList([l0, l1])
        self.degrees = l1.degrees
        nn.init.uniform_(l1.weight, -1e-3, 1e-3)
        nn.init.uniform_(l1.bias, -1e-3, 1e-3)

    def forward(self, inputs, context=None):
        t = self.linear_layers[0](F.relu(inputs))
        if context is not None:
            t = t + self.context_layer(context)
        t = self.linear_layers[1](self.dropout(F.relu(t)))             # flows.py:231-233
        return inputs + t


class MADEWithMaskedContext(nn.Module):
    """flows.py:237-303"""

    def __init__(self, features, hidden_features, n_blocks, block_dim, num_blocks=2, output_multiplier=1,
                 dropout_probability=0.0, full_context=True):
        super().__init__()
        self.initial_layer = MaskedLinear(input_degrees(features), hidden_features, features, False)
        self.initial_layer.split_input = True
        deg = self.initial_layer.degrees
        # the reference never passes full_context (flows.py:268, 275-283): True everywhere; False kept reachable for the
        # product's keyword of the same name
        self.context_layer = MaskedContextLinear(n_blocks, block_dim, deg, full_context)
        self.blocks = nn.ModuleList([MaskedContextResidualBlock(deg, features, n_blocks, block_dim, dropout_probability,
                                                                full_context) for _ in range(num_blocks)])
        self.final_layer = MaskedLinear(deg, features * output_multiplier, features, True)

    def forward(self, inputs, context=None):
        h = self.initial_layer(inputs)
        if context is not None:
            h = h + F.relu(self.context_layer(context))
        for block in self.blocks:
            h = block(h, context)
        return self.final_layer(h)


# --------------------------------------------------------------------------- #
# rational-quadratic spline (upstream nflows.transforms.splines.rational_quadratic,
# nflows.utils.torchutils.searchsorted)
# --------------------------------------------------------------------------- #
def searchsorted(bin_locations, inputs, eps=1e-6):
    bin_locations[..., -1] += eps
    return torch.sum(inputs[..., None] >= bin_locations, dim=-1) - 1


def rational_quadratic_spline(inputs, unnormalized_widths, unnormalized_heights,
                              unnormalized_derivatives, inverse=False,
                              left=0.0, right=1.0, bottom=0.0, top=1.0,
                              min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                              min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
                              min_derivative=DEFAULT_MIN_DERIVATIVE):
    if inputs.numel() and (torch.min(inputs) < left or torch.max(inputs) > right):
        raise ValueError("input outside spline domain")
    num_bins = unnormalized_widths.shape[-1]
    if min_bin_width * num_bins > 1.0 or min_bin_height * num_bins > 1.0:
        raise ValueError("minimal bin size too large for the number of bins")

    widths = F.softmax(unnormalized_widths, dim=-1)
    widths = min_bin_width + (1 - min_bin_width * num_bins) * widths
    cumwidths = torch.cumsum(widths, dim=-1)
    cumwidths = F.pad(cumwidths, pad=(1, 0), mode="constant", value=0.0)
    cumwidths = (right - left) * cumwidths + left
    cumwidths[..., 0] = left
    cumwidths[..., -1] = right
    widths = cumwidths[..., 1:] - cumwidths[..., :-1]

    derivatives = min_derivative + F.softplus(unnormalized_derivatives)

    heights = F.softmax(unnormalized_heights, dim=-1)
    heights = min_bin_height + (1 - min_bin_height * num_bins) * heights
    cumheights = torch.cumsum(heights, dim=-1)
    cumheights = F.pad(cumheights, pad=(1, 0), mode="constant", value=0.0)
    cumheights = (top - bottom) * cumheights + bottom
    cumheights[..., 0] = bottom
    cumheights[..., -1] = top
    heights = cumheights[..., 1:] - cumheights[..., :-1]

    if inverse:
        bin_idx = searchsorted(cumheights, inputs)[..., None]
    else:
        bin_idx = searchsorted(cumwidths, inputs)[..., None]

    in_cumwidths = cumwidths.gather(-1, bin_idx)[..., 0]
    in_widths = widths.gather(-1, bin_idx)[..., 0]
    in_cumheights = cumheights.gather(-1, bin_idx)[..., 0]
    delta = heights / widths
    in_delta = delta.gather(-1, bin_idx)[..., 0]
    in_d = derivatives.gather(-1, bin_idx)[..., 0]
    in_d1 = derivatives[..., 1:].gather(-1, bin_idx)[..., 0]
    in_heights = heights.gather(-1, bin_idx)[..., 0]

    if inverse:
        dy = inputs - in_cumheights
        a = dy * (in_d + in_d1 - 2 * in_delta) + in_heights * (in_delta - in_d)
        b = in_heights * in_d - dy * (in_d + in_d1 - 2 * in_delta)
        c = -in_delta * dy
        disc = b.pow(2) - 4 * a * c
        assert (disc >= 0).all()
        root = (2 * c) / (-b - torch.sqrt(disc))
        outputs = root * in_widths + in_cumwidths
        tt = root * (1 - root)
        den = in_delta + (in_d + in_d1 - 2 * in_delta) * tt
        num = in_delta.pow(2) * (in_d1 * root.pow(2) + 2 * in_delta * tt
                                 + in_d * (1 - root).pow(2))
        logabsdet = torch.log(num) - 2 * torch.log(den)
        return outputs, -logabsdet

    theta = (inputs - in_cumwidths) / in_widths
    tt = theta * (1 - theta)
    numer = in_heights * (in_delta * theta.pow(2) + in_d * tt)
    den = in_delta + (in_d + in_d1 - 2 * in_delta) * tt
    outputs = in_cumheights + numer / den
    num = in_delta.pow(2) * (in_d1 * theta.pow(2) + 2 * in_delta * tt
                             + in_d * (1 - theta).pow(2))
    logabsdet = torch.log(num) - 2 * torch.log(den)
    return outputs, logabsdet


def unconstrained_rational_quadratic_spline(inputs, unnormalized_widths,
                                            unnormalized_heights,
                                            unnormalized_derivatives,
                                            inverse=False, tail_bound=1.0,
                                            min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                                            min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
                                            min_derivative=DEFAULT_MIN_DERIVATIVE):
    """tails='linear' (flows.py:516): identity and zero log-det outside
    [-tail_bound, tail_bound]; boundary derivatives pinned to exactly 1."""
    inside = (inputs >= -tail_bound) & (inputs <= tail_bound)
    outside = ~inside
    outputs = torch.zeros_like(inputs)
    logabsdet = torch.zeros_like(inputs)

    unnormalized_derivatives = F.pad(unnormalized_derivatives, pad=(1, 1))
    constant = math.log(math.exp(1 - min_derivative) - 1)
    unnormalized_derivatives[..., 0] = constant
    unnormalized_derivatives[..., -1] = constant

    outputs[outside] = inputs[outside]
    logabsdet[outside] = 0
    if torch.any(inside):
        outputs[inside], logabsdet[inside] = rational_quadratic_spline(
            inputs=inputs[inside],
            unnormalized_widths=unnormalized_widths[inside, :],
            unnormalized_heights=unnormalized_heights[inside, :],
            unnormalized_derivatives=unnormalized_derivatives[inside, :],
            inverse=inverse,
            left=-tail_bound, right=tail_bound, bottom=-tail_bound, top=tail_bound,
            min_bin_width=min_bin_width, min_bin_height=min_bin_height,
            min_derivative=min_derivative)
    return outputs, logabsdet


# --------------------------------------------------------------------------- #
# transforms
# --------------------------------------------------------------------------- #
class ReversePermutation(nn.Module):
    """upstream permutations.ReversePermutation (flows.py:460): x[:, ::-1]."""

    def __init__(self, features):
        super().__init__()
        self.register_buffer("_permutation", torch.arange(features - 1, -1, -1))

    def forward(self, inputs, context=None):
        return (torch.index_select(inputs, 1, self._permutation),
                inputs.new_zeros(inputs.shape[0]))

    def inverse(self, inputs, context=None):
        inv = torch.argsort(self._permutation)
        return torch.index_select(inputs, 1, inv), inputs.new_zeros(inputs.shape[0])


class MaskedPiecewiseRationalQuadraticAutoregressiveTransform(nn.Module):
    """upstream autoregressive.MaskedPiecewiseRationalQuadraticAutoregressive-
    Transform with tails='linear', as constructed at flows.py:510-525."""

    def __init__(self, features, hidden_features, context_features=None,
                 num_bins=10, tail_bound=1.0, num_blocks=2,
                 dropout_probability=0.0, scale_by_sqrt_hidden=False, masked_context_blocks=None,
                 min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
                 min_derivative=DEFAULT_MIN_DERIVATIVE):
        super().__init__()
        self.features = features
        self.hidden_features = hidden_features
        self.num_bins = num_bins
        self.tail_bound = tail_bound
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        # upstream divides widths/heights by sqrt(hidden) only if the conditioner
        # exposes a `hidden_features` attribute; upstream MADE does not
        #  -> inactive by default, switchable for audit.
        self.scale_by_sqrt_hidden = scale_by_sqrt_hidden
        if masked_context_blocks:     # (n_blocks, block_dim): flows.py:306-360
            self.autoregressive_net = MADEWithMaskedContext(features, hidden_features, *masked_context_blocks[:2],
                                                            num_blocks, self.output_multiplier(), dropout_probability,
                                                            *masked_context_blocks[2:])
        else:
            self.autoregressive_net = MADE(features, hidden_features, context_features,
                                           num_blocks, self.output_multiplier(),
                                           dropout_probability)

    def output_multiplier(self):
        return 3 * self.num_bins - 1

    def _elementwise(self, inputs, params, inverse):
        B, D = inputs.shape
        p = params.view(B, D, self.output_multiplier())
        uw = p[..., : self.num_bins]
        uh = p[..., self.num_bins: 2 * self.num_bins]
        ud = p[..., 2 * self.num_bins:]
        if self.scale_by_sqrt_hidden:
            uw = uw / math.sqrt(self.hidden_features)
            uh = uh / math.sqrt(self.hidden_features)
        out, lad = unconstrained_rational_quadratic_spline(
            inputs, uw, uh, ud, inverse=inverse, tail_bound=self.tail_bound,
            min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
            min_derivative=self.min_derivative)
        return out, lad.reshape(B, -1).sum(dim=1)

    def forward(self, inputs, context=None):
        params = self.autoregressive_net(inputs, context)
        return self._elementwise(inputs, params, inverse=False)

    def inverse(self, inputs, context=None):
        """upstream AutoregressiveTransform.inverse: D full conditioner passes,
        the log-det returned is the one of the last pass."""
        outputs = torch.zeros_like(inputs)
        logabsdet = None
        for _ in range(inputs.shape[1]):
            params = self.autoregressive_net(outputs, context)
            outputs, logabsdet = self._elementwise(inputs, params, inverse=True)
        return outputs, logabsdet


class CompositeTransform(nn.Module):
    """upstream base.CompositeTransform (flows.py:529)."""

    def __init__(self, transforms):
        super().__init__()
        self._transforms = nn.ModuleList(transforms)

    @staticmethod
    def _cascade(inputs, funcs, context):
        outputs = inputs
        total = inputs.new_zeros(inputs.shape[0])
        for f in funcs:
            outputs, lad = f(outputs, context)
            total = total + lad
        return outputs, total

    def forward(self, inputs, context=None):
        return self._cascade(inputs, self._transforms, context)

    def inverse(self, inputs, context=None):
        return self._cascade(inputs, (t.inverse for t in self._transforms[::-1]), context)
