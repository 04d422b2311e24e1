"""Interim backward of the flow (DESIGN.md "Backward"): the FORWARD value of every differentiable
call comes from the HIP kernel (``pf_flow_forward``); ``backward`` re-evaluates the layer chain with
device tensor ops under autograd and differentiates that (SURVEY.md 7.1 step 6 allows exactly this
until the hand-written RQS / masked-MLP backward kernels exist).  Nothing here runs on the CPU and
nothing imports the oracle.

The tensor-op evaluation follows nflows (MADE with GLU-gated residual blocks, rational-quadratic
spline with linear tails, ReversePermutation before every layer) but is written sync-free: no boolean
indexing, the tail branch is a ``torch.where`` over a computation done on clamped inputs so that the
unselected branch is always finite (no NaN gradients)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

_MIN = 1e-3            # nflows DEFAULT_MIN_BIN_WIDTH / HEIGHT / DERIVATIVE


def _knots(u, tail_bound):
    """raw bin sizes [..., K] -> cumulative knots [..., K+1] on [-tb, tb], ends pinned."""
    k = u.shape[-1]
    sizes = _MIN + (1.0 - _MIN * k) * F.softmax(u, dim=-1)
    cum = torch.cumsum(sizes, dim=-1)
    cum = F.pad(cum, (1, 0), value=0.0) * (2.0 * tail_bound) - tail_bound
    edge_lo = torch.full_like(cum[..., :1], -tail_bound)
    edge_hi = torch.full_like(cum[..., :1], tail_bound)
    return torch.cat([edge_lo, cum[..., 1:-1], edge_hi], dim=-1)


def rqs_forward(x, uw, uh, ud, tail_bound):
    """x [B, D]; uw, uh [B, D, K]; ud [B, D, K-1] -> (y [B, D], logabsdet [B, D])."""
    k = uw.shape[-1]
    inside = (x >= -tail_bound) & (x <= tail_bound)
    xc = x.clamp(-tail_bound, tail_bound)
    kx, ky = _knots(uw, tail_bound), _knots(uh, tail_bound)
    const = math.log(math.exp(1.0 - _MIN) - 1.0)
    ud = F.pad(ud, (1, 1), value=const)
    d = _MIN + F.softplus(ud)
    search = kx.detach().clone()
    search[..., -1] += 1e-6
    idx = ((xc[..., None] >= search).sum(dim=-1) - 1).clamp(0, k - 1)[..., None]
    xl, xr = kx.gather(-1, idx)[..., 0], kx.gather(-1, idx + 1)[..., 0]
    yl, yr = ky.gather(-1, idx)[..., 0], ky.gather(-1, idx + 1)[..., 0]
    dl, dr = d.gather(-1, idx)[..., 0], d.gather(-1, idx + 1)[..., 0]
    w, h = xr - xl, yr - yl
    delta = h / w
    th = (xc - xl) / w
    tt = th * (1.0 - th)
    den = delta + (dl + dr - 2.0 * delta) * tt
    y = yl + h * (delta * th * th + dl * tt) / den
    dnum = delta * delta * (dr * th * th + 2.0 * delta * tt + dl * (1.0 - th) * (1.0 - th))
    lad = torch.log(dnum) - 2.0 * torch.log(den)
    return torch.where(inside, y, x), torch.where(inside, lad, torch.zeros_like(lad))


def made_forward(net, x, ctx, additive=False):
    """nflows MADE over the parameter container ``flows._MADE`` (weights * masks); context enters
    through GLU gates (nflows) or, for the reference's masked-context variant, additively between the
    two masked linears of a block (flows.py:225-234)."""
    lin = lambda m, v: F.linear(v, m.weight * m.mask, m.bias)
    cl = lambda m, v: F.linear(v, m.weight * m.mask, m.bias) if hasattr(m, "mask") else m(v)
    h = lin(net.initial_layer, x)
    if ctx is not None:
        h = h + F.relu(cl(net.context_layer, ctx))
    for blk in net.blocks:
        t = lin(blk.linear_layers[0], F.relu(h))
        if additive and ctx is not None:
            t = t + cl(blk.context_layer, ctx)
        t = lin(blk.linear_layers[1], F.relu(t))
        if ctx is not None and not additive:
            t = t * torch.sigmoid(blk.context_layer(ctx))
        h = h + t
    return lin(net.final_layer, h)


def flow_forward(flow, x, ctx):
    """(z, logdet) of ``NSFPosteriorFlow`` with tensor ops (autograd-differentiable)."""
    x = x[:, flow._ar_perm]
    k, d = flow.num_bins, flow.features
    logdet = x.new_zeros(x.shape[0])
    additive = bool(getattr(flow, "use_masked_context", False))
    for layer in flow._ar_transforms:
        if not additive:
            x = x.flip(1)                                               # ReversePermutation
        p = made_forward(layer.autoregressive_net, x, ctx, additive).view(-1, d, 3 * k - 1)
        x, lad = rqs_forward(x, p[..., :k], p[..., k:2 * k], p[..., 2 * k:], float(flow._tail_bound))
        logdet = logdet + lad.sum(dim=1)
    return x, logdet


class FlowNLL(torch.autograd.Function):
    """nll[B] = -(log N(z; 0, diag(e^ls)^2) + logdet); forward on the HIP kernel."""

    @staticmethod
    def forward(ctx_, flow, x, context, log_sigma, *params):
        with torch.no_grad():
            z, logdet, nll = flow._forward_call(x, context, log_sigma, want_z=True, guard=False)
        ctx_.flow = flow
        ctx_.save_for_backward(x, context, log_sigma)
        ctx_.mark_non_differentiable(z, logdet)
        return nll, z, logdet

    @staticmethod
    def backward(ctx_, g_nll, _gz, _gld):
        flow = ctx_.flow
        x, context, log_sigma = ctx_.saved_tensors
        params = [p for p in flow._ordered_parameters()]
        with torch.enable_grad():
            xs = x.detach().requires_grad_(x.requires_grad)
            cs = None if context is None else context.detach().requires_grad_(context.requires_grad)
            ls = None if log_sigma is None else log_sigma.detach().requires_grad_(log_sigma.requires_grad)
            z, logdet = flow_forward(flow, xs, cs)
            if ls is None:
                logp = -0.5 * (z.square().sum(dim=1) + flow.features * math.log(2.0 * math.pi))
            else:
                logp = -0.5 * ((z * torch.exp(-ls)).square().sum(dim=1) + 2.0 * ls.sum(dim=1)
                               + flow.features * math.log(2.0 * math.pi))
            nll = -(logp + logdet)
            wanted = [t for t in (xs, cs, ls) if t is not None and t.requires_grad]
            wanted += [p for p in params if p.requires_grad]
            grads = torch.autograd.grad(nll, wanted, grad_outputs=g_nll, allow_unused=True)
        it = iter(grads)
        gx = next(it) if xs.requires_grad else None
        gc = next(it) if cs is not None and cs.requires_grad else None
        gl = next(it) if ls is not None and ls.requires_grad else None
        gp = [next(it) if p.requires_grad else None for p in params]
        return (None, gx, gc, gl, *gp)


class FlowForward(torch.autograd.Function):
    """(z, logdet) = flow.forward(x, context); forward on the HIP kernel."""

    @staticmethod
    def forward(ctx_, flow, x, context, *params):
        with torch.no_grad():
            z, logdet, _ = flow._forward_call(x, context, None, want_z=True, guard=False)
        ctx_.flow = flow
        ctx_.save_for_backward(x, context)
        return z, logdet

    @staticmethod
    def backward(ctx_, gz, gld):
        flow = ctx_.flow
        x, context = ctx_.saved_tensors
        params = [p for p in flow._ordered_parameters()]
        with torch.enable_grad():
            xs = x.detach().requires_grad_(x.requires_grad)
            cs = None if context is None else context.detach().requires_grad_(context.requires_grad)
            z, logdet = flow_forward(flow, xs, cs)
            wanted = [t for t in (xs, cs) if t is not None and t.requires_grad]
            wanted += [p for p in params if p.requires_grad]
            grads = torch.autograd.grad([z, logdet], wanted, grad_outputs=[gz, gld], allow_unused=True)
        it = iter(grads)
        gx = next(it) if xs.requires_grad else None
        gc = next(it) if cs is not None and cs.requires_grad else None
        gp = [next(it) if p.requires_grad else None for p in params]
        return (None, gx, gc, *gp)
