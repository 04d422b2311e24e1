"""Backward of the flow.  The FORWARD value of every differentiable call comes from the HIP kernel
(``pf_flow_forward_train``), which also keeps the input of every layer's conditioner (``U [L, B, D]``).

``flow_backward`` (both conditioner flavours, every precision) is three groups of HIP launches, no library GEMM:
  1. ``pf_flow_reevaluate``: the conditioners of all layers re-evaluated from the kept inputs in one launch
     (bf16: ``flow_reeval_kernel`` on the packed ``PF_FLAG_BWD`` stream; fp32, generic shapes and the masked-context
     conditioner: the generic kernel's conditioner, ``PF_FLAG_GENERIC`` layout);
  2. ``pf_flow_backward_chain``: the layer chain walked backwards in ONE launch -- the hand-derived spline backward, the
     transposed masked GEMMs on MFMA and the gate / ReLU algebra (additive form for the masked-context conditioner);
  3. ``pf_dense_tn``: every weight / bias gradient by the hand-written transposed GEMM, batched over layers, accumulated
     into ONE flat fp32 buffer in ``state_dict`` order; ``pf_dense_nt``: the context gradient by the strip GEMM over the
     transposed context weights.
Nothing runs on the CPU and nothing imports the oracle.

``flow_forward`` below is the same chain written with device tensor ops under autograd (nflows' MADE with GLU-gated
residual blocks, rational-quadratic spline with linear tails, ReversePermutation before every layer; sync-free: no
boolean indexing, the tail branch is a ``torch.where`` over a computation on clamped inputs so that the unselected branch
is finite).  It is NOT on the default path: it is the replay used only for masked-context flows outside the HIP
backward's shapes (``_fast``: D > 16 or H not in the instantiated set), and taking it is logged once."""
from __future__ import annotations

import logging
import math

import torch
import torch.nn.functional as F

from . import _lib

_log = logging.getLogger(__name__)
_replay_logged = False


def _note_replay(flow, why: str) -> None:
    """the tensor-op replay is taken: say so once per process"""
    global _replay_logged
    if not _replay_logged:
        _replay_logged = True
        _log.warning("posteriflow_amd: flow backward by autograd replay over device tensor ops (%s; D=%d, H=%d, K=%d, "
                     "masked_context=%s) -- the HIP backward does not cover this shape", why, flow.features,
                     flow.hidden_features, flow.num_bins, bool(getattr(flow, "use_masked_context", False)))

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


def made_forward(net, x, ctx, additive=False, drop=None):
    """nflows MADE over the parameter container ``flows._MADE`` (weights * masks); context enters
    through GLU gates (nflows) or, for the reference's masked-context variant, additively between the
    two masked linears of a block (flows.py:225-234).  ``drop``: per-block dropout factors [nb][B, H]
    (0 or 1 / (1 - p), ``dropout_mask``) applied where both put their nn.Dropout, or None."""
    lin = lambda m, v: F.linear(v, m.weight * m.mask, m.bias)
    cl = lambda m, v: F.linear(v, m.weight * m.mask, m.bias) if hasattr(m, "mask") else m(v)
    h = lin(net.initial_layer, x)
    if ctx is not None:
        h = h + F.relu(cl(net.context_layer, ctx))
    for j, blk in enumerate(net.blocks):
        t = lin(blk.linear_layers[0], F.relu(h))
        if additive and ctx is not None:
            t = t + cl(blk.context_layer, ctx)
        t = F.relu(t)
        if drop is not None:
            t = t * drop[j]
        t = lin(blk.linear_layers[1], t)
        if ctx is not None and not additive:
            t = t * torch.sigmoid(blk.context_layer(ctx))
        h = h + t
    return lin(net.final_layer, h)


def dropout_mask(flow, batch, seed, device):
    """the factors flow_train_kernel applied with this seed: fp32 [2, L, batch, H], 0 or 1 / (1 - p)
    (block, layer, row, hidden unit in nflows order); pf_flow_dropout_mask evaluates the kernel's counter hash."""
    m = torch.empty(2, flow.num_layers, batch, flow.hidden_features, dtype=torch.float32, device=device)
    _lib.check(_lib.lib().pf_flow_dropout_mask(flow._desc(), float(flow.dropout), int(seed), batch, m.data_ptr(),
                                               torch.cuda.current_stream(device).cuda_stream), "pf_flow_dropout_mask")
    return m


def flow_forward(flow, x, ctx, drop=None):
    """(z, logdet) of ``NSFPosteriorFlow`` with tensor ops (autograd-differentiable); ``drop``: ``dropout_mask``."""
    x = x[:, flow._ar_perm]
    k, d = flow.num_bins, flow.features
    logdet = x.new_zeros(x.shape[0])
    additive = bool(getattr(flow, "use_masked_context", False))
    for l, layer in enumerate(flow._ar_transforms):
        if not additive:
            x = x.flip(1)                                               # ReversePermutation
        dl = None if drop is None else drop[:, l]
        p = made_forward(layer.autoregressive_net, x, ctx, additive, dl).view(-1, d, 3 * k - 1)
        x, lad = rqs_forward(x, p[..., :k], p[..., k:2 * k], p[..., 2 * k:], float(flow._tail_bound))
        logdet = logdet + lad.sum(dim=1)
    return x, logdet


# ---- fast path: HIP spline backward + layer-batched GEMMs ---------------------------------------------
def _rqs_backward(flow, u, params, gy, glad, gparams):
    """pf_flow_rqs_backward on [B, D] pairs; writes dL/d(raw parameters) into ``gparams`` and returns the
    direct part of dL/du."""
    gu = torch.empty_like(u)
    _lib.check(_lib.lib().pf_flow_rqs_backward(
        flow._desc(), u.data_ptr(), params.data_ptr(), gy.data_ptr(), glad.data_ptr(), u.shape[0],
        gparams.data_ptr(), gu.data_ptr(), torch.cuda.current_stream(u.device).cuda_stream), "pf_flow_rqs_backward")
    return gu


REEVAL_HIP = True      # tests: False keeps the tensor-op re-evaluation in bf16 mode (isolates the bf16 chain kernel)
COMPACT = True         # tests: False keeps fp32 activations / gradient vectors between the bf16 backward kernels


def _reevaluate_hip(flow, U, ctx, drop=None, compact=False, fp32=False):
    """Every layer's conditioner from its kept input in ONE launch (pf_flow_reevaluate): bf16 mode in the arithmetic of the
    bf16 forward kernel (csrc/pf_flow_reeval.hip); ``fp32``: exact-fp32 MFMA by the generic kernel's conditioner
    (csrc/pf_flow_generic.hip, the PF_FLAG_GENERIC layout).  Returns (hs, t1s, t2s, gates, pc, h2, params): fp32 raw values, or
    with ``compact`` (bf16 only) bf16 tensors already in the form the backward uses them (hs = relu(h_j), t1s = relu(t1_j) .
    drop[j], pc = relu(pc); params stay fp32)."""
    Ln, B, D = U.shape
    H, dev = flow.hidden_features, U.device
    has_ctx = ctx is not None
    act = torch.bfloat16 if compact else torch.float32
    new = lambda *shape: torch.empty(*shape, dtype=act, device=dev)
    HS, T1, H2 = new(2, Ln, B, H), new(2, Ln, B, H), new(Ln, B, H)
    params = torch.empty(Ln, B, D * (3 * flow.num_bins - 1), dtype=torch.float32, device=dev)
    T2 = G = PC = None
    packed = flow.packed_weights("fp32", generic=True) if fp32 else flow.packed_weights(bwd=True)
    a = _lib.PfFlowReevalArgs()
    a.batch, a.packed, a.U = B, packed.data_ptr(), U.data_ptr()
    a.hs, a.t1s, a.h2, a.params = HS.data_ptr(), T1.data_ptr(), H2.data_ptr(), params.data_ptr()
    if has_ctx:
        T2, PC = new(2, Ln, B, H), new(Ln, B, H)
        a.ctx, a.t2s, a.pc = ctx.data_ptr(), T2.data_ptr(), PC.data_ptr()
        if not flow.use_masked_context:          # GLU gates (the masked-context conditioner adds its projections instead)
            G = new(2, Ln, B, H)
            a.gates = G.data_ptr()
    if drop is not None:
        a.drop = drop.data_ptr()
    a.compact = 1 if compact else 0
    _lib.check(_lib.lib().pf_flow_reevaluate(flow._desc("fp32" if fp32 else "bf16"), a, torch.cuda.current_stream(dev).cuda_stream),
               "pf_flow_reevaluate")
    return HS, T1, T2, G, PC, H2, params


@torch.no_grad()
def _flow_backward_batched(flow, U, ctx, g_z, g_lad, drop=None, nll=None):
    """U [L, B, D] conditioner inputs kept by the forward kernel; g_z [B, D] = dL/dz, g_lad [B] =
    dL/dlogdet -- or both None and nll = (g_nll [B], z [B, D], log_sigma [B, D] or None): the chain kernel then forms
    dL/dz = g_nll z e^{-2 ls} and dL/dlogdet = -g_nll itself; drop: the forward's dropout factors (``dropout_mask``) or
    None.  Returns dL/dx [B, D], dL/dctx (or None) and the parameter gradients batched over layers."""
    nets = [t.autoregressive_net for t in flow._ar_transforms]
    Ln, B, D = U.shape
    nb = len(nets[0].blocks)
    has_ctx = ctx is not None
    H = flow.hidden_features
    st = lambda f: torch.stack([f(n) for n in nets])
    net0 = nets[0]                                   # every layer has the same degrees, hence masks
    m0, mf = net0.initial_layer.mask, net0.final_layer.mask
    m1 = [net0.blocks[j].linear_layers[0].mask for j in range(nb)]
    m2 = [net0.blocks[j].linear_layers[1].mask for j in range(nb)]
    if has_ctx:
        C = ctx.shape[1]
    Wcat_ = []

    def wcat():      # [L, 1 + nb, H, C] context weights stacked (tensor-op re-evaluation / library fallback only)
        if not Wcat_:
            cw = lambda m: m.masked_weight() if hasattr(m, "masked_weight") else m.weight
            Wcat_.append(torch.cat([st(lambda n: cw(n.context_layer))]
                                   + [st(lambda n: cw(n.blocks[j].context_layer)) for j in range(nb)], dim=1))
        return Wcat_[0]
    # bf16 mode: re-evaluation and chain are HIP kernels reading the packed PF_FLAG_BWD stream (one gather per weight update)
    generic = flow._generic_shape()      # shapes of the generic forward kernel: fp32 re-evaluation + fp32 chain in either precision
    additive = bool(flow.use_masked_context) and has_ctx     # masked-context conditioner (flows.py:225-234): same, fp32 kernels
    bf = (flow.precision == "bf16" and H % 32 == 0 and nb == 2 and U.is_contiguous() and (ctx is None or ctx.is_contiguous())
          and not generic and not additive)

    # 1. conditioners
    cmp = bf and REEVAL_HIP and COMPACT
    # fp32 (parity) mode: the same launch in exact-fp32 MFMA arithmetic (the generic kernel's conditioner)
    f32_hip = ((not bf) and REEVAL_HIP and (flow.precision != "bf16" or generic or additive) and nb == 2 and U.is_contiguous()
               and (ctx is None or ctx.is_contiguous()))
    hip_reeval = (bf and REEVAL_HIP) or f32_hip
    if hip_reeval:
        try:
            re = _reevaluate_hip(flow, U, ctx, drop, compact=cmp, fp32=f32_hip)
        except NotImplementedError:      # PF_ERR_UNSUPPORTED (e.g. a context too wide for the kernel's LDS image):
            hip_reeval = cmp = f32_hip = False     # the tensor-op re-evaluation below, fp32 interface of the chain
    if hip_reeval and cmp:   # bf16 activations in their backward form; bf16 gradient vectors; bf16 weight-gradient GEMMs
        HSk, T1k, T2k, Gk, pck, h_last, params = re
        relu_h, a1s = [HSk[j] for j in range(nb)], [T1k[j] for j in range(nb)]
    elif hip_reeval:
        HSk, T1k, T2k, Gk, pck, h_last, params = re
        relu_h = [F.relu(HSk[j]) for j in range(nb)]
        a1s = [F.relu(T1k[j]) if drop is None else F.relu(T1k[j]) * drop[j] for j in range(nb)]
    if not bf or not hip_reeval:     # masked weights stacked over the layers: operands of the fp32 chain (transposed below) and of the
        #            tensor-op re-evaluation
        W0 = st(lambda n: n.initial_layer.weight) * m0
        Wf = st(lambda n: n.final_layer.weight) * mf
        W1 = [st(lambda n: n.blocks[j].linear_layers[0].weight) * m1[j] for j in range(nb)]
        W2 = [st(lambda n: n.blocks[j].linear_layers[1].weight) * m2[j] for j in range(nb)]
    if not hip_reeval:
        b0, bf_ = st(lambda n: n.initial_layer.bias), st(lambda n: n.final_layer.bias)
        b1 = [st(lambda n: n.blocks[j].linear_layers[0].bias) for j in range(nb)]
        b2 = [st(lambda n: n.blocks[j].linear_layers[1].bias) for j in range(nb)]
        if has_ctx:       # batched over layers: one GEMM for all context projections
            bcat = torch.cat([st(lambda n: n.context_layer.bias)]
                             + [st(lambda n: n.blocks[j].context_layer.bias) for j in range(nb)], dim=1)
            proj = torch.addmm(bcat.reshape(-1), ctx, wcat().reshape(-1, C).t())
            proj = proj.view(B, Ln, 1 + nb, H).permute(1, 2, 0, 3)                   # [L, 1+nb, B, H]
            pc = proj[:, 0]
            gates = [None if additive else torch.sigmoid(proj[:, 1 + j]) for j in range(nb)]
        h = torch.baddbmm(b0[:, None, :], U, W0.transpose(1, 2))
        if has_ctx:
            h = h + F.relu(pc)
        hs, t1s, t2s, a1s = [h], [], [], []
        for j in range(nb):
            t1 = torch.baddbmm(b1[j][:, None, :], F.relu(h), W1[j].transpose(1, 2))
            if additive:
                t1 = t1 + proj[:, 1 + j]
            a1 = F.relu(t1) if drop is None else F.relu(t1) * drop[j]                 # the second linear's input
            t2 = torch.baddbmm(b2[j][:, None, :], a1, W2[j].transpose(1, 2))
            h = h + (t2 * gates[j] if (has_ctx and not additive) else t2)
            hs.append(h), t1s.append(t1), t2s.append(t2), a1s.append(a1)
        params = torch.baddbmm(bf_[:, None, :], h, Wf.transpose(1, 2))                # [L, B, D(3K-1)]
        h_last, relu_h = hs[nb], [F.relu(hs[j]) for j in range(nb)]
        HSk, T1k = torch.stack(hs[:nb]), torch.stack(t1s)
        if has_ctx:
            T2k, Gk, pck = torch.stack(t2s), (None if additive else torch.stack(gates)), pc.contiguous()

    # 2. the chain, last layer first: ONE launch (pf_flow_backward_chain, csrc/pf_flow_bwd_chain.hip) instead of ~25
    #    small ones per layer -- spline backward, the transposed masked GEMMs on MFMA, the gate / ReLU algebra
    DM = params.shape[2]
    gdt = torch.bfloat16 if cmp else U.dtype
    DMp = (DM + 7) // 8 * 8 if cmp else (DM + 3) // 4 * 4    # rows padded to whole 16-byte pieces (the GEMM operand form)
    Gp = torch.empty(Ln, B, DMp, dtype=gdt, device=U.device)
    # Gt1 | Gt2 | Gh0 in one buffer: their bias gradients are ONE column reduction instead of five
    G5 = torch.empty(2 * nb + 1, Ln, B, H, dtype=gdt, device=U.device)
    GT1, GT2, Gh0 = G5[:nb], G5[nb:2 * nb], G5[2 * nb]
    gx_perm = torch.empty(B, D, dtype=U.dtype, device=U.device)
    a = _lib.PfFlowBwdChainArgs()
    a.batch = B
    keep = [HSk, T1k, GT1, GT2, Gp, Gh0, gx_perm, params]
    ops = [("U", U), ("params", params), ("hs", HSk), ("t1s", T1k), ("Gp", Gp), ("Gh0", Gh0),
           ("Gt1", GT1), ("Gt2", GT2), ("g_x", gx_perm)]
    if nll is not None:
        ops += [("g_nll", nll[0].contiguous()), ("nll_z", nll[1].contiguous())]
        if nll[2] is not None:
            ops.append(("log_sigma", nll[2].contiguous()))
    else:
        ops += [("g_z", g_z.contiguous()), ("g_lad", g_lad.contiguous())]
    keep += [t for _, t in ops[-3:]]
    if bf:      # the transposed matrices come from the PF_FLAG_BWD stream
        packed = flow.packed_weights(bwd=True)
        a.packed = packed.data_ptr()
        keep.append(packed)
    else:
        PM = (DM + 15) // 16 * 16
        WfT = F.pad(Wf.transpose(1, 2), (0, PM - DM)).contiguous()                   # [L, H, PM]
        W2T = torch.stack([w.transpose(1, 2) for w in W2]).contiguous()              # [nb, L, H(in), H(out)]
        W1T = torch.stack([w.transpose(1, 2) for w in W1]).contiguous()
        W0T = F.pad(W0.transpose(1, 2), (0, 0, 0, 16 - D)).contiguous()              # [L, 16, H]
        ops += [("WfT", WfT), ("W2T", W2T), ("W1T", W1T), ("W0T", W0T)]
        keep += [WfT, W2T, W1T, W0T]
    if drop is not None:        # gt1 = (W2^T gt2) . factor . [t1 > 0]: the chain reads the forward's factors
        ops.append(("drop", drop))
    a.gp_ld = DMp
    if cmp:
        a.compact, a.drop_scale = 1, (1.0 if drop is None else 1.0 / (1.0 - float(flow.dropout)))
    half = {"hs", "t1s", "Gp", "Gh0", "Gt1", "Gt2"} if cmp else set()
    for name, t in ops:
        if cmp and name == "drop":
            continue                                   # compact: the factor is drop_scale where t1s > 0
        assert t.is_contiguous() and t.dtype == (torch.bfloat16 if name in half else torch.float32), name
        setattr(a, name, t.data_ptr())
    if has_ctx:
        Gc = torch.empty(Ln, 1 + nb, B, H, dtype=gdt, device=U.device)
        keep += [T2k, Gk, pck, Gc]
        a.pc, a.Gc = pck.data_ptr(), Gc.data_ptr()
        if not additive:
            a.t2s, a.gates = T2k.data_ptr(), Gk.data_ptr()
    _lib.check(_lib.lib().pf_flow_backward_chain(flow._desc("bf16" if bf else "fp32"), a,
                                                 torch.cuda.current_stream(U.device).cuda_stream), "pf_flow_backward_chain")
    g_x = gx_perm[:, flow._ar_inv_perm]              # the kernel returns dL/d x[:, ar_perm]

    # 3. weight gradients: hand-written split-M GEMMs (pf_dense_tn, csrc/pf_dense.hip), batched over the layers, accumulating
    #    straight into ONE flat gradient buffer in the raw parameter layout (include/pf_hip.h) -- bias gradients come out of the
    #    same launches (column sums of the gradient operand) and the autoregressive masks are applied to the partial sums;
    #    9 launches replaced 18 library GEMMs, 6 reductions and the per-parameter bookkeeping.  The context gradient is pf_dense_nt over the 3 L slabs of Gc.
    lay = flow._raw_layout()
    P = lay["P"]
    dev = U.device
    prec = _lib.PF_PREC_BF16 if cmp else _lib.PF_PREC_F32
    adt = torch.bfloat16 if cmp else torch.float32
    lanes = 8 if cmp else 4
    up = lambda n: (n + lanes - 1) // lanes * lanes
    g_flat = torch.zeros(Ln, P, dtype=torch.float32, device=dev)
    L_ = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    keep2 = []

    mask = flow._raw_mask(dev)

    def tn(G, g_off, g_bs, ldg, A, a_bs, lda, n1, n2, w_off, b_off, ldw, n1_rows=0, n2_cols=0, masked=False):
        t = _lib.PfDenseTnArgs()
        t.G, t.g_seq_stride, t.ldg = G.data_ptr() + g_off * G.element_size(), 0, ldg
        t.A, t.a_seq_stride, t.lda = A.data_ptr(), 0, lda
        t.M, t.rows_per_seq, t.N1, t.N2 = B, B, n1, n2
        t.dW, t.ldw, t.db, t.splits = g_flat.data_ptr() + 4 * w_off, ldw, g_flat.data_ptr() + 4 * b_off, 0
        t.batch, t.g_batch_stride, t.a_batch_stride, t.w_batch_stride, t.b_batch_stride = Ln, g_bs, a_bs, P, P
        t.n1_rows, t.n2_cols = n1_rows, n2_cols
        t.mask = mask.data_ptr() + 4 * w_off if masked else None
        _lib.check(L_.pf_dense_tn(prec, t, stream), "pf_dense_tn")

    Dp = up(D)
    Upad = F.pad(U.to(adt), (0, Dp - D)) if (Dp != D or U.dtype != adt) else U
    Gpp = F.pad(Gp, (0, up(DMp) - DMp)) if up(DMp) != DMp else Gp
    DMp = up(DMp)
    keep2 += [Upad, Gpp]
    tn(Gh0, 0, B * H, H, Upad, B * Dp, Dp, H, Dp, lay["W0"], lay["b0"], D, n2_cols=D, masked=True)
    for j in range(nb):
        tn(GT1[j], 0, B * H, H, relu_h[j], B * H, H, H, H, lay["W1"][j], lay["b1"][j], H, masked=True)
        tn(GT2[j], 0, B * H, H, a1s[j], B * H, H, H, H, lay["W2"][j], lay["b2"][j], H, masked=True)
    tn(Gpp, 0, B * DMp, DMp, h_last, B * H, H, DMp, H, lay["Wf"], lay["bf"], H, n1_rows=DM, masked=True)
    g_ctx = None
    if has_ctx:
        Cp = up(C)
        ctxa = F.pad(ctx.to(adt), (0, Cp - C)) if (Cp != C or ctx.dtype != adt) else ctx
        keep2.append(ctxa)
        for j in range(1 + nb):
            tn(Gc, j * B * H, (1 + nb) * B * H, H, ctxa, 0, Cp, H, Cp, lay["Wc"][j], lay["bc"][j], C, n2_cols=C,
               masked=additive and not flow.full_context)
        wt = flow.packed_ctx_transposed("bf16" if cmp else "fp32")
        if wt is not None:
            # few rows, long reduction: B / 128 strips x ceil(tiles / 12) calls would leave most CUs idle (2048 rows: 32
            # workgroups, 307 us), so the 3 L slabs are divided over workgroups that add into a zeroed g_ctx
            nks_total = (1 + nb) * Ln * (H // (32 if cmp else 16))
            tiles = C // 16
            per_call = 12
            wgs = -(-B // 128) * -(-tiles // per_call)
            splits = max(1, min((1 + nb) * Ln, 256 // max(wgs, 1)))
            g_ctx = (torch.zeros if splits > 1 else torch.empty)(B, C, dtype=torch.float32, device=dev)
            for t0 in range(0, tiles, per_call):
                nt = min(per_call, tiles - t0)
                a2 = _lib.PfDenseArgs()
                a2.A, a2.M, a2.rows_per_seq, a2.a_seq_stride, a2.lda = Gc.data_ptr(), B, B, 0, H
                # a slab = one (layer, j) of Gc, H wide; its LDS image (128 rows x KC) must fit: H = 384 in fp32 goes in 3 chunks
                kc = H
                while 128 * kc * (2 if cmp else 4) > 96 * 1024 and kc % 128 == 0:
                    kc //= 2
                if 128 * kc * (2 if cmp else 4) > 96 * 1024 and H % 3 == 0 and (H // 3) % 64 == 0:
                    kc = H // 3
                a2.K, a2.N, a2.KC, a2.a_chunk_stride, a2.a_slab_chunks = (1 + nb) * Ln * H, 16 * nt, kc, B * H, H // kc
                a2.wfrags = wt.data_ptr() + t0 * nks_total * 64 * 16
                a2.out, a2.o_seq_stride, a2.ldo, a2.out_f32 = g_ctx.data_ptr() + 4 * 16 * t0, 0, C, 1
                a2.k_splits = splits
                _lib.check(L_.pf_dense_nt(prec, _lib.PF_EPI_PLAIN, a2, stream), "pf_dense_nt (context gradient)")
        else:       # a context width the packed form is not built for: one library GEMM
            flat = Gc.permute(2, 0, 1, 3).reshape(B, Ln * (1 + nb) * H)
            g_ctx = (flat @ wcat().reshape(-1, C).to(gdt)).float()
    return dict(g_x=g_x, g_ctx=g_ctx, flat=g_flat)


def _per_parameter(flow, g_flat):
    """the flat gradient [L, P] as one VIEW per parameter, in ``flow._ordered_parameters()`` order (no copies)."""
    shapes = flow._raw_layout()["shapes"]
    out = []
    for l in range(g_flat.shape[0]):
        row, off = g_flat[l], 0
        for shp, n in shapes:
            out.append(row[off:off + n].view(shp))
            off += n
    return out


def flow_backward(flow, U, ctx, g_z, g_lad, drop_seed=None, nll=None):
    """(dL/dx, dL/dctx or None, per-parameter gradients in ``flow._ordered_parameters()`` order); drop_seed: the seed
    the training forward applied dropout with, or None.
    (Replaying the chain from a captured hipGraph was tried: inside a training step the launches are
    already hidden behind queued encoder work, so it bought nothing and was dropped.)"""
    drop = None if drop_seed is None else dropout_mask(flow, U.shape[1], drop_seed, U.device)
    if nll is not None:
        g = _flow_backward_batched(flow, U, ctx, None, None, drop,
                                   (nll[0].float(), nll[1].float(), None if nll[2] is None else nll[2].float()))
    else:
        g = _flow_backward_batched(flow, U, ctx, g_z.contiguous().float(), g_lad.contiguous().float(), drop)
    if flow._theta is not None:               # flat-parameter mode: the gradient of the one leaf
        return g["g_x"], g["g_ctx"], [g["flat"].reshape(-1)]
    return g["g_x"], g["g_ctx"], _per_parameter(flow, g["flat"])


def _fast(flow) -> bool:
    """the HIP backward (re-evaluation + chain + split GEMMs) covers this flow; otherwise autograd over ``flow_forward``"""
    if not bool(getattr(flow, "use_masked_context", False)):
        return True
    # masked-context conditioner: the fp32 re-evaluation and chain kernels carry its additive form
    return (REEVAL_HIP and flow.features <= 16 and flow.hidden_features in (64, 128, 192, 256, 384, 512)
            and flow.num_bins <= 32)


def _layer_inputs(flow, x):
    return torch.empty(flow.num_layers, x.shape[0], flow.features, dtype=torch.float32, device=x.device)


class FlowNLL(torch.autograd.Function):
    """nll[B] = -(log N(z; 0, diag(e^ls)^2) + logdet); forward on the HIP kernel."""

    @staticmethod
    def forward(ctx_, flow, x, context, log_sigma, *params):
        U = _layer_inputs(flow, x) if _fast(flow) else None
        seed = flow._draw_dropout_seed() if flow._drop_active() else None
        with torch.no_grad():
            z, logdet, nll = flow._forward_call(x, context, log_sigma, want_z=True, guard=False, layer_inputs=U,
                                                dropout_seed=seed)
        ctx_.flow, ctx_.drop_seed = flow, seed
        ctx_.save_for_backward(x, context, log_sigma, U, z)
        ctx_.mark_non_differentiable(z, logdet)
        ctx_.set_materialize_grads(False)          # no zero tensors for the unused outputs' gradients
        return nll, z, logdet

    @staticmethod
    def backward(ctx_, g_nll, _gz, _gld):
        flow = ctx_.flow
        x, context, log_sigma, U, z = ctx_.saved_tensors
        params = flow._autograd_parameters()
        if g_nll is None:
            return (None,) * (4 + len(params))
        if U is not None:
            # nll = 0.5 sum (z e^-ls)^2 + sum ls + const - logdet
            # (dL/dz = g_nll z e^{-2 ls} and dL/dlogdet = -g_nll are formed inside the chain kernel)
            gl = None
            if log_sigma is not None and ctx_.needs_input_grad[3]:
                gl = g_nll[:, None] * (1.0 - (z * torch.exp(-log_sigma)).square())
            try:
                gx, gc, gp = flow_backward(flow, U, context, None, None, ctx_.drop_seed, nll=(g_nll, z, log_sigma))
                return (None, gx if ctx_.needs_input_grad[1] else None,
                        gc if (context is not None and ctx_.needs_input_grad[2]) else None, gl,
                        *[g if p.requires_grad else None for g, p in zip(gp, params)])
            except NotImplementedError:      # PF_ERR_UNSUPPORTED from a backward kernel (e.g. D = 16, K = 32 at H = 512: the chain's
                if not flow.use_masked_context:      # LDS image): the masked-context conditioner still has the replay below
                    raise
                _note_replay(flow, "PF_ERR_UNSUPPORTED from a backward kernel")
        else:
            _note_replay(flow, "shape outside the HIP backward's set")
        with torch.enable_grad():
            xs = x.detach().requires_grad_(x.requires_grad)
            cs = None if context is None else context.detach().requires_grad_(context.requires_grad)
            ls = None if log_sigma is None else log_sigma.detach().requires_grad_(log_sigma.requires_grad)
            drop = None if ctx_.drop_seed is None else dropout_mask(flow, x.shape[0], ctx_.drop_seed, x.device)
            z, logdet = flow_forward(flow, xs, cs, drop)
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
        U = _layer_inputs(flow, x) if _fast(flow) else None
        seed = flow._draw_dropout_seed() if flow._drop_active() else None
        with torch.no_grad():
            z, logdet, _ = flow._forward_call(x, context, None, want_z=True, guard=False, layer_inputs=U,
                                              dropout_seed=seed)
        ctx_.flow, ctx_.drop_seed = flow, seed
        ctx_.save_for_backward(x, context, U)
        return z, logdet

    @staticmethod
    def backward(ctx_, gz, gld):
        flow = ctx_.flow
        x, context, U = ctx_.saved_tensors
        params = flow._autograd_parameters()
        if U is not None:
            gz = torch.zeros_like(x) if gz is None else gz
            gld = torch.zeros(x.shape[0], device=x.device) if gld is None else gld
            try:
                gx, gc, gp = flow_backward(flow, U, context, gz, gld, ctx_.drop_seed)
                return (None, gx if ctx_.needs_input_grad[1] else None,
                        gc if (context is not None and ctx_.needs_input_grad[2]) else None,
                        *[g if p.requires_grad else None for g, p in zip(gp, params)])
            except NotImplementedError:      # (see FlowNLL.backward)
                if not flow.use_masked_context:
                    raise
                _note_replay(flow, "PF_ERR_UNSUPPORTED from a backward kernel")
        else:
            _note_replay(flow, "shape outside the HIP backward's set")
        with torch.enable_grad():
            xs = x.detach().requires_grad_(x.requires_grad)
            cs = None if context is None else context.detach().requires_grad_(context.requires_grad)
            drop = None if ctx_.drop_seed is None else dropout_mask(flow, x.shape[0], ctx_.drop_seed, x.device)
            z, logdet = flow_forward(flow, xs, cs, drop)
            wanted = [t for t in (xs, cs) if t is not None and t.requires_grad]
            wanted += [p for p in params if p.requires_grad]
            grads = torch.autograd.grad([z, logdet], wanted, grad_outputs=[gz, gld], allow_unused=True)
        it = iter(grads)
        gx = next(it) if xs.requires_grad else None
        gc = next(it) if cs is not None and cs.requires_grad else None
        gp = [next(it) if p.requires_grad else None for p in params]
        return (None, gx, gc, *gp)
