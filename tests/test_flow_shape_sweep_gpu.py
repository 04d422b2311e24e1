"""Seeded sweep over flow shapes the reference could be built with (features, context width, hidden width,
bins, layers, tail bound, batch, permuted order): forward / log-density, gradient-free inverse round trip and
the pack map, each against the CPU oracle, in fp32 mode.  Complements the hand-picked configurations of the
other GPU tests with shapes nobody picked."""
import random

import pytest
import torch

from helpers import flow_inputs, make_pair

pytestmark = pytest.mark.gpu


def _configs(n=12, seed=2024):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        H = rng.choice([64, 128, 192, 256])
        D = rng.randint(2, min(16, H // 16))
        C = rng.choice([0, 1, 7, 32, 33, 96, 288, 300])
        K = rng.choice([2, 5, 8, 13, 16])
        L = rng.randint(1, 5)
        tb = rng.choice([1.0, 3.0, 5.0])
        B = rng.choice([1, 15, 16, 17, 100, 257])
        out.append((D, C, H, K, L, tb, B, rng.random() < 0.5, 1000 + i))
    return out


@pytest.mark.parametrize("D,C,H,K,L,tb,B,permute,seed", _configs(),
                         ids=lambda v: str(v) if not isinstance(v, bool) else ("perm" if v else "id"))
def test_random_shape_matches_oracle(D, C, H, K, L, tb, B, permute, seed):
    ref, ref64, flow = make_pair(D, C, H, L, K, tb, seed=seed)
    if permute:
        order = list(range(D))
        random.Random(seed).shuffle(order)
        for m in (ref, ref64, flow):
            m.set_autoregressive_order(order)
    x, ctx = flow_inputs(B, D, C, tb, seed=seed + 1)
    with torch.no_grad():
        z64, ld64 = ref64(x.double(), None if ctx is None else ctx.double())
        z32, ld32 = ref(x, ctx)
        zg, ldg = flow(x.cuda(), None if ctx is None else ctx.cuda())
    # fp32 kernel vs fp64 oracle, bounded by a small multiple of the CPU fp32 path's own error
    for got, w32, w64 in ((zg.cpu(), z32, z64), (ldg.cpu(), ld32, ld64)):
        err = (got.double() - w64).abs()
        cpu = (w32.double() - w64).abs()
        scale = w64.abs().clamp_min(1.0)
        assert (err / scale).max() < max(2e-5, 6 * (cpu / scale).max().item()), ((err / scale).max(), (cpu / scale).max())
    # inverse round trip through the kernels
    with torch.no_grad():
        xb, ldi = flow.flow_inverse_raw(zg, None if ctx is None else ctx.cuda()) if hasattr(flow, "flow_inverse_raw") \
            else flow.inverse(zg, None if ctx is None else ctx.cuda())
    inside = (x.abs() < 3.0).all(dim=1)                      # inverse() clamps to +-3 like the reference
    if inside.any():
        back = xb.cpu()[inside]
        assert (back - x[inside]).abs().max() < 5e-3, (back - x[inside]).abs().max()


@pytest.mark.parametrize("D,C,H,K,L,tb,B,permute,seed", _configs(8, seed=77),
                         ids=lambda v: str(v) if not isinstance(v, bool) else ("perm" if v else "id"))
def test_random_shape_bf16_and_gradients(D, C, H, K, L, tb, B, permute, seed):
    """bf16 mode stays close to the fp32 oracle (bf16-sized tolerance: catches layout bugs, not rounding), and
    the gradients of the fast backward path match autograd through the oracle on the same shape."""
    ref, _, flow = make_pair(D, C, H, L, K, tb, seed=seed)
    if permute:
        order = list(range(D))
        random.Random(seed).shuffle(order)
        ref.set_autoregressive_order(order), flow.set_autoregressive_order(order)
    B = max(B, 8)
    x, ctx = flow_inputs(B, D, C, tb, seed=seed + 1)
    xr = x.clone().requires_grad_(True)
    cr = None if ctx is None else ctx.clone().requires_grad_(True)
    nll_ref = ref.compute_psd_aware_nll(xr, cr, torch.zeros_like(x))
    nll_ref.sum().backward()
    xg = x.cuda().requires_grad_(True)
    cg = None if ctx is None else ctx.cuda().requires_grad_(True)
    nll = flow.compute_psd_aware_nll(xg, cg, None)
    nll.sum().backward()
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-9)).item()
    assert rel(nll.detach().cpu(), nll_ref.detach()) < 1e-4
    assert rel(xg.grad.cpu(), xr.grad) < 5e-5                 # measured <= 2.4e-5 over the eight shapes
    if ctx is not None:
        assert rel(cg.grad.cpu(), cr.grad) < 5e-5             # measured <= 2.0e-5
    ref_params = dict(ref.named_parameters())
    for name, p in flow.named_parameters():
        if name.startswith("transform.") and p.grad is not None and ref_params[name].grad.abs().max() > 1e-6:
            assert rel(p.grad.cpu(), ref_params[name].grad) < 2e-3, name
    flow.precision = "bf16"
    with torch.no_grad():
        got = flow.compute_psd_aware_nll(x.cuda(), None if ctx is None else ctx.cuda(), None).cpu()
    assert torch.isfinite(got).all()
    err = (got - nll_ref.detach()).abs() / nll_ref.detach().abs().clamp_min(1.0)
    assert err.median() < 4e-3 and err.max() < 1.5e-2, (err.median(), err.max())      # measured <= 1.7e-3 / 6.3e-3


@pytest.mark.parametrize("D,block,H,K,L", [(4, 5, 64, 8, 2), (7, 16, 128, 16, 3), (12, 3, 192, 13, 2), (15, 20, 256, 16, 2)])
def test_masked_context_variant_shapes(D, block, H, K, L):
    """the reference's masked-context conditioner (flows.py:112-360, auto-on when C % D == 0) on other shapes
    than the one of test_flow_forward_gpu.py: forward / log-det against the fp64 oracle, with a permuted order."""
    from helpers import oracle_state_for_product
    from oracle.flow_ref import NSFPosteriorFlowRef
    from posteriflow_amd import NSFPosteriorFlow
    C, tb = D * block, 3.0
    torch.manual_seed(D)
    ref = NSFPosteriorFlowRef(D, C, H, L, K, tb, temperature_scale=1.0, use_masked_context=True)
    ref64 = NSFPosteriorFlowRef(D, C, H, L, K, tb, temperature_scale=1.0, use_masked_context=True).double()
    ref64.load_state_dict(ref.state_dict())
    flow = NSFPosteriorFlow(D, C, H, L, K, tb, temperature_scale=1.0)
    assert flow.use_masked_context
    flow.load_state_dict(oracle_state_for_product(ref))
    flow = flow.cuda()
    order = list(range(D))
    random.Random(D).shuffle(order)
    for f in (ref, ref64, flow):
        f.set_autoregressive_order(order)
    x, ctx = flow_inputs(90, D, C, tb, seed=D)
    with torch.no_grad():
        z64, ld64 = ref64(x.double(), ctx.double())
        z32, ld32 = ref(x, ctx)
        z, ld = flow(x.cuda(), ctx.cuda())
    assert (z.cpu().double() - z64).abs().max() < max(6 * (z32.double() - z64).abs().max().item(), 3e-5)
    assert (ld.cpu().double() - ld64).abs().max() < max(6 * (ld32.double() - ld64).abs().max().item(), 1e-4)


# ---- shapes none of the scheduled kernels is built for: the generic kernel (csrc/pf_flow_generic.hip) -----------------------
GENERIC = [  # D, C, H, K, L, tail, B, permute, seed
    (11, 288, 384, 24, 12, 3.0, 100, False, 4001),     # FlowHead(12, 384, 24), experiments/frozen_context_heads.py:159-163
    (17, 33, 256, 16, 2, 5.0, 37, True, 4002),         # D > H / 16
    (6, 0, 48, 3, 3, 1.0, 16, False, 4003),            # H not a scheduled width, context-free
    (8, 96, 256, 20, 3, 3.0, 257, True, 4004),         # K > 16
    (1, 7, 64, 5, 2, 3.0, 15, False, 4005),            # would be scheduled ... D = 1 is; kept as a control of the same test
    (32, 64, 512, 6, 2, 5.0, 33, False, 4006),         # the widest: H = 512, D = 32
]


@pytest.mark.parametrize("D,C,H,K,L,tb,B,permute,seed", GENERIC,
                         ids=lambda v: str(v) if not isinstance(v, bool) else ("perm" if v else "id"))
def test_generic_shape_matches_oracle(D, C, H, K, L, tb, B, permute, seed):
    """forward (z, log-det, nll) in fp32 against the fp64 oracle at the CPU fp32 path's own distance, bf16 against the
    same-rounding oracle; the D-pass inverse against the oracle's inverse and as a round trip; gradients are refused."""
    ref, ref64, flow = make_pair(D, C, H, L, K, tb, seed=seed)
    if permute:
        order = list(range(D))
        random.Random(seed).shuffle(order)
        for m in (ref, ref64, flow):
            m.set_autoregressive_order(order)
    x, ctx = flow_inputs(B, D, C, tb, seed=seed + 1)
    cg = None if ctx is None else ctx.cuda()
    with torch.no_grad():
        z64, ld64 = ref64(x.double(), None if ctx is None else ctx.double())
        z32, ld32 = ref(x, ctx)
        zg, ldg = flow(x.cuda(), cg)
        nll = flow.compute_psd_aware_nll(x.cuda(), cg, None)
        want_nll = ref64.compute_psd_aware_nll(x.double(), None if ctx is None else ctx.double(), torch.zeros_like(x).double())
    for got, w32, w64 in ((zg.cpu(), z32, z64), (ldg.cpu(), ld32, ld64), (nll.cpu(), -(-ld32), want_nll)):
        err = (got.double() - w64).abs()
        scale = w64.abs().clamp_min(1.0)
        if got is nll.cpu():
            assert (err / scale).max() < 1e-4, (err / scale).max()
            continue
        cpu = (w32.double() - w64).abs()
        assert (err / scale).max() < max(2e-5, 6 * (cpu / scale).max().item()), ((err / scale).max(), (cpu / scale).max())
    # inverse: against the oracle's D-pass inverse of the SAME z, and the round trip
    with torch.no_grad():
        xb, ldi = flow.inverse(zg, cg)
        xo, ldo = ref64.inverse(zg.cpu().double(), None if ctx is None else ctx.double())
    assert (xb.cpu().double() - xo).abs().max() < 5e-4, (xb.cpu().double() - xo).abs().max()
    assert ((ldi.cpu().double() - ldo).abs() / ldo.abs().clamp_min(1.0)).max() < 1e-3
    inside = (x.abs() < 3.0).all(dim=1)
    if inside.any():
        assert (xb.cpu()[inside] - x[inside]).abs().max() < 5e-3
    # grouped context (one context row per group of draws) == the expanded form
    if ctx is not None and B % 2 == 0:
        half = ctx[: B // 2].cuda()
        zz = torch.randn(B, D, device="cuda")
        with torch.no_grad():
            a, _ = flow.inverse(zz, half.repeat_interleave(2, dim=0))
            b, _ = flow.inverse(zz, half)
        assert torch.equal(a, b)
    # bf16 mode: the kernel against the oracle evaluated with the same operand rounding
    from oracle import nflows_restated
    flow.precision = "bf16"
    with torch.no_grad():
        zb, lb = flow(x.cuda(), cg)
        with nflows_restated.gemm_emulation("bf16"):
            ze, le = ref(x, ctx)
    dz = (zb.cpu() - ze).abs()
    dl = (lb.cpu() - le).abs() / le.abs().clamp_min(1.0)
    assert dz.median() < 5e-3 and dl.median() < 1e-2, (dz.median(), dl.median())
    flow.precision = "fp32"
    if flow._generic_shape() and not flow._generic_trainable():       # D > 16 or a width the backward chain is not built for
        with pytest.raises(NotImplementedError):
            flow(x.cuda().requires_grad_(True), cg)


@pytest.mark.parametrize("D,C,H,K,L,tb,precision", [(11, 288, 384, 24, 3, 3.0, "fp32"), (8, 96, 256, 20, 2, 3.0, "fp32"),
                                                    (6, 0, 512, 6, 2, 5.0, "fp32"), (11, 288, 384, 24, 2, 3.0, "bf16")])
def test_generic_shape_gradients_match_oracle_autograd(D, C, H, K, L, tb, precision):
    """Training of the shapes the generic forward kernel serves (H = 384 / 512, K <= 32, D <= 16): generic forward with the
    layer inputs kept -> fp32 re-evaluation by the generic kernel's conditioner -> fp32 chain (32-bin spline backward) ->
    transposed GEMMs.  Gradients of x, context, log sigma and every parameter tensor against autograd through the oracle."""
    ref, _, flow = make_pair(D, C, H, L, K, tb, seed=5000 + H + K)
    flow.precision = precision
    B = 80
    x, ctx = flow_inputs(B, D, C, tb, seed=77)
    g = torch.Generator().manual_seed(4)
    w, ls = torch.rand(B, generator=g) + 0.5, torch.randn(B, D, generator=g) * 0.2
    xr = x.clone().requires_grad_(True)
    cr = ctx.clone().requires_grad_(True) if C else None
    lr = ls.clone().requires_grad_(True)
    (ref.compute_psd_aware_nll(xr, cr, lr) * w).sum().backward()
    xg = x.cuda().requires_grad_(True)
    cg = ctx.cuda().requires_grad_(True) if C else None
    lg = ls.cuda().requires_grad_(True)
    (flow.compute_psd_aware_nll(xg, cg, lg) * w.cuda()).sum().backward()
    if precision == "fp32":
        rel = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
        tol = 3e-4
        assert rel(xg.grad.cpu(), xr.grad) < tol and rel(lg.grad.cpu(), lr.grad) < 1e-4
        if C:
            assert rel(cg.grad.cpu(), cr.grad) < tol
    else:       # bf16 forward (generic kernel, bf16 operands) + fp32 backward on its trajectory: direction, not digits
        rel = lambda a, b: 1.0 - torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()
        tol = 5e-2
        assert rel(xg.grad.cpu(), xr.grad) < tol
    ref_params = dict(ref.named_parameters())
    n_checked = 0
    for name, p in flow.named_parameters():
        if name.startswith("transform.") and p.grad is not None:
            assert rel(p.grad.cpu(), ref_params[name].grad) < tol, (name, rel(p.grad.cpu(), ref_params[name].grad))
            n_checked += 1
    assert n_checked == L * (18 if C else 12)


@pytest.mark.parametrize("D,block,H,K,L,full,prec", [(4, 5, 64, 8, 2, True, "fp32"), (12, 3, 192, 13, 2, True, "fp32"),
                                                    (11, 24, 256, 16, 3, True, "bf16"), (15, 20, 256, 16, 2, False, "fp32"),
                                                    (7, 16, 128, 16, 3, False, "bf16"), (6, 8, 384, 24, 2, True, "fp32")])
def test_masked_context_gradients_on_the_hip_backward(D, block, H, K, L, full, prec):
    """backward of the reference's masked-context conditioner (flows.py:186-234: additive projections inside the blocks, no gates,
    no ReversePermutation) on the fp32 re-evaluation + chain kernels: every gradient against the fp32 oracle's autograd, and the
    kernels are the ones that ran (no tensor-op forward replay)."""
    from helpers import oracle_state_for_product
    from oracle.flow_ref import NSFPosteriorFlowRef
    from posteriflow_amd import NSFPosteriorFlow, _flow_autograd as fa
    C, tb, B = D * block, 4.0, 150
    torch.manual_seed(D + H)
    ref = NSFPosteriorFlowRef(D, C, H, L, K, tb, temperature_scale=1.0, use_masked_context=True, full_context=full)
    with torch.no_grad():
        for t in ref.transform._transforms:
            if hasattr(t, "autoregressive_net"):
                for blk in t.autoregressive_net.blocks:
                    blk.linear_layers[1].weight.mul_(40.0 if prec == "fp32" else 8.0)
    flow = NSFPosteriorFlow(D, C, H, L, K, tb, temperature_scale=1.0, full_context=full)
    flow.load_state_dict(oracle_state_for_product(ref))
    flow = flow.cuda()
    flow.precision = prec
    assert flow.use_masked_context and fa._fast(flow)
    order = list(range(D))
    random.Random(D).shuffle(order)
    for f in (ref, flow):
        f.set_autoregressive_order(order)
    x, ctx = flow_inputs(B, D, C, tb, seed=D)
    g = torch.Generator().manual_seed(4)
    w, ls = torch.rand(B, generator=g) + 0.5, torch.randn(B, D, generator=g) * 0.2
    xr, cr, lr = x.clone().requires_grad_(True), ctx.clone().requires_grad_(True), ls.clone().requires_grad_(True)
    (ref.compute_psd_aware_nll(xr, cr, lr) * w).sum().backward()
    xg, cg, lg = (t.cuda().requires_grad_(True) for t in (x, ctx, ls))
    calls = []
    orig = fa.flow_forward
    fa.flow_forward = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        (flow.compute_psd_aware_nll(xg, cg, lg) * w.cuda()).sum().backward()
    finally:
        fa.flow_forward = orig
    assert not calls                                            # the tensor-op replay did not run
    if prec == "fp32":
        tol = 5e-4
        relg = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
    else:
        # bf16: z and the layer inputs come from the bf16 forward kernel, the gradient is the fp32 one AT those inputs.  A pair
        # that the bf16 conditioner puts in the neighbouring bin has a different log-det gradient (the spline is C1, the
        # derivative of its log-slope jumps at a knot), so a few rows differ by O(1) while the rest agree to bf16 accuracy:
        # (parameter gradients sum over the rows, so every element carries some of it): error relative to the gradient's norm
        tol = 0.25
        relg = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    assert relg(xg.grad.cpu(), xr.grad) < tol and relg(cg.grad.cpu(), cr.grad) < tol and relg(lg.grad.cpu(), lr.grad) < tol
    ref_params = dict(ref.named_parameters())
    n = 0
    for name, prm in flow.named_parameters():
        if name.startswith("transform."):
            assert prm.grad is not None, name
            assert relg(prm.grad.cpu(), ref_params[name].grad) < tol, name
            if "context_layer.weight" in name and not full:      # the per-position context mask holds in the gradient
                mod = flow.get_submodule(name.rsplit(".", 1)[0])
                assert (prm.grad * (1 - mod.mask)).abs().max() == 0
            n += 1
    assert n > 0

