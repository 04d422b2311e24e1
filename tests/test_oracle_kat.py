"""Known-answer tests for the nflows restatement (SURVEY.md 8c KATs 1-6).
The reference holds no test or vector for the flow ("parity unpinned"), so these
self-consistency properties are what anchors oracle/nflows_restated.py."""
import math

import pytest
import torch

from oracle import nflows_restated as nfr
from oracle.flow_ref import NSFPosteriorFlowRef, scale_final_layers, flops_per_sample


def make_flow(D=5, C=7, H=64, L=3, K=8, tb=3.0, scale=3.0, seed=0, dtype=torch.float64):
    torch.manual_seed(seed)
    f = NSFPosteriorFlowRef(D, C, H, L, K, tb, temperature_scale=1.0)
    scale_final_layers(f, scale)
    return f.to(dtype)


def inputs(B, D, C, tb, dtype=torch.float64, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(B, D, generator=g, dtype=dtype) * 2 - 1) * (tb * 0.9)
    x[0, 0] = tb            # on the bound: last bin, theta = 1
    x[1, 1] = -tb
    x[2, :2] = torch.tensor([tb * 1.3, -tb * 2.0], dtype=dtype)   # outside: identity
    ctx = torch.randn(B, C, generator=g, dtype=dtype)
    return x, ctx


def test_kat1_roundtrip_and_logdet_sign():
    f = make_flow()
    x, ctx = inputs(48, 5, 7, 3.0)
    with torch.no_grad():
        z, ld = f(x, ctx)
        xr, ldi = f.inverse_raw(z, ctx)
    assert (xr - x).abs().max() < 1e-6          # SURVEY 8c KAT 1: <= 1e-5
    assert (ld + ldi).abs().max() < 1e-6


def test_kat1_tails_identity_single_layer():
    t = nfr.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(4, 32, None, 8, 2.0).double()
    x = torch.tensor([[2.5, -7.0, 0.3, 2.0]], dtype=torch.float64)
    with torch.no_grad():
        y, ld = t(x)
    assert y[0, 0] == 2.5 and y[0, 1] == -7.0
    assert abs(y[0, 3].item() - 2.0) < 1e-9      # knot at the bound maps onto itself


def test_kat2_logdet_equals_autograd_slogdet():
    f = make_flow()
    x, ctx = inputs(6, 5, 7, 3.0)
    _, ld = f(x, ctx)
    for i in (3, 4, 5):
        J = torch.autograd.functional.jacobian(lambda v: f(v[None], ctx[i:i + 1])[0][0], x[i])
        assert abs(torch.linalg.slogdet(J)[1].item() - ld[i].item()) < 1e-9


def test_kat3_autoregressive_structure():
    t = nfr.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(6, 48, 3, 8, 3.0).double()
    with torch.no_grad():
        t.autoregressive_net.final_layer.weight.mul_(10)
    x = torch.rand(6, dtype=torch.float64) * 2 - 1
    c = torch.randn(1, 3, dtype=torch.float64)
    J = torch.autograd.functional.jacobian(lambda v: t(v[None], c)[0][0], x)
    assert torch.all(torch.triu(J, diagonal=1) == 0)          # dz_i/dx_j = 0 for j > i
    Jp = torch.autograd.functional.jacobian(
        lambda v: t.autoregressive_net(v[None], c)[0].view(6, -1), x)      # [6, M, 6]
    for i in range(6):
        assert torch.all(Jp[i, :, i:] == 0)                   # params_i blind to x_j, j >= i


def test_kat4_zero_conditioner():
    K, tb = 8, 2.0
    x = torch.tensor([[-1.7, -0.2, 0.0, 0.6, 1.99]], dtype=torch.float64)
    z = torch.zeros(1, 5, K, dtype=torch.float64)
    y, ld = nfr.unconstrained_rational_quadratic_spline(x, z, z.clone(), z[..., : K - 1].clone(),
                                                        tail_bound=tb)
    # uniform bins, interior derivative 1e-3 + softplus(0); boundary derivative exactly 1
    d_in = 1e-3 + math.log(2.0)
    assert abs(1e-3 + math.log1p(math.exp(math.log(math.exp(1 - 1e-3) - 1))) - 1.0) < 1e-12
    w = 2 * tb / K
    k = torch.floor((x + tb) / w).long().clamp(max=K - 1)
    theta = (x + tb) / w - k
    dk = torch.where(k == 0, torch.ones_like(x), torch.full_like(x, d_in))
    dk1 = torch.where(k == K - 1, torch.ones_like(x), torch.full_like(x, d_in))
    tt = theta * (1 - theta)
    den = 1 + (dk + dk1 - 2) * tt
    y_ref = -tb + w * k + w * (theta ** 2 + dk * tt) / den
    ld_ref = torch.log(dk1 * theta ** 2 + 2 * tt + dk * (1 - theta) ** 2) - 2 * torch.log(den)
    assert (y - y_ref).abs().max() < 1e-12 and (ld - ld_ref).abs().max() < 1e-12


@pytest.mark.parametrize("D", [1, 2])
def test_kat5_density_integrates_to_one(D):
    f = make_flow(D=D, C=3, H=32, L=2, K=8, tb=3.0, scale=1.5)
    n = 40001 if D == 1 else 1201
    g = torch.linspace(-8, 8, n, dtype=torch.float64)
    pts = g[:, None] if D == 1 else torch.cartesian_prod(g, g)
    ctx = torch.randn(1, 3, dtype=torch.float64).expand(pts.shape[0], 3)
    with torch.no_grad():
        logp = -f.compute_psd_aware_nll(pts, ctx, torch.zeros_like(pts))
    integral = torch.exp(logp).sum().item() * ((g[1] - g[0]).item() ** D)
    assert abs(integral - 1.0) < 1e-4


def test_kat6_masks_and_param_count():
    t = nfr.MaskedPiecewiseRationalQuadraticAutoregressiveTransform(11, 256, 288, 16, 5.0)
    net = t.autoregressive_net
    assert abs(net.initial_layer.mask.mean().item() - 0.4957) < 1e-4
    assert abs(net.blocks[0].linear_layers[0].mask.mean().item() - 0.5500) < 1e-4
    assert abs(net.final_layer.mask.mean().item() - 0.5043) < 1e-4
    assert sum(p.numel() for p in t.parameters()) == 621061
    assert net.final_layer.weight.shape == (11 * 47, 256)
    # output layout [D, M]: row d*M + j has degree d + 1
    assert net.final_layer.degrees.view(11, 47)[:, 0].tolist() == list(range(1, 12))
    assert flops_per_sample(11, 288, 256, 16, 10) == 12369920
    assert flops_per_sample(15, 288, 256, 16, 8) == 10682368
    assert flops_per_sample(15, 288, 256, 16, 12) == 16023552


def test_wrapper_semantics():
    f = make_flow(D=4, C=6, H=32, L=2, K=8, tb=5.0, scale=1.0, dtype=torch.float32)
    assert f.tail_bound == 5.0
    assert NSFPosteriorFlowRef(4, 6, 32, 1, 8, tail_bound=5).tail_bound == 3.0   # flows.py:517 quirk
    x, ctx = inputs(16, 4, 6, 2.0, torch.float32)
    with torch.no_grad():
        z = torch.randn(16, 4) * 3
        xs, _ = f.inverse(z, ctx)
        assert xs.abs().max() <= 3.0                                         # flows.py:654 clamp
        bad = ctx.clone(); bad[0, 0] = float("nan"); bad[1, 1] = float("inf")
        xs2, _ = f.inverse(z, bad)
        assert torch.isfinite(xs2).all()
        nll = f.compute_psd_aware_nll(x, ctx, torch.zeros_like(x))
        assert torch.allclose(f.log_prob(x, ctx), nll, atol=1e-5)            # T = 1
        f.set_autoregressive_order([2, 0, 3, 1])
        z2, _ = f(x, ctx)
        xr, _ = f.inverse_raw(z2, ctx)
        assert (xr - x).abs().max() < 1e-3
    with pytest.raises(ValueError):
        f.set_autoregressive_order([0, 0, 1, 2])
