"""Conditioner dropout of the training forward (flow_train_kernel, pf_flow_forward_train_dropout / pf_flow_dropout_mask).

nflows' MaskedResidualBlock (and the reference's masked-context block, flows.py:225-234) applies nn.Dropout after the second
activation of every residual block while the module is in train mode; the flows are built with dropout_probability = dropout
(flows.py:522; create_flow_model defaults to 0.15, flows.py:1008).  nn.Dropout's random stream is not reproducible across
devices, so parity is stated on the FUNCTION: the HIP path draws its keep decisions from a counter hash of a per-call seed,
pf_flow_dropout_mask returns those factors, and with the oracle's nn.Dropout modules replaced by a multiplication with exactly
those factors
  * the training forward (z, log|det|, NLL, kept layer inputs) must equal the oracle's to the fp32 tolerances of
    tests/test_flow_forward_gpu.py, and the gradients those of tests/test_flow_backward_gpu.py;
  * the factors themselves are checked bit for bit against a numpy restatement of the hash, and statistically
    (keep rate 1 - p, factor 1 / (1 - p), independent across rows / layers / blocks / seeds).
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import flow_inputs, oracle_state_for_product

pytestmark = pytest.mark.gpu

M32 = np.uint64(0xFFFFFFFF)


def _mul(a, b):
    return (a * np.uint64(b)) & M32


def hash_factors(seed64, p, L, B, H, D):
    """numpy restatement of csrc/pf_flow_params.h drop_row_hash / drop_hash / drop_factor and of the position map
    (hidden units stable-sorted by MADE degree): [2, L, B, H] float32"""
    seed = np.uint64((seed64 ^ (seed64 >> 32)) & 0xFFFFFFFF)
    rows = np.arange(B, dtype=np.uint64)
    x = seed ^ _mul(rows, 0x9E3779B1)
    x ^= x >> np.uint64(16); x = _mul(x, 0x7FEB352D); x ^= x >> np.uint64(15); x = _mul(x, 0x846CA68B); x ^= x >> np.uint64(16)
    deg = np.arange(H) % max(1, D - 1) + min(1, D - 1)              # nflows _get_hidden_degrees (random_mask=False)
    perm = np.argsort(deg, kind="stable")                           # sorted position -> unit
    pos = np.empty(H, dtype=np.uint64)
    pos[perm] = np.arange(H, dtype=np.uint64)
    out = np.empty((2, L, B, H), dtype=np.float32)
    thresh = np.uint64(int(np.floor(float(np.float32(p)) * 16777216.0 + 0.5)))   # lround of the fp32 probability
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    for j in range(2):
        for l in range(L):
            lb = np.uint64(2 * l + j)
            y = (x[:, None] + _mul(lb * np.uint64(256) + pos[None, :] + np.uint64(1), 0x85EBCA77)) & M32
            y ^= y >> np.uint64(15); y = _mul(y, 0x2C1B3C6D); y ^= y >> np.uint64(12); y = _mul(y, 0x297A2D39); y ^= y >> np.uint64(15)
            out[j, l] = np.where((y >> np.uint64(8)) >= thresh, scale, np.float32(0.0))
    return out


class _Factors(nn.Module):
    """stands in for a block's nn.Dropout: multiplies by the factors the HIP forward applied"""

    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, t):
        return t * self.m.to(t.dtype)


def _install(ref, mask):
    """mask [2, L, B, H] (cpu) -> the oracle's blocks"""
    nets = [t.autoregressive_net for t in ref.transform._transforms if hasattr(t, "autoregressive_net")]
    for l, net in enumerate(nets):
        for j, blk in enumerate(net.blocks):
            blk.dropout = _Factors(mask[j, l])


def _pair(D, C, H, L, K, tb, p, masked=False, seed=0):
    from oracle.flow_ref import NSFPosteriorFlowRef
    from posteriflow_amd import NSFPosteriorFlow
    torch.manual_seed(seed)
    ref = NSFPosteriorFlowRef(D, C, H, L, K, tb, temperature_scale=1.0, use_masked_context=masked)
    # the second linear of a block is initialised at 1e-3 (zero_initialization): scale it up so that dropout matters
    with torch.no_grad():
        for t in ref.transform._transforms:
            if hasattr(t, "autoregressive_net"):
                for blk in t.autoregressive_net.blocks:
                    blk.linear_layers[1].weight.mul_(60.0)
    ref64 = NSFPosteriorFlowRef(D, C, H, L, K, tb, temperature_scale=1.0, use_masked_context=masked).double()
    ref64.load_state_dict(ref.state_dict())
    flow = NSFPosteriorFlow(D, C, H, L, K, tb, dropout=p, temperature_scale=1.0, use_masked_context=masked)
    assert flow.use_masked_context == masked
    flow.load_state_dict(oracle_state_for_product(ref))
    return ref, ref64, flow.cuda()


@pytest.mark.parametrize("D,H,L,B,p", [(15, 256, 3, 333, 0.15), (11, 256, 2, 64, 0.5), (4, 64, 2, 1000, 0.1), (1, 64, 1, 17, 0.3)])
def test_dropout_factors_match_the_restated_hash(D, H, L, B, p):
    from posteriflow_amd import NSFPosteriorFlow
    from posteriflow_amd._flow_autograd import dropout_mask
    flow = NSFPosteriorFlow(D, 0, H, L, 8, 3.0, dropout=p).cuda()
    for seed in (0, 1, (1 << 62) - 1, 0x1234_5678_9ABC_DEF0 >> 2):
        got = dropout_mask(flow, B, seed, torch.device("cuda")).cpu().numpy()
        want = hash_factors(seed, p, L, B, H, D)
        assert got.shape == want.shape and np.array_equal(got, want), (seed, np.abs(got - want).max())
    m = hash_factors(7, p, L, B, H, D)
    keep = (m > 0).mean()
    n = m.size
    assert abs(keep - (1 - p)) < 5 * np.sqrt(p * (1 - p) / n) + 1e-7, keep        # threshold is round(p 2^24) / 2^24
    assert np.all((m == 0) | (m == np.float32(1.0) / (np.float32(1.0) - np.float32(p))))
    if B * H >= 4096:
        # independent across blocks / layers / rows / seeds: agreement rate of two keep patterns = keep^2 + drop^2
        a, b = (m[0, 0] > 0), (m[1, 0] > 0)
        c = hash_factors(8, p, L, B, H, D)[0, 0] > 0
        r = np.roll(a, 1, axis=0)
        exp = (1 - p) ** 2 + p ** 2
        tol = 5 * np.sqrt(exp * (1 - exp) / a.size)
        for other in (b, c, r):
            assert abs((a == other).mean() - exp) < tol


@pytest.mark.parametrize("D,C,H,L,K,p,order", [(11, 288, 256, 3, 16, 0.15, None), (15, 288, 256, 2, 16, 0.3, "perm"),
                                               (4, 0, 64, 3, 8, 0.5, None), (6, 40, 128, 2, 10, 0.15, None)])
def test_training_forward_and_gradients_with_dropout_match_the_oracle(D, C, H, L, K, p, order):
    from posteriflow_amd._flow_autograd import dropout_mask
    tb, B = 5.0, 96
    ref, ref64, flow = _pair(D, C, H, L, K, tb, p)
    if order:
        perm = torch.randperm(D, generator=torch.Generator().manual_seed(1)).tolist()
        for f in (ref, ref64, flow):
            f.set_autoregressive_order(perm)
    x, ctx = flow_inputs(B, D, C, tb)
    g = torch.Generator().manual_seed(4)
    w, ls = torch.rand(B, generator=g) + 0.5, torch.randn(B, D, generator=g) * 0.2
    flow.train()
    flow.precision = "fp32"
    # the seed the differentiable call will draw
    torch.manual_seed(123)
    seed = flow._draw_dropout_seed()
    mask = dropout_mask(flow, B, seed, torch.device("cuda")).cpu()
    assert 0.0 < (mask == 0).float().mean() < 1.0
    _install(ref, mask), _install(ref64, mask)

    # forward: value parity, with and without the factors (they must matter)
    with torch.no_grad():
        z64, ld64 = ref64(x.double(), None if ctx is None else ctx.double())
        U = torch.empty(L, B, D, device="cuda")
        z, ld, nll = flow._forward_call(x.cuda(), None if ctx is None else ctx.cuda(), ls.cuda(), layer_inputs=U, dropout_seed=seed)
        n64 = ref64.compute_psd_aware_nll(x.double(), None if ctx is None else ctx.double(), ls.double())
        flow.eval()
        z_eval, _, _ = flow._forward_call(x.cuda(), None if ctx is None else ctx.cuda(), ls.cuda())
        flow.train()
    ez, eld = (z.cpu().double() - z64).abs().max().item(), (ld.cpu().double() - ld64).abs().max().item()
    moved = (z_eval - z).abs().max().item()
    print(f"\n[D{D} C{C} H{H} L{L} p{p}] |z - z64| {ez:.2e}  |ld - ld64| {eld:.2e}  dropout moves z by {moved:.2e}")
    assert ez < 2e-4 and eld < 2e-3
    assert moved > 100 * ez
    rel = ((nll.cpu().double() - n64).abs() / n64.abs().clamp_min(1.0)).max().item()
    assert rel < 1e-4, rel

    # gradients: the differentiable call draws the same seed
    xr = x.clone().requires_grad_(True)
    cr = ctx.clone().requires_grad_(True) if C else None
    lr = ls.clone().requires_grad_(True)
    (ref.compute_psd_aware_nll(xr, cr, lr) * w).sum().backward()
    xg = x.cuda().requires_grad_(True)
    cg = ctx.cuda().requires_grad_(True) if C else None
    lg = ls.cuda().requires_grad_(True)
    torch.manual_seed(123)
    out = flow.compute_psd_aware_nll(xg, cg, lg)
    assert torch.equal(out.detach(), nll)                       # same seed, same kernel: bit for bit
    (out * w.cuda()).sum().backward()
    relg = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
    assert relg(xg.grad.cpu(), xr.grad) < 2e-4 and relg(lg.grad.cpu(), lr.grad) < 2e-5
    if C:
        assert relg(cg.grad.cpu(), cr.grad) < 2e-4
    ref_params = dict(ref.named_parameters())
    n_checked = 0
    for name, prm in flow.named_parameters():
        if name.startswith("transform.") and prm.grad is not None:
            assert relg(prm.grad.cpu(), ref_params[name].grad) < 3e-4, name
            n_checked += 1
    assert n_checked == L * (18 if C else 12)

    # forward() is differentiable too
    for m in (ref, flow):
        m.zero_grad()
    xr.grad = None
    xg.grad = None
    zr, ldr = ref(xr, cr)
    (zr.square().sum() + (ldr * w).sum()).backward()
    torch.manual_seed(123)
    zg, ldg = flow(xg, cg)
    (zg.square().sum() + (ldg * w.cuda()).sum()).backward()
    assert relg(xg.grad.cpu(), xr.grad) < 2e-4
    for name, prm in flow.named_parameters():
        if name.startswith("transform.") and prm.grad is not None:
            assert relg(prm.grad.cpu(), ref_params[name].grad) < 3e-4, name


def test_dropout_modes_seeds_and_precisions():
    D, C, H, L, K, tb, B, p = 11, 288, 256, 3, 16, 5.0, 256, 0.15
    ref, _, flow = _pair(D, C, H, L, K, tb, p)
    x, ctx = flow_inputs(B, D, C, tb)
    xg, cg = x.cuda(), ctx.cuda()
    with torch.no_grad():
        flow.eval()                                             # eval: no dropout, whatever p
        ze, lde = flow(xg, cg)
        zr, ldr = ref(x, ctx)
        assert (ze.cpu() - zr).abs().max() < 2e-4 and (lde.cpu() - ldr).abs().max() < 2e-3
        flow.train()
        torch.manual_seed(1)
        z1, ld1 = flow(xg, cg)                                  # a no-grad call in train mode drops too (nn.Dropout does)
        torch.manual_seed(1)
        z1b, _ = flow(xg, cg)
        z2, _ = flow(xg, cg)                                    # the generator moved on: another mask
        assert torch.equal(z1, z1b) and not torch.equal(z1, z2) and not torch.equal(z1, ze)
        # bf16 mode: the same factors on the bf16 kernel -- close to the fp32 kernel's result with the same seed
        flow.precision = "bf16"
        torch.manual_seed(1)
        zb, ldb = flow(xg, cg)
        flow.precision = "fp32"
        dz = (zb - z1).abs().max(dim=1).values
        assert dz.median() < 5e-3 and dz.max() < 0.5, (dz.median(), dz.max())
        # the large-batch evaluation kernel is never used for a dropout forward, and the serving calls refuse
        flow.precision = "bf16"
        flow.wide_min_batch = 1
        torch.manual_seed(1)
        zb2, _ = flow(xg, cg)
        assert torch.equal(zb2, zb)
        with pytest.raises(RuntimeError, match="eval"):
            flow.nll_into(xg, cg, torch.empty(B, device="cuda"))
        with pytest.raises(RuntimeError, match="eval"):
            flow.sample(4, cg[:1])
    # dropout = 0 in train mode is the plain forward, bit for bit
    from posteriflow_amd import NSFPosteriorFlow
    plain = NSFPosteriorFlow(D, C, H, L, K, tb, dropout=0.0, temperature_scale=1.0, use_masked_context=False).cuda()
    plain.load_state_dict(oracle_state_for_product(ref))
    with torch.no_grad():
        plain.train()
        zt, _ = plain(xg, cg)
        plain.eval()
        zv, _ = plain(xg, cg)
        assert torch.equal(zt, zv)
    # a training step's worth: the loss of a dropout flow is finite and its gradients populate every parameter
    flow.precision = "fp32"
    flow.train()
    flow.zero_grad()
    flow.compute_psd_aware_nll(xg, cg, None).mean().backward()
    for name, prm in flow.named_parameters():
        if name.startswith("transform."):
            assert prm.grad is not None and torch.isfinite(prm.grad).all(), name


def test_dropout_backward_in_bf16_mode_follows_the_fp32_mode():
    """precision = "bf16": the training forward (bf16 flow_train_kernel), the HIP re-evaluation and the bf16 chain all take
    the same factors.  On the same seed the bf16-mode gradients must be close to the fp32-mode ones: cosine > 0.98 over all
    parameters and for x / context (the bf16 forward's own distance from fp32, tests/test_flow_backward_gpu.py), and the
    chain alone, on identical layer inputs / activations / factors, within 5e-2 / cosine 0.995 of the fp32 chain; the
    re-evaluation kernel's second linears against same-rounding tensor ops with the factors applied (2e-3)."""
    from posteriflow_amd import _flow_autograd as fa
    D, C, H, L, K, tb, B, p = 11, 288, 256, 3, 16, 5.0, 160, 0.15
    _, _, flow = _pair(D, C, H, L, K, tb, p)
    x, ctx = flow_inputs(B, D, C, tb)
    flow.train()
    cos = lambda a, b: torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()
    grads = {}
    for prec in ("fp32", "bf16"):
        flow.precision = prec
        flow.zero_grad(set_to_none=True)
        xg, cg = x.cuda().requires_grad_(True), ctx.cuda().requires_grad_(True)
        torch.manual_seed(77)
        flow.compute_psd_aware_nll(xg, cg, None).mean().backward()
        grads[prec] = (xg.grad.clone(), cg.grad.clone(),
                       torch.cat([q.grad.flatten() for n, q in flow.named_parameters() if n.startswith("transform.")]))
        assert all(torch.isfinite(t).all() for t in grads[prec])
    cs = [cos(a, b) for a, b in zip(grads["fp32"], grads["bf16"])]
    print(f"\n[dropout, bf16 vs fp32 mode] cosine x {cs[0]:.5f} context {cs[1]:.5f} parameters {cs[2]:.5f}")
    assert min(cs) > 0.98
    # the backward alone: same layer inputs, same factors, same incoming gradients
    flow.precision = "fp32"
    U = torch.empty(L, B, D, device="cuda")
    with torch.no_grad():
        flow._forward_call(x.cuda(), ctx.cuda(), None, layer_inputs=U, dropout_seed=5)
        drop = fa.dropout_mask(flow, B, 5, torch.device("cuda"))
        g = torch.Generator().manual_seed(3)
        gz, gl = torch.randn(B, D, generator=g).cuda(), torch.randn(B, generator=g).cuda()
        out = {}
        fa.REEVAL_HIP = False                  # the chain's handling of the factors, on identical activations
        try:
            for prec in ("fp32", "bf16"):
                flow.precision = prec
                out[prec] = fa._flow_backward_batched(flow, U, ctx.cuda(), gz, gl, drop)
        finally:
            fa.REEVAL_HIP = True
        # the re-evaluation kernel's handling: the second linear of each block sees relu(t1) . factor (same-rounding tensor ops)
        import torch.nn.functional as F
        rb = lambda t: t.bfloat16().float()
        HS, T1, T2, G, PC, H2, params = fa._reevaluate_hip(flow, U, ctx.cuda(), drop)
        for l, layer in enumerate(flow._ar_transforms):
            for j, blk in enumerate(layer.autoregressive_net.blocks):
                lin2 = blk.linear_layers[1]
                want = F.linear(rb(F.relu(T1[j, l]) * drop[j, l]), rb(lin2.weight * lin2.mask), lin2.bias)
                err = (T2[j, l] - want).abs().max() / want.abs().max()
                assert err < 2e-3, (l, j, err.item())
                without = F.linear(rb(F.relu(T1[j, l])), rb(lin2.weight * lin2.mask), lin2.bias)
                assert (T2[j, l] - without).abs().max() / want.abs().max() > 0.05      # the factors matter
    for k, v in out["fp32"].items():
        if v is None:
            continue
        for a, b in zip(v if isinstance(v, list) else [v], out["bf16"][k] if isinstance(v, list) else [out["bf16"][k]]):
            r_ = ((a - b).abs().max() / a.abs().max().clamp_min(1e-12)).item()
            assert r_ < 5e-2 and cos(a, b) > 0.995, (k, r_, cos(a, b))


def test_masked_context_flow_trains_with_dropout():
    """the reference's own masked-context block drops at the same place (flows.py:232); its backward is the fp32 re-evaluation
    + chain kernels in their additive form, reading the forward's factors"""
    from posteriflow_amd._flow_autograd import dropout_mask
    D, C, H, L, K, tb, B, p = 11, 264, 256, 2, 16, 5.0, 64, 0.2
    ref, ref64, flow = _pair(D, C, H, L, K, tb, p, masked=True)
    x, ctx = flow_inputs(B, D, C, tb)
    flow.train()
    flow.precision = "fp32"
    torch.manual_seed(9)
    seed = flow._draw_dropout_seed()
    mask = dropout_mask(flow, B, seed, torch.device("cuda")).cpu()
    _install(ref, mask), _install(ref64, mask)
    with torch.no_grad():
        z64, ld64 = ref64(x.double(), ctx.double())
    xr, cr = x.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    ref.compute_psd_aware_nll(xr, cr, torch.zeros(B, D)).sum().backward()
    xg, cg = x.cuda().requires_grad_(True), ctx.cuda().requires_grad_(True)
    torch.manual_seed(9)
    z, ld = flow(xg, cg)
    assert (z.detach().cpu().double() - z64).abs().max() < 2e-4 and (ld.detach().cpu().double() - ld64).abs().max() < 2e-3
    torch.manual_seed(9)
    flow.compute_psd_aware_nll(xg, cg, None).sum().backward()
    relg = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
    assert relg(xg.grad.cpu(), xr.grad) < 5e-4 and relg(cg.grad.cpu(), cr.grad) < 5e-4
    ref_params = dict(ref.named_parameters())
    for name, prm in flow.named_parameters():
        if name.startswith("transform.") and prm.grad is not None:
            assert relg(prm.grad.cpu(), ref_params[name].grad) < 5e-4, name
