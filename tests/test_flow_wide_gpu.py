"""GPU parity of the two kernels over the PF_FLAG_WIDE layout: the large-batch ("wide") forward kernel
(csrc/pf_flow_wide_kernel.h: 128 rows per workgroup, every wave owns 32 rows through all layers, weights fetched once per
workgroup through an LDS ring) and the mid-batch kernel (csrc/pf_flow_mid_kernel.h, round 4: 64 rows per workgroup, 8 waves =
two per SIMD, a wave owns one hidden tile for both row blocks; what pf_flow_forward dispatches up to 16 384 rows).  Every
test runs for both ($PF_FLOW_MID forces one or the other at any batch size).

It computes the same function in the same arithmetic as the bf16 mode of the 16-row kernel (bf16 operands, x as a hi + lo
pair, fp32 accumulate / residual / spline) but sums in a different order, so it is checked
  * against the oracle evaluated with the same operand rounding (oracle.nflows_restated.gemm_emulation("bf16")):
    tolerances of tests/test_flow_forward_gpu.py::test_forward_bf16_tolerance (median 5e-4 on z, 5e-3 on log|det|;
    worst row 0.1 / 0.5: one bf16 rounding-boundary flip amplified by the later layers);
  * against the 16-row kernel on the same inputs: the two must be as close to each other as each is to the emulation;
  * for ragged batches, a permuted autoregressive order, a PSD log-scale, the in-kernel loss reduction and the
    training forward's layer inputs.
"""
import os

import pytest
import torch

from helpers import flow_inputs, make_pair

pytestmark = pytest.mark.gpu

KERNELS = {"wide": ("0", "pf::flow_wide_kernel<%d, 18>", 128), "mid": ("1", "pf::flow_mid_kernel<%d, 18>", 64)}


@pytest.fixture(params=["wide", "mid"])
def kernel(request):
    """forces one of the two PF_FLAG_WIDE kernels for the duration of a test; yields (name template, rows per workgroup)"""
    env, name, rows = KERNELS[request.param]
    old = os.environ.get("PF_FLOW_MID")
    os.environ["PF_FLOW_MID"] = env
    yield name, rows
    if old is None:
        os.environ.pop("PF_FLOW_MID", None)
    else:
        os.environ["PF_FLOW_MID"] = old


def _pair(D, L, scale=1.0):
    ref, ref64, flow = make_pair(D, 288, 256, L, 16, 5.0, scale=scale)
    flow.precision = "bf16"
    return ref, ref64, flow


def _both(flow, fn):
    """fn() through the 16-row kernel and through the wide kernel"""
    flow.wide_min_batch = 1 << 40
    a = fn()
    flow.wide_min_batch = 1
    b = fn()
    return a, b


@pytest.mark.parametrize("D,L,B,scale", [(15, 8, 1024, 1.0), (11, 10, 300, 1.0), (15, 2, 4096 + 17, 2.0), (15, 8, 128, 1.0)])
def test_wide_forward_matches_the_oracle_and_the_16_row_kernel(D, L, B, scale, kernel):
    from oracle import nflows_restated as nfr
    from posteriflow_amd import _lib
    ref, ref64, flow = _pair(D, L, scale)
    x, ctx = flow_inputs(B, D, 288, 5.0)
    with torch.no_grad():
        with nfr.gemm_emulation("bf16"):
            zemu, ldemu = ref(x, ctx)
        z64, ld64 = ref64(x.double(), ctx.double())
        (za, lda), (zw, ldw) = _both(flow, lambda: flow(x.cuda(), ctx.cuda()))
        assert flow.forward_kernel_name(B) == kernel[0] % D
        assert _lib.lib().pf_flow_rows_per_workgroup(flow._desc(wide=True), B) == kernel[1]
    q = lambda t: "med %.1e p99 %.1e max %.1e" % tuple(t.quantile(torch.tensor([0.5, 0.99, 1.0], dtype=t.dtype)).tolist())
    ez_w, el_w = (zw.cpu() - zemu).abs().max(dim=1).values, (ldw.cpu() - ldemu).abs()
    ez_a, el_a = (za.cpu() - zemu).abs().max(dim=1).values, (lda.cpu() - ldemu).abs()
    ez_p, el_p = (zw - za).abs().max(dim=1).values.cpu(), (ldw - lda).abs().cpu()
    print(f"\n[D{D} L{L} B{B} x{scale:g}] wide vs emulation: |z| {q(ez_w)}  |ld| {q(el_w)}\n"
          f"      16-row vs emulation: |z| {q(ez_a)}  |ld| {q(el_a)}\n      wide vs 16-row: |z| {q(ez_p)}  |ld| {q(el_p)}\n"
          f"      wide vs fp64: |z| {(zw.cpu().double() - z64).abs().max():.2e} |ld| {(ldw.cpu().double() - ld64).abs().max():.2e}")
    assert torch.isfinite(zw).all() and torch.isfinite(ldw).all()
    # typical row: the tolerances of test_forward_bf16_tolerance, or within 3x of the 16-row kernel's own median.  (An
    # eight-layer flow with the final layers x2 AND tail entries of 1.2 B is chaotic for both kernels -- worst rows 0.4 /
    # 0.6 from the emulation, 3 from fp64 -- and is not a parity case; the bench workload (x2, tails of 6 = 1.2 B on 2 %
    # of the entries) is covered at full size by test_wide_full_size_statistics and bench.py's own check.)
    assert ez_w.median() < max(5e-4, 3 * ez_a.median().item()) and el_w.median() < max(5e-3, 3 * el_a.median().item())
    # worst row: no further from the emulation than the 16-row kernel's own worst row (x1.5), floors as in
    # test_forward_bf16_tolerance (the x2 eight-layer case with tail entries is ill-conditioned for both: 0.1 / 1.2)
    assert ez_w.max() < max(0.1, 1.5 * ez_a.max().item()) and el_w.max() < max(0.5, 1.5 * el_a.max().item())
    assert ez_p.median() < max(5e-4, 3 * ez_a.median().item()) and el_p.median() < max(5e-3, 3 * el_a.median().item())
    # against fp64 both bf16 kernels are bounded only loosely (test_forward_bf16_tolerance: 1.0 / 4.0 on the worst row of a
    # well-conditioned map); here: the wide kernel's typical row is as close to fp64 as the 16-row kernel's
    e64 = lambda zz, ll: ((zz.cpu().double() - z64).abs().max(dim=1).values.median().item(), (ll.cpu().double() - ld64).abs().median().item())
    assert e64(zw, ldw)[0] < max(1e-3, 1.5 * e64(za, lda)[0]) and e64(zw, ldw)[1] < max(1e-2, 1.5 * e64(za, lda)[1])


def test_wide_ragged_order_log_sigma_reduction_and_layer_inputs(kernel):
    D, L = 15, 3
    ref, _, flow = _pair(D, L)
    order = [2, 0, 1, 10, 9, 3, 4, 8, 5, 7, 6, 14, 12, 13, 11]
    ref.set_autoregressive_order(order)
    flow.set_autoregressive_order(order)
    for B in (1, 31, 64, 65, 129, 1000):
        x, ctx = flow_inputs(B, D, 288, 5.0, tails=B > 3)
        ls = torch.randn(B, D, generator=torch.Generator().manual_seed(B)) * 0.3
        xg, cg, lg = x.cuda().contiguous(), ctx.cuda().contiguous(), ls.cuda()
        with torch.no_grad():
            a, w = _both(flow, lambda: flow.compute_psd_aware_nll(xg, cg, lg))
            assert w.shape == (B,) and torch.isfinite(w).all()
            err = (w - a).abs() / a.abs().clamp_min(1.0)
            assert err.median() < 1e-4 and err.max() < 5e-2, (B, err.max())
            # in-kernel (sum nll, rows) of the wide kernel
            nll = torch.empty(B, device="cuda")
            slots = torch.zeros(16, 2, device="cuda")
            flow.nll_into(xg, cg, nll, sum_count=slots)
            acc = slots.double().sum(0)
            want = flow.compute_psd_aware_nll(xg, cg, None)
            assert torch.equal(nll, want) and acc[1].item() == B
            assert abs(acc[0].item() - want.double().sum().item()) <= 2e-5 * want.double().abs().sum().item() + 1e-3
            # training forward: the conditioner input of every layer (what the backward re-evaluates from)
            ua = torch.empty(L, B, D, device="cuda"); uw = torch.empty(L, B, D, device="cuda")
            flow.wide_min_batch = 1 << 40
            flow._forward_call(xg, cg, None, guard=False, layer_inputs=ua)
            flow.wide_min_batch = 1
            flow._forward_call(xg, cg, None, guard=False, layer_inputs=uw)
            assert torch.equal(ua[0], uw[0])                    # layer 0's input is x itself, permuted
            assert (ua - uw).abs().median() < 1e-4 and (ua - uw).abs().max() < 5e-2


def test_wide_is_refused_where_it_is_not_built():
    from posteriflow_amd import _lib
    _, _, flow = make_pair(7, 40, 128, 2, 10, 2.5)
    flow.precision = "bf16"
    flow.wide_min_batch = 1
    assert not flow._use_wide(1 << 20)                          # shape not built: the 16-row kernel serves it
    x, ctx = flow_inputs(64, 7, 40, 2.5)
    z, _ = flow(x.cuda(), ctx.cuda())
    assert torch.isfinite(z).all()
    _, _, flow = _pair(15, 2)
    flow.precision = "fp32"
    assert not flow._use_wide(1 << 20)                          # fp32 parity mode never uses it
    d = flow._desc(wide=True)
    assert _lib.lib().pf_flow_inverse(d, 16, 16, 16, 1, None, 1, 16, None, None, None, 0, None) == _lib.PF_ERR_UNSUPPORTED


@pytest.mark.parametrize("B,name,rows", [(65536, "pf::flow_wide_kernel<15, 18>", 128), (16384, "pf::flow_mid_kernel<15, 18>", 64),
                                         (9000, "pf::flow_mid_kernel<15, 18>", 64), (20000, "pf::flow_wide_kernel<15, 18>", 128),
                                         (40000, "pf::flow_mid_kernel<15, 18>", 64),
                                         (8192, "pf::flow_kernel<true, 16, 2, 9, 0, false>", 32)])
def test_wide_full_size_statistics(B, name, rows):
    """The sizes the two kernels exist for, through pf_flow_forward's OWN choice (no forcing): 65 536 rows -> the large-batch
    kernel, 16 384 / 9 000 -> the mid-batch kernel (one round), 40 000 -> the mid-batch kernel again (three rounds against two
    mostly-empty ones of the large-batch kernel), 20 000 -> the large-batch kernel, 8 192 -> still the 16-row kernel.  Finite everywhere, the same mean NLL as
    the 16-row kernel to 1e-4 relative, and 64 sampled rows against the fp32 oracle like
    test_rows_per_workgroup_choice_and_parity."""
    from posteriflow_amd import _lib
    D, L = 15, 8
    ref, _, flow = _pair(D, L, 2.0)
    assert "PF_FLOW_MID" not in os.environ and "PF_FLOW_WIDE" not in os.environ
    assert flow.forward_kernel_name(B) == name
    assert _lib.lib().pf_flow_rows_per_workgroup(flow._desc(wide=flow._use_wide(B)), B) == rows
    x, ctx = flow_inputs(B, D, 288, 5.0)
    xg, cg = x.cuda(), ctx.cuda()
    with torch.no_grad():
        flow.wide_min_batch = 1 << 40
        a = flow.compute_psd_aware_nll(xg, cg, None)
        flow.wide_min_batch = type(flow).wide_min_batch          # the product's own threshold
        w = flow.compute_psd_aware_nll(xg, cg, None)
        idx = torch.linspace(0, B - 1, 64).long()
        want = ref.compute_psd_aware_nll(x[idx], ctx[idx], torch.zeros(64, D))
    assert torch.isfinite(w).all()
    assert abs(w.double().mean().item() - a.double().mean().item()) < 1e-4 * abs(a.double().mean().item())
    err = (w.cpu()[idx] - want).abs() / want.abs().clamp_min(1.0)
    assert err.median() < 1e-2 and err.max() < 0.16           # bf16 on the 8-layer x2 flow, 64 rows: measured 4.7e-3 / 7.9e-2 (2x)
