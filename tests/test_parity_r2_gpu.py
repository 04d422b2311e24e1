"""Round-2 parity cases ("close the parity gaps"):

  * the bench workload itself (BASELINE config 3 flow, B = 4096, final layer x2.0, bench.py's seed-1 batch) in
    fp32 and bf16 against the oracle on ALL rows;
  * a shallow (L = 2) flow with the final layer x30 (raw spline parameters of std ~ 1: every bin, large
    derivatives) forward + inverse;
  * the product's log_prob with temperature != 1 and the clamp (a11, flows.py:657-695), compute_bounds_penalty and
    compute_endpoint_loss (a13, flows.py:910-939) against the oracle;
  * CoherentEncoder on the GPU through the HIP stem + (bf16) the HIP token mixer with its 4 geometry tokens (a19)
    against the reference-made golden context;
  * a features = 1 flow round trip.

Tolerances are written at each assert.
"""
import os
import sys

import numpy as np
import pytest
import torch

import recipe
from helpers import flow_inputs, make_pair, oracle_state_for_product

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_pair():
    sys.path.insert(0, ROOT)
    import bench
    ref, ref64, flow = make_pair(bench.D, bench.C, bench.H, bench.L, bench.K, bench.TB, scale=bench.FINAL_LAYER_SCALE)
    x, ctx = bench.make_inputs(4096, 1, "cpu")
    return bench, ref, ref64, flow, x, ctx


def test_bench_workload_full_size_fp32_and_bf16():
    """D15 / C288 / H256 / L8, B = 4096, final layer x2.0 -- every row against the oracle.
    fp32: 1e-5 relative on the NLL against fp64 at the MEDIAN (measured 1.3e-6); at the 99th percentile this
    8-layer map with the final layers x2 is ill-conditioned for fp32 arithmetic as such (the CPU fp32 oracle is
    2.3e-5 from fp64 there, worst row 9e-4), so the tail is held to 2x the CPU fp32 path's own distance from fp64
    (measured 1.25x) at p99, 3x at p99.9, 4x on the worst row; and the two fp32 evaluations agree to 5e-5 at p99.
    bf16: against the oracle with the same operand rounding (median 1e-2 abs on the NLL, p99 3); against fp64: no
    further than that emulation itself (bf16 operands cost ~0.6 nats at the median on this workload), and reported."""
    from oracle import nflows_restated as nfr
    bench, ref, ref64, flow, x, ctx = _bench_pair()
    zeros = torch.zeros_like(x)
    with torch.no_grad():
        n64 = ref64.compute_psd_aware_nll(x.double(), ctx.double(), zeros.double())
        n32 = ref.compute_psd_aware_nll(x, ctx, zeros).double()
        with nfr.gemm_emulation("bf16"):
            nemu = ref.compute_psd_aware_nll(x, ctx, zeros).double()
        xg, cg = x.cuda(), ctx.cuda()
        flow.precision = "fp32"
        got32 = flow.compute_psd_aware_nll(xg, cg, None).cpu().double()
        z32, ld32 = flow(xg, cg)
        flow.precision = "bf16"
        got16 = flow.compute_psd_aware_nll(xg, cg, None).cpu().double()
    den = n64.abs().clamp_min(1.0)
    rel, rel_ref = (got32 - n64).abs() / den, (n32 - n64).abs() / den
    print(f"\n[bench workload fp32, 4096 rows] rel nll: p50 {rel.median():.2e} p99 {rel.quantile(0.99):.2e} "
          f"max {rel.max():.2e} (cpu fp32: p99 {rel_ref.quantile(0.99):.2e} max {rel_ref.max():.2e})")
    rel_pair = (got32 - n32).abs() / den
    print(f"   HIP fp32 vs CPU fp32: p50 {rel_pair.median():.2e} p99 {rel_pair.quantile(0.99):.2e} max {rel_pair.max():.2e}")
    assert rel.median() < 1e-5 and rel.quantile(0.99) < max(1e-5, 2 * rel_ref.quantile(0.99).item())
    assert rel.max() < max(1e-5, 4 * rel_ref.max().item())
    assert rel.quantile(0.999) < max(1e-5, 3 * rel_ref.quantile(0.999).item())
    assert rel_pair.quantile(0.99) < 5e-5
    with torch.no_grad():
        z64, ld64 = ref64(x.double(), ctx.double())
        zc, ldc = ref(x, ctx)
    ez, el = (z32.cpu().double() - z64).abs(), (ld32.cpu().double() - ld64).abs()
    ez_ref, el_ref = (zc.double() - z64).abs(), (ldc.double() - ld64).abs()
    print(f"   |z - z64| p99 {ez.quantile(0.99):.2e} max {ez.max():.2e} (cpu fp32: {ez_ref.quantile(0.99):.2e} / {ez_ref.max():.2e})  "
          f"|ld - ld64| p99 {el.quantile(0.99):.2e} max {el.max():.2e} (cpu fp32: {el_ref.quantile(0.99):.2e} / {el_ref.max():.2e})")
    assert ez.quantile(0.99) < max(2e-5, 2 * ez_ref.quantile(0.99).item()) and ez.max() < max(1e-4, 4 * ez_ref.max().item())
    assert el.quantile(0.99) < max(1e-4, 2 * el_ref.quantile(0.99).item()) and el.max() < max(1e-3, 4 * el_ref.max().item())
    e_emu, e64 = (got16 - nemu).abs(), (got16 - n64).abs()
    print(f"[bench workload bf16, 4096 rows] |nll - bf16-emulating oracle| p50 {e_emu.median():.2e} p99 "
          f"{e_emu.quantile(0.99):.2e} max {e_emu.max():.2e};  vs fp64: p50 {e64.median():.2e} p99 "
          f"{e64.quantile(0.99):.2e} max {e64.max():.2e} (mean nll {n64.mean():.1f})")
    eo = (nemu - n64).abs()
    print(f"   same-rounding oracle vs fp64: p50 {eo.median():.2e} p99 {eo.quantile(0.99):.2e} max {eo.max():.2e}")
    # the kernel IS the bf16 arithmetic: tight against the same-rounding oracle (measured p50 2.7e-3, p90 0.2, p99 1.2: a
    # rounding-boundary flip of one activation, amplified by the later layers).  What bf16 operands cost against fp64 on
    # this random-weight 8-layer x2 workload (median 0.6 nats of ~135, p99 8.5, worst row 33 -- the CPU emulation shows
    # the same numbers) is not the kernel's to fix: it must be no further from fp64 than the emulation is.
    assert e_emu.median() < 1e-2 and e_emu.quantile(0.9) < 0.5 and e_emu.quantile(0.99) < 3.0
    assert e64.median() < 1.2 * eo.median() + 1e-3 and e64.quantile(0.99) < 1.2 * eo.quantile(0.99) + 1e-2
    assert abs(got16.mean().item() - nemu.mean().item()) < 2e-2      # the loss a trainer would log, same arithmetic


@pytest.mark.parametrize("scale", [5.0, 30.0])
def test_shallow_flow_scaled_final_layer_forward_and_inverse(scale):
    """L = 2, final layer x5 (raw spline parameters of std ~ 1.1: the regime BASELINE.md section 3 describes) and x30
    (its literal factor: std ~ 6.6, near one-hot bin widths, derivatives up to e^20), shallow enough to stay
    well-conditioned: all 16 bins, derivatives far from 1.  fp32 forward within 4x the CPU fp32 path's own error
    against fp64 (floors 2e-5 / 1e-4), NLL 1e-5 relative at p99 (or 4x the CPU path's p99); inverse likewise."""
    D, C, H, L, K, tb, B = 15, 288, 256, 2, 16, 5.0, 1024
    ref, ref64, flow = make_pair(D, C, H, L, K, tb, scale=scale)
    x, ctx = flow_inputs(B, D, C, tb)
    with torch.no_grad():
        net = ref.transform._transforms[1].autoregressive_net
        raw = net(x.flip(1), ctx)
        print(f"\n[x{scale:g}] raw spline parameter std {raw.std():.2f}")
        assert raw.std() > 1.0
        z64, ld64 = ref64(x.double(), ctx.double())
        z32, ld32 = ref(x, ctx)
        n64 = ref64.compute_psd_aware_nll(x.double(), ctx.double(), torch.zeros_like(x).double())
        flow.precision = "fp32"
        z, ld = flow(x.cuda(), ctx.cuda())
        nll = flow.compute_psd_aware_nll(x.cuda(), ctx.cuda(), None).cpu().double()
        rz, rld = (z.cpu().double() - z64).abs().max(dim=1).values, (ld.cpu().double() - ld64).abs()
        rz_ref, rld_ref = (z32.double() - z64).abs().max(dim=1).values, (ld32.double() - ld64).abs()
        ez, eld, ez_ref, eld_ref = rz.max().item(), rld.max().item(), rz_ref.max().item(), rld_ref.max().item()
        n32 = ref.compute_psd_aware_nll(x, ctx, torch.zeros_like(x)).double()
        rel = (nll - n64).abs() / n64.abs().clamp_min(1.0)
        rel_ref = (n32 - n64).abs() / n64.abs().clamp_min(1.0)
        print(f"[x{scale:g} fp32 fwd] |z-z64| {ez:.2e} (cpu {ez_ref:.2e}) |ld-ld64| {eld:.2e} (cpu {eld_ref:.2e}) "
              f"rel nll p99 {rel.quantile(0.99):.2e} max {rel.max():.2e}")
        if scale < 10:
            assert ez < max(4 * ez_ref, 2e-5) and eld < max(4 * eld_ref, 1e-4)
        else:
            # x30: derivatives up to e^20 and near one-hot bins -- a row whose input lands within rounding of a knot takes
            # the neighbouring bin under ANY change of summation order, and its log|det| then moves by the log-ratio of
            # two derivatives (the CPU fp32 path's own worst row is 0.4 from fp64; the kernel's, on another row, 3.9).
            # The worst row is therefore not a parity statistic here: hold p99 to 4x the CPU path's p99 and bound the
            # share of rows beyond 4x the CPU path's worst row to 0.5 %.
            q = lambda t: t.quantile(0.99).item()
            print(f"      p99 |z-z64| {q(rz):.2e} (cpu {q(rz_ref):.2e})  p99 |ld-ld64| {q(rld):.2e} (cpu {q(rld_ref):.2e})  "
                  f"rows beyond 4x cpu max: z {(rz > 4 * ez_ref).sum().item()} ld {(rld > 4 * eld_ref).sum().item()} of {B}")
            assert q(rz) < max(4 * q(rz_ref), 2e-5) and q(rld) < max(4 * q(rld_ref), 1e-4)
            assert (rz > 4 * ez_ref).float().mean() < 5e-3 and (rld > 4 * eld_ref).float().mean() < 5e-3
        assert rel.quantile(0.99) < max(1e-5, 4 * rel_ref.quantile(0.99).item())
        # inverse of points the flow maps to: oracle fp64 inverse, both kernels
        zz = z64.float()
        x64, ldi64 = ref64.inverse_raw(zz.double(), ctx.double())
        try:
            x32, ldi32 = ref.inverse_raw(zz, ctx)
            rx_ref, rli_ref = (x32.double() - x64).abs().max(dim=1).values, (ldi32.double() - ldi64).abs()
            ex_ref, eli_ref = rx_ref.max().item(), rli_ref.max().item()
        except AssertionError:
            # x30: nflows' own `assert (discriminant >= 0).all()` fires in fp32 on the CPU -- the reference cannot invert
            # this regime in fp32 at all.  The kernels flag such rows (bit 0 of fail_flags) instead of aborting; they are
            # held to the fp64 inverse on the rows they do not flag.
            x32, ex_ref, eli_ref = None, float("nan"), float("nan")
            print(f"[x{scale:g}] the CPU fp32 oracle inverse hits nflows' negative-discriminant assertion")
        for inc in (None, False):
            flow.incremental_inverse = inc
            xi, ldi, flags = flow._inverse_call(zz.cuda().contiguous(), ctx.cuda().contiguous(), B)
            rx, rli = (xi.cpu().double() - x64).abs().max(dim=1).values, (ldi.cpu().double() - ldi64).abs()
            ex, eli = rx.max().item(), rli.max().item()
            print(f"[x{scale:g} fp32 inv {'incremental' if inc is None else 'D-pass'}] |x-x64| {ex:.2e} (cpu {ex_ref:.2e}) "
                  f"|ld-ld64| {eli:.2e} (cpu {eli_ref:.2e})")
            if x32 is None:
                # ... and the fp64 oracle shows why: one fp32 ulp of z (1e-7 relative) moves the fp64 inverse by the whole
                # interval on the typical row (measured p50 9.9 of a possible 10), i.e. the inverse of this map is not
                # defined at fp32 input resolution.  What remains checkable: finite, few rows flagged.
                xp, _ = ref64.inverse_raw(zz.double() * (1 + 1e-7), ctx.double())
                sens = (xp - x64).abs().max(dim=1).values
                ok = flags.cpu() == 0
                print(f"      fp64 inverse under a 1e-7 relative perturbation of z moves by p50 {sens.median():.2e}; flagged rows "
                      f"{int((~ok).sum())} of {B}; |x-x64| p50 {rx.median():.2e}")
                assert sens.median() > 1.0                      # if this ever fails the accuracy asserts below apply again
                assert (~ok).float().mean() < 0.02 and torch.isfinite(xi).all()
                continue
            assert int(flags.sum()) == 0
            if scale < 10:
                assert ex < max(4 * ex_ref, 1e-4) and eli < max(4 * eli_ref, 5e-4)
            else:                   # as for the forward: p99 and the share of rows beyond the CPU path's worst row
                q = lambda t: t.quantile(0.99).item()
                print(f"      p99 |x-x64| {q(rx):.2e} (cpu {q(rx_ref):.2e})  p99 |ld-ld64| {q(rli):.2e} (cpu {q(rli_ref):.2e})")
                assert q(rx) < max(4 * q(rx_ref), 1e-4) and q(rli) < max(4 * q(rli_ref), 5e-4)
                assert (rx > 4 * ex_ref).float().mean() < 5e-3 and (rli > 4 * eli_ref).float().mean() < 5e-3
            # round trip: the inverse of a map with derivatives down to 1e-3 amplifies the fp32 rounding of z by up to
            # 1e3 per layer (the CPU fp32 inverse is itself up to 10 away from the fp64 one on the worst row at x5, the
            # round trip closes to 8e-3 at the median): this regime is held to the CPU path's own distance from fp64
            # above, the round-trip property to test_round_trip_full_size (default init)
            z2, ldf = flow(xi, ctx.cuda())
            rt, rl = (z2.cpu() - zz).abs().max(dim=1).values, (ldf + ldi).abs().cpu()
            print(f"      round trip |z2 - z| p50 {rt.median():.1e} p90 {rt.quantile(0.9):.1e} max {rt.max():.1e}  "
                  f"|ld_f + ld_i| p50 {rl.median():.1e} p90 {rl.quantile(0.9):.1e}")
            assert torch.isfinite(rt).all() and torch.isfinite(rl).all()   # reported above; test_round_trip_full_size holds the well-conditioned case
        # bf16 throughput mode on the same regime: against the same-rounding oracle
        from oracle import nflows_restated as nfr
        with nfr.gemm_emulation("bf16"):
            zemu, ldemu = ref(x, ctx)
        flow.precision = "bf16"
        zb, ldb = flow(x.cuda(), ctx.cuda())
        dz, dl = (zb.cpu() - zemu).abs().max(dim=1).values, (ldb.cpu() - ldemu).abs()
        print(f"[x{scale:g} bf16 fwd vs bf16-emulating oracle] |z| p50 {dz.median():.1e} max {dz.max():.1e}  |ld| p50 "
              f"{dl.median():.1e} max {dl.max():.1e}")
        assert dz.median() < 2e-3 and dl.median() < 2e-2
        if scale < 10:
            assert dz.max() < 0.5 and dl.max() < 2.0


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_log_prob_temperature_and_clamp(precision):
    """a11 (flows.py:657-695 to its documented math): -[log p(x / T) - D log T], T = clamp(temperature, 0.5, 3) when
    not passed, used as given when passed.  fp32: 1e-5 relative against the oracle (fp32 CPU); bf16: 2e-2."""
    D, C = 11, 288
    ref, _, flow = make_pair(D, C, 256, 3, 16, 5.0)
    flow.precision = precision
    tol = 1e-5 if precision == "fp32" else 2e-2
    x, ctx = flow_inputs(200, D, C, 5.0)
    ctx[3, 5] = float("nan"); ctx[4, 7] = float("inf")               # sanitised like flows.py:664-669
    with torch.no_grad():
        for T_param, T_arg in ((1.5, None), (4.0, None), (0.1, None), (1.0, 0.7), (1.0, 2.0), (1.0, 5.0)):
            ref.temperature.fill_(T_param); flow.temperature.fill_(T_param)
            want = ref.log_prob(x, ctx, T_arg)
            got = flow.log_prob(x.cuda(), ctx.cuda(), T_arg).cpu()
            rel = ((got - want).abs() / want.abs().clamp_min(1.0))
            print(f"\n[log_prob {precision} T_param={T_param} T_arg={T_arg}] rel err p99 {rel.quantile(0.99):.2e} max {rel.max():.2e}")
            assert rel.quantile(0.99) < tol and rel.max() < 20 * tol
        # T = 1: identical to the live density (a10)
        ref.temperature.fill_(1.0); flow.temperature.fill_(1.0)
        a = flow.log_prob(x.cuda(), ctx.nan_to_num(0.0, 1e-3, -1e-3).cuda())
        b = flow.compute_psd_aware_nll(x.cuda(), ctx.nan_to_num(0.0, 1e-3, -1e-3).cuda(), None)
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-5)
        assert torch.allclose(flow.compute_nll_loss(x.cuda(), ctx.cuda()).cpu(), ref.log_prob(x, ctx).mean(),
                              rtol=10 * tol, atol=10 * tol)


def test_bounds_penalty_and_endpoint_loss_against_the_oracle():
    """a13: compute_bounds_penalty (flows.py:910-920, exact) and compute_endpoint_loss (:922-939: inverse at
    z = -3 / +3 per context row, relu distances; 1e-4 absolute in fp32)."""
    D, C = 11, 288
    ref, _, flow = make_pair(D, C, 256, 3, 16, 5.0, scale=4.0)
    g = torch.Generator().manual_seed(4)
    p = torch.randn(64, D, generator=g) * 2.5                          # many entries beyond +-3
    ctx = torch.randn(64, C, generator=g)
    with torch.no_grad():
        assert torch.equal(flow.compute_bounds_penalty(p.cuda()).cpu(), ref.compute_bounds_penalty(p))
        assert torch.allclose(flow.compute_bounds_penalty(p.cuda(), (-1.0, 0.5)).cpu(),
                              ref.compute_bounds_penalty(p, (-1.0, 0.5)))
        want = ref.compute_endpoint_loss(p, ctx)
        for inc in (None, False):
            flow.incremental_inverse = inc
            got = flow.compute_endpoint_loss(p.cuda(), ctx.cuda()).cpu()
            print(f"\n[endpoint loss {'incremental' if inc is None else 'D-pass'}] {got.item():.6f} vs oracle {want.item():.6f}")
            assert want.item() > 1e-3 and abs(got.item() - want.item()) < 1e-4


def _load(enc, seed, skip):
    shapes = {k: v.shape for k, v in enc.state_dict().items() if k not in skip}
    missing = enc.load_state_dict(recipe.fill_state_dict(shapes, seed=seed), strict=False)
    assert sorted(missing.missing_keys) == sorted(skip)
    for p in enc.parameters():
        p.requires_grad_(False)
    return enc.eval()


def test_coherent_encoder_on_the_gpu_against_reference_golden(golden_encoder):
    """a19 (coherent_encoder.py:42-123): geometry features (rocFFT) + HIP stem + token mixer with the 4 geometry
    tokens prepended (187 tokens).  fp32 mode (HIP stem, fp32 transformer): 2e-3 relative / 1e-3 absolute against
    the reference's own context; bf16 mode runs the fused HIP mixer: 4e-2 of the context's scale."""
    from posteriflow_amd import npe
    enc = _load(npe.CoherentEncoder(context_dim=256, psd_bands=16), 200, ("pos.pe", "Bsum", "bcount", "lags_norm")).cuda()
    strain = recipe.strain_batch(4, 3, seed=9).cuda()
    asd = torch.from_numpy(golden_encoder["coh_asd"]).cuda()
    gold = golden_encoder["coh_ctx"]
    with torch.no_grad():
        rel = enc._geometry_rel(enc._sanitize(strain)).cpu().numpy()
        np.testing.assert_allclose(rel, golden_encoder["coh_rel"], rtol=2e-3, atol=2e-3)
        enc.precision = "fp32"
        ctx = enc(strain, asd).cpu().numpy()
        print(f"\n[coherent fp32] max abs err {np.abs(ctx - gold).max():.2e} (scale {np.abs(gold).max():.2f})")
        np.testing.assert_allclose(ctx, gold, rtol=2e-3, atol=1e-3)
        enc.precision = "bf16"
        enc.__dict__.pop("_mixer_state", None)
        ctx16 = enc(strain, asd).cpu().numpy()
        assert "_mixer_state" in enc.__dict__                          # the fused HIP mixer ran (187 tokens)
        err = np.abs(ctx16 - gold).max() / np.abs(gold).max()
        print(f"[coherent bf16, HIP mixer] max err / scale {err:.2e}")
        assert err < 4e-2


def test_single_feature_flow_round_trip():
    """features = 1: every hidden unit has degree 0... the D-pass kernel serves it."""
    ref, ref64, flow = make_pair(1, 8, 64, 3, 8, 3.0, scale=3.0)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(100, 1, generator=g) * 5 - 2.5
    ctx = torch.randn(100, 8, generator=g)
    with torch.no_grad():
        z, ld = flow(x.cuda(), ctx.cuda())
        zr, ldr = ref(x, ctx)
        assert torch.allclose(z.cpu(), zr, atol=2e-5) and torch.allclose(ld.cpu(), ldr, atol=1e-4)
        xi, ldi = flow.inverse(z, ctx.cuda())
        # x3 final layers: derivatives down to ~1e-3, so the inverse amplifies the fp32 rounding of z by up to 1e3 on a few
        # rows -- held to the CPU fp32 inverse's own distance from the fp64 inverse of the same z, typical row to 1e-5
        x64, ldi64 = ref64.inverse_raw(z.cpu().double(), ctx.double())
        x32, ldi32 = ref.inverse_raw(z.cpu(), ctx)
        ex, ex_ref = (xi.cpu().double() - x64).abs(), (x32.double() - x64).abs()
        el, el_ref = (ldi.cpu().double() - ldi64).abs(), (ldi32.double() - ldi64).abs()
        print(f"\n[D=1] |x - x64| p50 {ex.median():.1e} max {ex.max():.1e} (cpu fp32 {ex_ref.max():.1e});  |ld - ld64| max {el.max():.1e} "
              f"(cpu fp32 {el_ref.max():.1e});  round trip |x' - x| p50 {(xi.cpu() - x.clamp(-3, 3)).abs().median():.1e} max {(xi.cpu() - x.clamp(-3, 3)).abs().max():.1e}")
        assert ex.max() < max(4 * ex_ref.max().item(), 1e-4) and el.max() < max(4 * el_ref.max().item(), 1e-3)
        assert (xi.cpu() - x.clamp(-3, 3)).abs().median() < 1e-5 and (ld + ldi).abs().median() < 1e-4
        s = flow.sample(7, ctx[:3].cuda())
        assert s.shape == (3, 7, 1) and torch.isfinite(s).all()
