"""pf_geom_features (csrc/pf_geom.hip) -- the coherent encoder's frequency-domain geometry features
(src/ahsd/models/coherent_encoder.py:79-116) -- against a float64 evaluation of the same formulas, the fp32 oracle
(oracle/lean_ref.py CoherentGeometry, golden-pinned in test_oracle_golden.py) and the reference-made golden vector."""
import math
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import recipe  # noqa: E402
from oracle import lean_ref  # noqa: E402

pytestmark = pytest.mark.gpu


def rel64(geom, clean):
    """coherent_encoder.py:93-116 in float64 (numpy), plus |cc| on the lag window for the tie analysis"""
    x = clean.double().numpy()
    fd = np.fft.rfft(x, norm="ortho", axis=-1)[..., geom.band_lo: geom.band_lo + geom.Nf]
    P = fd.real ** 2 + fd.imag ** 2
    amp = np.sqrt(P + 1e-12)
    Bs, cnt = geom.Bsum.double().numpy(), geom.bcount.double().numpy()
    feats, windows = [np.log(P @ Bs.T / cnt + 1e-8).reshape(x.shape[0], -1)], []
    for i, j in geom.pairs:
        X = fd[:, i] * np.conj(fd[:, j])
        den = (amp[:, i] * amp[:, j]) @ Bs.T + 1e-8
        gr, gi = X.real @ Bs.T / den, X.imag @ Bs.T / den
        gm = np.sqrt(gr ** 2 + gi ** 2) + 1e-8
        feats += [gm, gr / gm, gi / gm]
        full = np.zeros((x.shape[0], geom.n_rfft), complex)
        full[:, geom.band_lo: geom.band_lo + geom.Nf] = X
        cc = np.fft.irfft(full, n=x.shape[-1], axis=-1)
        a = np.abs(np.concatenate([cc[:, -geom.maxlag:], cc[:, : geom.maxlag + 1]], axis=1))
        windows.append(a)
        lag = (a.argmax(-1) - geom.maxlag) / geom.maxlag
        feats += [lag[:, None], (a.max(-1) / (a.mean(-1) + 1e-8))[:, None],
                  (np.log(P[:, i].sum(-1) + 1e-8) - np.log(P[:, j].sum(-1) + 1e-8))[:, None]]
    return np.concatenate(feats, -1), windows


def delayed_events(batch, n_det, seed):
    """unit noise + a common band-limited burst arriving at detector d with a delay of 9 d - 4 b samples: a clear GCC peak"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, n_det, 16384, generator=g)
    t = torch.arange(16384, dtype=torch.float32) / 4096.0
    burst = torch.sin(2 * math.pi * (60.0 * t + 45.0 * t * t)) * torch.exp(-((t - 2.0) / 0.4) ** 2)
    for b in range(batch):
        for d in range(n_det):
            x[b, d] += (3.0 + b) * torch.roll(burst, 9 * d - 4 * b)
    return x


def check(enc, geom, clean, tol=2e-5):
    from posteriflow_amd import npe  # noqa: F401
    edges = enc._geometry_plan()
    assert edges is not None and edges[0] == 0 and edges[-1] == geom.Nf
    got = enc._geometry_rel_hip(clean.cuda(), edges).cpu().numpy().astype(np.float64)
    want, windows = rel64(geom, clean)
    K, nd = enc.K, enc.n_detectors
    assert got.shape == want.shape
    lag_cols = [nd * K + p * (3 * K + 3) + 3 * K for p in range(len(geom.pairs))]
    other = np.ones(want.shape[1], bool)
    other[lag_cols] = False
    # everything but the arg-max: fp32 transforms and sums against float64
    err = np.abs(got[:, other] - want[:, other]) / np.maximum(np.abs(want[:, other]), 1.0)
    print(f"\n[geometry features] max err vs float64 {err.max():.2e} (bound {tol:.0e})")
    assert err.max() < tol, (err.max(), np.unravel_index(err.argmax(), err.shape))
    for p, col in enumerate(lag_cols):          # the lag: identical unless float64 itself has a near tie on the window
        a = windows[p]
        top2 = np.sort(a, axis=-1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-5 * top2[:, 1]
        assert np.array_equal(got[clear, col], want[clear, col].astype(np.float32).astype(np.float64)), (p, got[:, col], want[:, col])
        # a near tie may pick the other peak: its height must be within fp32 noise of the maximum
        idx = np.rint(got[:, col] * geom.maxlag).astype(int) + geom.maxlag
        assert np.all(a[np.arange(a.shape[0]), idx] >= top2[:, 1] * (1 - 1e-5))
    return got


@pytest.mark.parametrize("n_det,batch", [(3, 5), (2, 3), (1, 2), (3, 1)])
def test_geometry_features_match_float64(n_det, batch):
    from posteriflow_amd import npe
    torch.manual_seed(0)
    enc = npe.CoherentEncoder(context_dim=256, psd_bands=16, n_detectors=n_det).cuda().eval()
    geom = lean_ref.CoherentGeometry(n_det=n_det)
    assert [enc.band_lo, enc.Nf, enc.maxlag] == [geom.band_lo, geom.Nf, geom.maxlag]
    clean = delayed_events(batch, n_det, seed=n_det)
    got = check(enc, geom, clean)
    if n_det == 3:      # the injected delays are found: pair (i, j) peaks at lag 9 (i - j) samples
        K = enc.K
        for p, (i, j) in enumerate(geom.pairs):
            col = 3 * K + p * (3 * K + 3) + 3 * K
            assert np.allclose(got[:, col] * geom.maxlag, 9 * (i - j)), (p, got[:, col] * geom.maxlag)
    # the module's own entry point takes the HIP route on the GPU and agrees with its tensor-op (rocFFT) route
    with torch.no_grad():
        a = enc._geometry_rel(clean.cuda())
        enc.__dict__["_geom_plan"] = False          # no plan: tensor ops
        b = enc._geometry_rel(clean.cuda())
        enc.__dict__.pop("_geom_plan")
    assert np.array_equal(a.cpu().numpy(), got.astype(np.float32))
    assert torch.allclose(a, b, rtol=2e-3, atol=2e-3)


def test_geometry_features_match_the_reference_golden():
    """the reference's own CoherentEncoder._geometry_rel on recipe.strain_batch(4, 3, seed=9) (tests/golden/make_golden.py),
    non-finite samples sanitised as lean_npe.py:207 does"""
    from posteriflow_amd import npe
    gold = np.load(os.path.join(ROOT, "tests", "golden", "encoder.npz"))
    enc = npe.CoherentEncoder(context_dim=256, psd_bands=16).cuda().eval()
    geom = lean_ref.CoherentGeometry()
    clean = lean_ref.sanitize_strain(recipe.strain_batch(4, 3, seed=9))
    got = check(enc, geom, clean, tol=1e-4)       # (cos, sin) of a weak band coherence amplify the fp32 rounding of the sums
    np.testing.assert_allclose(got, gold["coh_rel"], rtol=1e-4, atol=2e-5)
    raw = recipe.strain_batch(4, 3, seed=9).cuda()                 # event 1 holds nan / +-inf / out-of-range samples
    assert not torch.isfinite(raw).all()
    with torch.no_grad():
        rel = enc._geometry_rel(enc._sanitize(raw))
        rel_raw = enc._geometry_rel_hip(raw, enc._geometry_plan(), sanitize=True)      # sanitised on load inside the kernel
    np.testing.assert_allclose(rel.cpu().numpy(), gold["coh_rel"], rtol=1e-4, atol=2e-5)
    assert torch.equal(rel, rel_raw)


def test_spectrum_workspace_is_the_ortho_rfft():
    from posteriflow_amd import _lib, npe
    enc = npe.CoherentEncoder(context_dim=256, psd_bands=16).cuda().eval()
    clean = delayed_events(2, 3, seed=5)
    x = clean.cuda()
    tw = torch.empty(8192, 2)
    _lib.check(_lib.lib().pf_geom_twiddles(tw.data_ptr()), "tw")
    m = np.arange(8192)
    assert np.array_equal(tw[:, 0].numpy(), np.cos(-2 * np.pi * m / 16384).astype(np.float32))
    a = _lib.PfGeomArgs()
    a.clean, a.batch, a.n_det, a.band_lo, a.nf, a.n_bands, a.maxlag = x.data_ptr(), 2, 3, enc.band_lo, enc.Nf, enc.K, enc.maxlag
    for i, e in enumerate(enc._geometry_plan()):
        a.band_edge[i] = e
    twd = tw.cuda()
    spec = torch.empty(2, 3, enc.Nf, 2, device="cuda")
    etot, rel = torch.empty(2, 3, device="cuda"), torch.empty(2, 201, device="cuda")
    a.twiddle, a.spec, a.etot, a.rel = twd.data_ptr(), spec.data_ptr(), etot.data_ptr(), rel.data_ptr()
    _lib.check(_lib.lib().pf_geom_features(a, torch.cuda.current_stream().cuda_stream), "pf_geom_features")
    want = np.fft.rfft(clean.double().numpy(), norm="ortho", axis=-1)[..., enc.band_lo: enc.band_lo + enc.Nf]
    got = spec.cpu().numpy().astype(np.float64)
    err = np.abs((got[..., 0] + 1j * got[..., 1]) - want).max() / np.abs(want).max()
    assert err < 2e-6, err
    assert np.allclose(etot.cpu().numpy(), (np.abs(want) ** 2).sum(-1), rtol=1e-5)
    # argument checks
    a.maxlag = 200
    assert _lib.lib().pf_geom_features(a, None) == _lib.PF_ERR_UNSUPPORTED
    a.maxlag, a.nf = enc.maxlag, 5000
    assert _lib.lib().pf_geom_features(a, None) == _lib.PF_ERR_UNSUPPORTED
