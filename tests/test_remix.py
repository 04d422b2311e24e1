"""Remix augmentation: product host logic vs the oracle on the CPU, and the HIP
assembly (pf_remix_forward) vs the oracle and the reference's golden vectors on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import recipe                                               # noqa: E402
from make_golden_remix import CASES, STRIDE                 # noqa: E402
from oracle.remix_ref import RemixRef, Decisions            # noqa: E402
from posteriflow_amd import _lib                            # noqa: E402
from posteriflow_amd.remix import RemixDataset, T_LEN       # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "remix.npz"))
HAS_GPU = torch.cuda.is_available()


@pytest.fixture(scope="module")
def cache(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("remix_cache"))
    recipe.remix_cache(d)
    return d


def _kw(cache, kw):
    kw = dict(kw)
    if kw.get("real_noise_dir") == "bank":
        kw["real_noise_dir"] = os.path.join(cache, "real_bank")
    return kw


def _decisions_of(plan, b):
    """oracle Decisions from row b of a product plan (host copies)."""
    n = int(plan.nsig[b])
    dec = Decisions(noise_idx=int(plan.noise_row[b]),
                    scale=[float(v) for v in plan.scale[b, :n].cpu()],
                    shift=[int(v) for v in plan.shift[b, :n].cpu()],
                    keep=tuple(d for d in range(3) if bool(plan.keep[b, d])))
    real = plan.real is not None and bool(plan.real["mask"][b])
    if real:
        r = plan.real
        dec.real = [(int(r["seg"][b, d]), int(r["off"][b, d]), bool(r["flip"][b, d])) for d in range(3)]
    for d in range(3):
        if d in dec.keep:
            continue
        if real:
            r = plan.real
            dec.refill[d] = (int(r["re_seg"][b, d]), int(r["re_off"][b, d]), bool(r["re_flip"][b, d]))
        else:
            dec.refill[d] = plan.fill[int(plan.fill_row[b, d])].cpu().numpy()
    return dec


# ---- host logic (CPU) -------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,kw,epoch", CASES, ids=[c[0] for c in CASES])
def test_reference_plan_reproduces_the_references_decisions(cache, tag, kw, epoch):
    kw = _kw(cache, kw)
    ours, ref = RemixDataset(cache, device="cpu", **kw), RemixRef(cache, **kw)
    ours.set_epoch(epoch), ref.set_epoch(epoch)
    idx = list(range(len(ref)))
    plan = ours.reference_plan(idx)
    np.testing.assert_array_equal(plan.labels.numpy(), GOLD[f"{tag}_pv"])          # the reference's labels
    np.testing.assert_array_equal(plan.nsig.numpy(), GOLD[f"{tag}_nsig"])
    for b in idx:
        want, got = ref.draw(b), _decisions_of(plan, b)
        assert got.noise_idx == want.noise_idx and got.keep == tuple(want.keep) and got.real == want.real
        np.testing.assert_array_equal(np.float32(got.scale), np.float32(want.scale))
        assert got.shift == want.shift
        assert sorted(got.refill) == sorted(want.refill)
        for d in got.refill:
            np.testing.assert_array_equal(np.asarray(got.refill[d]), np.asarray(want.refill[d]))


def test_device_plan_follows_the_relabel_rules(cache):
    """draws are the product's own; given them the labels must be what the reference's algebra gives,
    and the distributions must respect every guard."""
    ours = RemixDataset(cache, device="cpu", seed=1, det_dropout=0.5)
    ref = RemixRef(cache, seed=1, det_dropout=0.5)
    g = torch.Generator().manual_seed(5)
    idx = torch.arange(len(ours)).repeat(40)
    plan = ours.device_plan(idx, generator=g)
    assert plan.noise_row.min() >= 0 and plan.noise_row.max() < ours.n_noise
    assert len(torch.unique(plan.noise_row)) == ours.n_noise
    s = plan.scale[plan.scale != 1.0]
    assert s.min() >= 0.75 and s.max() <= 1.33 and abs(float(s.mean()) - 1.04) < 0.03
    nz = plan.shift[plan.shift != 0].float()
    assert plan.shift.abs().max() <= 409 and 200 < float(nz.std()) < 270
    assert 0.3 < float((~plan.keep).any(1).float().mean()) < 0.7 and plan.keep.any(1).all()
    for b in range(0, idx.numel(), 7):
        i = int(idx[b])
        np.testing.assert_allclose(plan.labels[b].numpy(), ref.relabel(i, _decisions_of(plan, b)), rtol=1e-6, atol=0)
        start, n = ref.events[i]
        for k in range(n):                                           # guards of remix_data.py:238, :246
            p = ref.params[start + k]
            if abs(p[8]) >= 1.45:
                assert int(plan.shift[b, k]) == 0
            sc = float(plan.scale[b, k])
            assert sc == 1.0 or 45.0 < p[2] / sc < 2100.0
        assert (plan.scale[b, n:] == 1).all() and (plan.shift[b, n:] == 0).all()


def test_assemble_refuses_the_cpu(cache):
    ds = RemixDataset(cache, device="cpu")
    with pytest.raises(_lib.PfError):
        ds.batch([0, 1], exact=True)


# ---- HIP assembly (GPU) -------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("tag,kw,epoch", CASES, ids=[c[0] for c in CASES])
def test_exact_batches_equal_the_references_examples(cache, tag, kw, epoch):
    kw = _kw(cache, kw)
    ds, ref = RemixDataset(cache, **kw), RemixRef(cache, **kw)
    ds.set_epoch(epoch), ref.set_epoch(epoch)
    idx = list(range(len(ds)))
    out = ds.batch(idx, exact=True)
    strain = out[0].cpu().numpy()
    want = [ref.item(i) for i in idx]
    real = np.array([ref.draw(i).real is not None for i in idx])
    for b in idx:
        if real[b]:        # re-coloured through hipFFT (fp32): tolerance instead of bit equality
            np.testing.assert_allclose(strain[b], want[b][0], rtol=0, atol=2e-5 * np.abs(want[b][0]).max())
        else:              # fp16 gather + fp32 scale/shift/sum: bit-exact, full length
            np.testing.assert_array_equal(strain[b], want[b][0])
    np.testing.assert_array_equal(strain[~real][:, :, ::STRIDE], GOLD[f"{tag}_strain_sub"][~real])
    np.testing.assert_array_equal(out[1].cpu().numpy(), GOLD[f"{tag}_pv"])
    np.testing.assert_array_equal(out[2].cpu().numpy(), GOLD[f"{tag}_nsig"])
    np.testing.assert_allclose(out[3].cpu().numpy(), GOLD[f"{tag}_snr"], rtol=2e-5 if real.any() else 2e-6)
    if kw.get("return_asd_bands"):
        np.testing.assert_array_equal(out[4].cpu().numpy(), GOLD[f"{tag}_asd_bands"])
    one = ds[3]                                                   # the reference's tuple layout
    assert one[0].shape == (3, T_LEN) and one[1].shape == (5, 11) and one[2].dtype == torch.int64
    np.testing.assert_array_equal(one[1].cpu().numpy(), GOLD[f"{tag}_pv"][3])


@pytest.mark.gpu
def test_device_plan_batches_equal_the_oracle_given_the_same_decisions(cache):
    kw = dict(seed=2, det_dropout=0.5, real_noise_dir=os.path.join(cache, "real_bank"), real_noise_prob=0.3,
              return_asd_bands=True, psd_bands=8)
    ds, ref = RemixDataset(cache, **kw), RemixRef(cache, **kw)
    g = torch.Generator(device="cuda").manual_seed(3)
    idx = torch.arange(len(ds)).repeat(6)
    plan = ds.device_plan(idx, generator=g)
    strain, labels, nsig, snr, asd = ds.assemble(plan)
    assert bool(plan.real["mask"].any()) and not bool(plan.real["mask"].all())
    for b in range(idx.numel()):
        i = int(idx[b])
        want = ref.apply(i, _decisions_of(plan, b))
        if bool(plan.real["mask"][b]):
            np.testing.assert_allclose(strain[b].cpu().numpy(), want[0], rtol=0, atol=2e-5 * np.abs(want[0]).max())
        else:
            np.testing.assert_array_equal(strain[b].cpu().numpy(), want[0])
        np.testing.assert_allclose(labels[b].cpu().numpy(), want[1], rtol=1e-6)
        assert int(nsig[b]) == want[2]
        np.testing.assert_allclose(float(snr[b]), want[3], rtol=2e-5)
        np.testing.assert_array_equal(asd[b].cpu().numpy(), want[4])


@pytest.mark.gpu
def test_remix_properties_at_training_batch_size():
    """size-independent properties on a 4096-example batch over a synthetic 512-row pool: with one
    unit-scale signal the example minus its noise row is the rolled signal; SNR is shift-invariant and
    scales linearly with the amplitude factor; rows outside the pools contribute nothing."""
    dev, B, L = torch.device("cuda"), 4096, _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(0)
    noise = torch.randn(512, 3, T_LEN, device=dev, generator=g).half()
    sig = (torch.randn(512, 3, T_LEN, device=dev, generator=g) * 0.1).half()
    row = torch.randint(0, 512, (B,), device=dev, generator=g)
    start = torch.randint(0, 512, (B,), device=dev, generator=g)
    nsig = torch.ones(B, dtype=torch.int32, device=dev)
    shift = torch.zeros(B, 5, dtype=torch.int32, device=dev)
    shift[:, 0] = torch.randint(-409, 410, (B,), device=dev, generator=g).int()
    scale = torch.ones(B, 5, device=dev)
    ws = torch.empty(L.pf_remix_workspace_bytes(B) // 8, dtype=torch.float64, device=dev)

    def run(scale, shift, row=row, start=start, nsig=nsig):
        strain, ssum, snr = (torch.empty(B, 3, T_LEN, device=dev), torch.empty(B, 3, T_LEN, device=dev),
                             torch.empty(B, device=dev))
        _lib.check(L.pf_remix_forward(noise.data_ptr(), 512, sig.data_ptr(), 512, row.data_ptr(), start.data_ptr(),
                                      nsig.data_ptr(), scale.data_ptr(), shift.data_ptr(), 0, 0, 0, B,
                                      strain.data_ptr(), ssum.data_ptr(), snr.data_ptr(), ws.data_ptr(),
                                      ws.numel() * 8, torch.cuda.current_stream().cuda_stream), "pf_remix_forward")
        return strain, ssum, snr

    strain, ssum, snr = run(scale, shift)
    for b in (0, 1, 777, 4095):
        want = torch.roll(sig[start[b]].float(), int(shift[b, 0]), dims=-1)
        assert torch.equal(ssum[b], want) and torch.equal(strain[b], noise[row[b]].float() + want)
    snr0 = run(scale, torch.zeros_like(shift))[2]
    torch.testing.assert_close(snr, snr0, rtol=1e-6, atol=0)
    torch.testing.assert_close(snr, sig[start].float().square().sum((1, 2)).sqrt(), rtol=2e-6, atol=0)
    snr2 = run(scale * 1.25, shift)[2]
    torch.testing.assert_close(snr2, 1.25 * snr, rtol=1e-6, atol=0)
    # out-of-pool rows and nsig = 0 give pure noise / zeros, not a fault
    strain, ssum, snr = run(scale, shift, start=torch.full_like(start, 10 ** 9), row=torch.full_like(row, -1))
    assert float(strain.abs().max()) == 0.0 and float(snr.max()) == 0.0
    strain, _, _ = run(scale, shift, nsig=torch.zeros_like(nsig))
    assert torch.equal(strain[5], noise[row[5]].float())


@pytest.mark.gpu
def test_remix_argument_errors():
    L = _lib.lib()
    x = torch.zeros(64, device="cuda")
    with pytest.raises(ValueError):
        _lib.check(L.pf_remix_forward(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, x.data_ptr(), 0, 0, 0, 0, 0), "remix")
    with pytest.raises(ValueError):        # workspace too small
        i = torch.zeros(8, dtype=torch.int64, device="cuda")
        _lib.check(L.pf_remix_forward(0, 0, 0, 0, i.data_ptr(), i.data_ptr(), i.data_ptr(), x.data_ptr(),
                                      i.data_ptr(), 0, 0, 0, 1, x.data_ptr(), 0, 0, x.data_ptr(), 8, 0), "remix")
    assert L.pf_remix_forward(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) == 0     # empty batch
