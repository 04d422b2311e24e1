"""oracle/remix_ref.py against the golden vectors produced by the reference's own RemixDataset
(tests/golden/make_golden_remix.py): every case, every output."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import recipe                                               # noqa: E402
from make_golden_remix import CASES, STRIDE                 # noqa: E402
from oracle.remix_ref import RemixRef                       # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "remix.npz"))


@pytest.fixture(scope="module")
def cache(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("remix_cache"))
    recipe.remix_cache(d)
    return d


@pytest.mark.parametrize("tag,kw,epoch", CASES, ids=[c[0] for c in CASES])
def test_remix_restatement_matches_reference(cache, tag, kw, epoch):
    kw = dict(kw)
    if kw.get("real_noise_dir") == "bank":
        kw["real_noise_dir"] = os.path.join(cache, "real_bank")
    ds = RemixRef(cache, **kw)
    ds.set_epoch(epoch)
    rows = [ds.item(i) for i in range(len(ds))]
    strain = np.stack([r[0] for r in rows])
    # the noise + signal arithmetic is fp32 element-wise in a fixed order: bit-exact
    np.testing.assert_array_equal(strain[:, :, ::STRIDE], GOLD[f"{tag}_strain_sub"])
    np.testing.assert_allclose(strain.astype(np.float64).sum(-1), GOLD[f"{tag}_strain_sum"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.abs(strain.astype(np.float64)).sum(-1), GOLD[f"{tag}_strain_abs"], rtol=1e-13)
    np.testing.assert_array_equal(np.stack([r[1] for r in rows]), GOLD[f"{tag}_pv"])
    np.testing.assert_array_equal(np.array([r[2] for r in rows]), GOLD[f"{tag}_nsig"])
    np.testing.assert_array_equal(np.array([r[3] for r in rows], dtype=np.float32), GOLD[f"{tag}_snr"])
    if kw.get("return_asd_bands"):
        np.testing.assert_array_equal(np.stack([r[4] for r in rows]), GOLD[f"{tag}_asd_bands"])


def test_cases_exercise_every_branch(cache):
    """the fixture really hits: rejected rescale, suppressed shift, dropout, real noise + recolour."""
    ds = RemixRef(cache, seed=3)
    decs = [ds.draw(i) for i in range(len(ds))]
    assert any(s == 1.0 for d in decs for s in d.scale) and any(s != 1.0 for d in decs for s in d.scale)
    assert any(x == 0 for d in decs for x in d.shift) and any(x != 0 for d in decs for x in d.shift)
    ds = RemixRef(cache, seed=6, real_noise_dir=os.path.join(cache, "real_bank"), real_noise_prob=0.6,
                  det_dropout=0.5)
    ds.set_epoch(1)
    decs = [ds.draw(i) for i in range(len(ds))]
    assert any(d.real is not None for d in decs) and any(d.real is None for d in decs)
    assert any(len(d.keep) < 3 for d in decs)
