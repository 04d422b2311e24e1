"""On-disk formats: the cache builder against the reference's own builder (golden
digests from tests/golden/make_golden_remix.py), checkpoint and result-directory round trips."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import recipe                                                    # noqa: E402
from posteriflow_amd import formats                              # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "remix.npz"))


def test_cache_builder_is_byte_identical_to_the_references(tmp_path):
    recipe.pickle_chunks(str(tmp_path / "data"))
    meta = formats.build_memmap_cache(str(tmp_path / "data"), "train", str(tmp_path / "cache"))
    assert [meta["n_noise"], meta["n_signals"]] == GOLD["cache_counts"].tolist()
    assert meta["events"] == GOLD["cache_events"].tolist()
    with open(tmp_path / "cache" / "events.json") as fh:
        assert json.load(fh) == meta
    for name in ("noise", "signals", "params"):
        arr = np.load(tmp_path / "cache" / f"{name}.npy")
        assert list(arr.shape) == GOLD[f"cache_{name}_shape"].tolist()
        assert hashlib.sha256(arr.tobytes()).digest() == GOLD[f"cache_{name}_sha256"].tobytes(), name
    np.testing.assert_array_equal(np.load(tmp_path / "cache" / "params.npy"), GOLD["cache_params"])
    with pytest.raises(FileNotFoundError):
        formats.build_memmap_cache(str(tmp_path / "data"), "validation", str(tmp_path / "c2"))
    # the cache it wrote is what RemixDataset reads (plans are host/tensor logic: fine on the CPU)
    from posteriflow_amd.remix import RemixDataset
    ds = RemixDataset(str(tmp_path / "cache"), device="cpu", seed=1)
    plan = ds.reference_plan(range(len(ds)))
    assert plan.nsig.tolist() == [2, 5, 1, 3] and int(plan.noise_row.max()) < meta["n_noise"]


def test_result_directory_round_trip(tmp_path):
    g = np.random.default_rng(0)
    samples, logq = g.normal(size=(500, 11)), g.normal(size=500)
    out = formats.save_posterior(str(tmp_path / "res"), samples, logq, config={"model_path": "x.pth", "premerger": False})
    back = formats.load_posterior(out)
    np.testing.assert_array_equal(back["samples"], samples)
    np.testing.assert_array_equal(back["log_prob"], logq)
    # the keys the reference's result.json carries (result.py:255-277)
    assert set(back) >= {"param_names", "trigger_gps", "truth", "summary", "covariance", "correlation",
                         "diagnostics", "config", "reproducibility", "preprocessing"}
    assert back["param_names"][2] == "luminosity_distance" and len(back["covariance"]) == 11
    header = open(os.path.join(out, "posterior_samples.csv")).readline().strip().split(",")
    assert header == back["param_names"] + ["log_prob"]


@pytest.mark.gpu
def test_checkpoint_round_trip_gpu(tmp_path):
    from posteriflow_amd import LeanNPE
    from posteriflow_amd.train import checkpoint_dict
    torch.manual_seed(0)
    model = LeanNPE(premerger=True, psd_cond=True, psd_bands=8)
    path = str(tmp_path / "best_model.pth")
    torch.save(checkpoint_dict(model, epoch=7, val_nll=1.25,
                               args={"premerger": True, "psd_cond": True, "psd_bands": 8, "encoder_type": "conv"}), path)
    loaded, meta = formats.load_model(path)
    assert meta["model_epoch"] == 7 and meta["model_val_nll"] == 1.25 and meta["premerger"] and meta["psd_cond"]
    assert next(loaded.parameters()).is_cuda and not loaded.training
    for (k, a), (_, b) in zip(model.state_dict().items(), loaded.state_dict().items()):
        assert torch.equal(a, b.cpu()), k
    strain = recipe.strain_batch(2, 3, seed=3).cuda()
    asd = torch.zeros(2, 3, 8, device="cuda")
    assert torch.isfinite(loaded.sample_posterior(strain, n_samples=8, asd_bands=asd)).all()
