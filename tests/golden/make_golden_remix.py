#!/usr/bin/env python3
"""Golden vectors for the remix augmentation: RUN the reference's own
``RemixDataset`` (``/root/reference/experiments/remix_data.py`` -- pure numpy/torch, importable) on the
tiny synthetic cache of ``recipe.remix_cache`` and store what it returned: labels, signal counts,
network SNR, asd_bands, and the strain both sub-sampled (every 29th sample) and as float64 channel
sums.  Only these vectors are committed; the reference never travels.

Run:  python tests/golden/make_golden_remix.py      (from the repo root)
"""
from __future__ import annotations

import importlib.util
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipe  # noqa: E402

REF_FILE = "/root/reference/experiments/remix_data.py"

# (tag, constructor kwargs, epoch); "bank" is replaced by the real-noise directory of the cache
CASES = [
    ("plain", dict(seed=3), 0),
    ("epoch2", dict(seed=3), 2),
    ("noremix", dict(seed=3, remix=False), 0),
    ("noshift", dict(seed=4, time_shift_max=0.0, dist_scale_range=(0.5, 2.0)), 1),
    ("dropout", dict(seed=9, det_dropout=0.7, return_asd_bands=True, psd_bands=16), 0),
    ("real", dict(seed=6, real_noise_dir="bank", real_noise_prob=0.6, det_dropout=0.5,
                  return_asd_bands=True, psd_bands=8), 1),
]
STRIDE = 29


def main():
    spec = importlib.util.spec_from_file_location("ref_remix_data", REF_FILE)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        meta = recipe.remix_cache(tmp)
        n_events = len(meta["events"])
        for tag, kw, epoch in CASES:
            kw = dict(kw)
            if kw.get("real_noise_dir") == "bank":
                kw["real_noise_dir"] = os.path.join(tmp, "real_bank")
            ds = ref.RemixDataset(tmp, **kw)
            ds.set_epoch(epoch)
            assert len(ds) == n_events
            rows = [ds[i] for i in range(n_events)]
            strain = np.stack([r[0].numpy() for r in rows])
            out[f"{tag}_strain_sub"] = strain[:, :, ::STRIDE].copy()
            out[f"{tag}_strain_sum"] = strain.astype(np.float64).sum(axis=-1)
            out[f"{tag}_strain_abs"] = np.abs(strain.astype(np.float64)).sum(axis=-1)
            out[f"{tag}_pv"] = np.stack([r[1].numpy() for r in rows])
            out[f"{tag}_nsig"] = np.array([int(r[2]) for r in rows], dtype=np.int64)
            out[f"{tag}_snr"] = np.array([float(r[3]) for r in rows], dtype=np.float32)
            if kw.get("return_asd_bands"):
                out[f"{tag}_asd_bands"] = np.stack([r[4].numpy() for r in rows])
    # ---- cache builder: the reference's build_memmap_cache on synthetic pickle chunks ----------------
    import hashlib
    import json
    with tempfile.TemporaryDirectory() as tmp:
        recipe.pickle_chunks(os.path.join(tmp, "data"))
        meta = ref.build_memmap_cache(os.path.join(tmp, "data"), "train", os.path.join(tmp, "cache"))
        for name in ("noise", "signals", "params"):
            arr = np.load(os.path.join(tmp, "cache", f"{name}.npy"))
            out[f"cache_{name}_shape"] = np.array(arr.shape)
            out[f"cache_{name}_sha256"] = np.frombuffer(hashlib.sha256(arr.tobytes()).digest(), dtype=np.uint8)
        out["cache_params"] = np.load(os.path.join(tmp, "cache", "params.npy"))
        out["cache_events"] = np.array(meta["events"])
        out["cache_counts"] = np.array([meta["n_noise"], meta["n_signals"]])
        with open(os.path.join(tmp, "cache", "events.json")) as fh:
            assert json.load(fh) == meta
    path = os.path.join(HERE, "remix.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.startswith("plain")})


if __name__ == "__main__":
    main()
