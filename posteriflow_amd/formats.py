"""On-disk formats of the reference that surround the hot path: read / write
compatibility only, host code.

* dataset chunks ``{data_dir}/{split}/batch_*.pkl`` (component storage, dataset_generator.py:340-389)
  -> the memmap cache ``noise.npy / signals.npy / params.npy / events.json`` that ``RemixDataset``
  loads into HBM (layout of experiments/remix_data.py:49-111): ``build_memmap_cache``;
* training checkpoints ``best_model.pth`` (train_lean_npe.py:421-427) -> ``LeanNPE`` on the GPU:
  ``load_model`` (inference/pipeline.py:34-54);
* posterior result directories (``posterior_samples.npy``, ``posterior_log_prob.npy``,
  ``posterior_samples.csv``, ``result.json``; inference/result.py:242-288): ``save_posterior`` /
  ``load_posterior``.
"""
from __future__ import annotations

import glob
import json
import os
import pickle
from typing import Dict, Iterator, Optional, Tuple

import numpy as np
import torch

from .remix import MAX_SIGNALS, PARAM_NAMES, T_LEN

_DETS = ("H1", "L1", "V1")


def _loudness(p: dict) -> float:
    m1, m2 = p["mass_1"], p["mass_2"]
    return ((m1 * m2) ** 0.6 / (m1 + m2) ** 0.2) ** (5.0 / 6.0) / max(p["luminosity_distance"], 1.0)


def _component_samples(files) -> Iterator[Tuple[np.ndarray, list, list]]:
    """(noise [3, T] f16, signals (list of [3, T] f16, loudest first), parameter dicts in the same order)
    for every sample that stores its components; samples without a ``noise`` entry are skipped."""
    for path in files:
        with open(path, "rb") as fh:
            chunk = pickle.load(fh)
        for s in chunk["samples"]:
            dd = s["detector_data"]
            if "noise" not in dd[_DETS[0]]:
                continue
            noise = np.stack([dd[d]["noise"] for d in _DETS])
            if str(s.get("event_type")) == "noise":
                yield noise, [], []
                continue
            plist = list(s.get("parameters", []))[:MAX_SIGNALS]
            order = sorted(range(len(plist)), key=lambda k: _loudness(plist[k]), reverse=True)
            yield (noise, [np.stack([dd[d]["signals"][k] for d in _DETS]) for k in order],
                   [plist[k] for k in order])


def build_memmap_cache(data_dir: str, split: str, out_dir: str) -> Dict:
    """pickle chunks -> memmap cache; returns ``{"n_noise", "n_signals", "events"}`` (also events.json).
    noise.npy [n_noise, 3, T] f16 holds the noise of EVERY component sample (pure-noise ones included),
    signals.npy [n_signals, 3, T] f16 / params.npy [n_signals, 11] f32 the signals of each event back to
    back, loudest first, at most 5 per event; events = [first signal row, count] per signal event."""
    files = sorted(glob.glob(os.path.join(data_dir, split, "batch_*.pkl")))
    if not files:
        raise FileNotFoundError(f"no batch_*.pkl chunks under {os.path.join(data_dir, split)}")
    n_noise = n_sig = 0
    for _, sigs, _ in _component_samples(files):          # pass 1: sizes
        n_noise += 1
        n_sig += len(sigs)
    os.makedirs(out_dir, exist_ok=True)
    opened = lambda name, dtype, shape: np.lib.format.open_memmap(
        os.path.join(out_dir, name), mode="w+", dtype=dtype, shape=shape)
    noise_mm = opened("noise.npy", np.float16, (n_noise, 3, T_LEN))
    sig_mm = opened("signals.npy", np.float16, (n_sig, 3, T_LEN))
    par_mm = opened("params.npy", np.float32, (n_sig, len(PARAM_NAMES)))
    events, ni, si = [], 0, 0
    for noise, sigs, plist in _component_samples(files):   # pass 2: fill
        noise_mm[ni] = noise
        ni += 1
        if not sigs:
            continue
        events.append([si, len(sigs)])
        for sig, p in zip(sigs, plist):
            sig_mm[si] = sig
            par_mm[si] = [p.get(name, 0.0) for name in PARAM_NAMES]
            si += 1
    for mm in (noise_mm, sig_mm, par_mm):
        mm.flush()
    meta = {"n_noise": int(ni), "n_signals": int(si), "events": events}
    with open(os.path.join(out_dir, "events.json"), "w") as fh:
        json.dump(meta, fh)
    return meta


def load_model(path: str, device: str = "cuda"):
    """(LeanNPE on ``device`` in eval mode, meta) from a reference checkpoint: ``args`` select
    premerger / psd_cond / psd_bands / encoder_type exactly as pipeline.py:40-52 does."""
    from .npe import LeanNPE
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    a = ckpt.get("args", {}) or {}
    a = a if isinstance(a, dict) else vars(a)
    model = LeanNPE(premerger=a.get("premerger", False), psd_cond=a.get("psd_cond", False) or False,
                    psd_bands=a.get("psd_bands", 16), encoder_type=a.get("encoder_type", "conv"))
    model.load_state_dict(ckpt["model_state_dict"])
    model.to(device).eval()
    meta = {"model_path": str(path), "model_epoch": int(ckpt.get("epoch", -1)),
            "model_val_nll": float(ckpt.get("val_nll", float("nan"))),
            "premerger": bool(a.get("premerger", False)), "psd_cond": bool(model.psd_cond), "device": device}
    return model, meta


def save_posterior(outdir: str, samples, log_prob, param_names=PARAM_NAMES, trigger_gps: Optional[float] = None,
                   diagnostics: Optional[dict] = None, config: Optional[dict] = None) -> str:
    """Write draws in the reference's result-directory layout (result.py:242-288, the files a reader of
    that directory parses: the two .npy arrays, the CSV and result.json with the same top-level keys)."""
    os.makedirs(outdir, exist_ok=True)
    samples = np.asarray(torch.as_tensor(samples).cpu(), dtype=np.float64)
    log_prob = np.asarray(torch.as_tensor(log_prob).cpu(), dtype=np.float64)
    np.save(os.path.join(outdir, "posterior_samples.npy"), samples)
    np.save(os.path.join(outdir, "posterior_log_prob.npy"), log_prob)
    np.savetxt(os.path.join(outdir, "posterior_samples.csv"), np.column_stack([samples, log_prob]),
               delimiter=",", header=",".join(list(param_names) + ["log_prob"]), comments="")
    q = np.quantile(samples, [0.05, 0.5, 0.95], axis=0)
    summary = {n: {"median": float(q[1, j]), "lo90": float(q[0, j]), "hi90": float(q[2, j]),
                   "mean": float(samples[:, j].mean()), "std": float(samples[:, j].std())}
               for j, n in enumerate(param_names)}
    cov = np.cov(samples, rowvar=False)
    sd = np.sqrt(np.clip(np.diag(cov), 1e-300, None))
    config = dict(config or {})
    meta = {"param_names": list(param_names), "trigger_gps": trigger_gps, "truth": None, "summary": summary,
            "covariance": cov.tolist(), "correlation": (cov / np.outer(sd, sd)).tolist(),
            "diagnostics": diagnostics or {}, "config": config,
            "reproducibility": {"git_commit": None, **{k: config.get(k) for k in
                                                       ("model_path", "model_epoch", "model_val_nll", "premerger")}},
            "preprocessing": None}
    with open(os.path.join(outdir, "result.json"), "w") as fh:
        json.dump(meta, fh, indent=2, default=float)
    return outdir


def load_posterior(outdir: str) -> Dict:
    """Read a result directory written by the reference (or by ``save_posterior``)."""
    with open(os.path.join(outdir, "result.json")) as fh:
        meta = json.load(fh)
    return {"samples": np.load(os.path.join(outdir, "posterior_samples.npy")),
            "log_prob": np.load(os.path.join(outdir, "posterior_log_prob.npy")), **meta}
