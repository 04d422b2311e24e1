"""Deterministic weight / input recipes shared by ``make_golden.py`` (which runs
the reference's own classes in the build container) and by the tests (which
feed the same tensors to the oracle and to the HIP path).  Data only: nothing
here comes from the reference's sources."""
from __future__ import annotations

import math

import torch


def fill_state_dict(shapes: dict, seed: int) -> dict:
    """shapes: {key: torch.Size}.  Returns {key: float32 tensor}, filled in sorted
    key order from one generator: weights ~ N(0, 1/fan_in), 1-D ~ N(0, 0.02^2),
    LayerNorm gains ~ 1 + N(0, 0.1^2)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        t = torch.randn(shp, generator=g, dtype=torch.float32)
        if len(shp) >= 2:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            t = t / math.sqrt(fan_in)
        elif "norm" in k and k.endswith("weight"):
            t = 1.0 + 0.1 * t
        else:
            t = 0.02 * t
        out[k] = t
    return out


def strain_batch(batch: int, n_det: int, seed: int, t_len: int = 16384) -> torch.Tensor:
    """Whitened-noise-like strain N(0,1) with a loud linear chirp in event 0 and
    non-finite / out-of-range samples in event 1 (exercises lean_npe.py:207)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, n_det, t_len, generator=g, dtype=torch.float32)
    t = torch.arange(t_len, dtype=torch.float32) / 4096.0
    chirp = torch.sin(2 * math.pi * (30.0 * t + 20.0 * t * t)) * torch.exp(-((t - 2.5) / 0.6) ** 2)
    for d in range(n_det):
        x[0, d] += (6.0 - d) * torch.roll(chirp, 7 * d)
    if batch > 1:
        x[1, 0, 5] = float("nan")
        x[1, 0, 77] = float("inf")
        x[1, n_det - 1, 1000] = float("-inf")
        x[1, 0, 2000] = 250.0
        x[1, n_det - 1, 3000] = -1e4
    return x


def physical_params(batch: int, seed: int) -> torch.Tensor:
    """[batch, 11] physical parameters in PARAM_NAMES order, incl. out-of-range,
    zero and negative entries for the log dims and beyond-period angles."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(batch, 11, generator=g)
    lo = torch.tensor([1.0, 1.0, 40.0, 0.0, -math.pi / 2, 0.0, 0.0, 0.0, -1.6, 0.0, 0.0])
    hi = torch.tensor([105.0, 105.0, 2200.0, 2 * math.pi, math.pi / 2, math.pi, math.pi,
                       2 * math.pi, 1.6, 1.0, 1.0])
    p = lo + (hi - lo) * u
    p[0, 0] = 0.0          # log of clamp_min(1e-6)
    p[1, 1] = -3.0
    p[2, 2] = 5000.0       # above range -> clamp
    p[3, 3] = 7.5          # beyond 2 pi
    p[4, 8] = 4.0          # geocent_time past +1.6 (inside premerger range)
    p[5, 9] = 1.5
    p[6, 2] = 10.0         # below range
    return p


# ---- remix cache (noise.npy / signals.npy / params.npy / events.json) ---------------------------
REMIX_T = 16384


def remix_cache(out_dir: str, seed: int = 5, n_noise: int = 7, with_real_bank: bool = True) -> dict:
    """A tiny synthetic cache in the on-disk layout ``RemixDataset`` reads (remix_data.py:49-111):
    6 events with 1, 2, 3, 5, 1, 2 signals; parameters chosen so that every relabel guard is hit
    (distance near both guard limits, |t_c| beyond 1.45).  With ``with_real_bank`` also a
    real-noise bank: per detector two whitened segments (5 s) with their ASDs and a design ASD."""
    import json
    import os

    import numpy as np

    rng = np.random.default_rng(seed)
    os.makedirs(out_dir, exist_ok=True)
    counts = [1, 2, 3, 5, 1, 2]
    m = sum(counts)
    t = np.arange(REMIX_T, dtype=np.float64) / 4096.0
    noise = rng.standard_normal((n_noise, 3, REMIX_T)).astype(np.float16)
    signals = np.zeros((m, 3, REMIX_T), dtype=np.float16)
    params = np.zeros((m, 11), dtype=np.float32)
    for k in range(m):
        f0, t0 = 25.0 + 9.0 * k, 1.0 + 0.17 * k
        env = np.exp(-((t - t0) / 0.25) ** 2)
        for d in range(3):
            amp = (0.6 + 0.3 * d) * (1.0 + 0.2 * k)
            signals[k, d] = (amp * env * np.sin(2 * np.pi * (f0 * t + 30.0 * t * t) + 0.7 * d)).astype(np.float16)
        params[k] = [5.0 + 60.0 * rng.uniform(), 5.0 + 40.0 * rng.uniform(), 100.0 + 1500.0 * rng.uniform(),
                     6.28 * rng.uniform(), rng.uniform(-1.5, 1.5), 3.14 * rng.uniform(), 3.14 * rng.uniform(),
                     6.28 * rng.uniform(), rng.uniform(-1.2, 1.2), rng.uniform(), rng.uniform()]
    params[0, 2] = 46.0         # d/s < 45 for s > 1.03: rescale rejected
    params[1, 2] = 2090.0       # d/s > 2100 for s < 0.995: rescale rejected
    params[2, 8] = 1.5          # |t_c| >= 1.45: no time shift
    params[4, 8] = -1.47
    params[3, 2] = 0.5          # loudness uses max(d, 1)
    events, start = [], 0
    for c in counts:
        events.append([start, c])
        start += c
    np.save(os.path.join(out_dir, "noise.npy"), noise)
    np.save(os.path.join(out_dir, "signals.npy"), signals)
    np.save(os.path.join(out_dir, "params.npy"), params)
    with open(os.path.join(out_dir, "events.json"), "w") as fh:
        json.dump({"n_noise": n_noise, "n_signals": m, "events": events}, fh)
    if with_real_bank:
        bank = os.path.join(out_dir, "real_bank")
        os.makedirs(bank, exist_ok=True)
        nf = REMIX_T // 2 + 1
        f = np.fft.rfftfreq(REMIX_T, 1.0 / 4096.0)
        for di, d in enumerate(("H1", "L1", "V1")):
            design = (1e-23 * (1.0 + (60.0 / np.maximum(f, 5.0)) ** 4 + (f / 900.0) ** 2)).astype(np.float64)
            np.save(os.path.join(bank, f"design_asd_{d}.npy"), design)
            for j in range(2):
                seg = rng.standard_normal(REMIX_T + 4096 + 512 * j).astype(np.float16)
                wobble = 1.0 + 0.5 * np.sin(f / (70.0 + 25.0 * di + 11.0 * j)) ** 2
                asd = design * wobble
                if j == 1:
                    asd[:40] = 0.0              # exercises the max(asd, 1e-30) + clip(…, 1/50, 50) guards
                    asd[nf - 30:] *= 1e3
                np.save(os.path.join(bank, f"{d}_{j:02d}_strain.npy"), seg)
                np.save(os.path.join(bank, f"{d}_{j:02d}_asd.npy"), asd)
    return {"n_noise": n_noise, "n_signals": m, "events": events}


def pickle_chunks(data_dir: str, split: str = "train", seed: int = 17) -> None:
    """Two ``batch_*.pkl`` chunks in the v2 component-storage schema the dataset generator writes
    (dataset_generator.py:340-389): per sample ``detector_data[det] = {strain, snr, noise (f16),
    signals [f16, ...]}``, ``parameters`` (list of dicts), ``event_type``.  Covers: a pure-noise sample,
    a sample stored without components (skipped by the cache builder), 1-, 2- and 6-signal events (the
    last is truncated to 5), parameters missing a key, and loudness order different from storage order."""
    import os
    import pickle

    import numpy as np

    rng = np.random.default_rng(seed)
    names = ["mass_1", "mass_2", "luminosity_distance", "ra", "dec", "theta_jn", "psi", "phase",
             "geocent_time", "a1", "a2"]

    def pars(k, drop=None):
        p = {n: float(v) for n, v in zip(names, rng.uniform(0.1, 1.0, len(names)))}
        p["mass_1"], p["mass_2"] = 10.0 + 7.0 * ((k * 5) % 7), 8.0 + 3.0 * (k % 4)
        p["luminosity_distance"] = 200.0 + 450.0 * ((k * 3) % 5)
        p["event_type"] = "BBH"
        if drop:
            p.pop(drop)
        return p

    def sample(n_sig, kind="BBH", components=True, drop=None):
        dd = {}
        for det in ("H1", "L1", "V1"):
            e = {"strain": rng.standard_normal(REMIX_T).astype(np.float32), "snr": 9.0}
            if components:
                e["noise"] = rng.standard_normal(REMIX_T).astype(np.float16)
                e["signals"] = [(0.1 * rng.standard_normal(REMIX_T)).astype(np.float16) for _ in range(n_sig)]
            dd[det] = e
        plist = [pars(k, drop if k == 0 else None) for k in range(n_sig)]
        if kind == "noise":
            plist = [{"event_type": "noise"}]
        return {"detector_data": dd, "parameters": plist, "n_signals": n_sig, "event_type": kind}

    chunks = [[sample(2), sample(0, kind="noise"), sample(1, components=False)],
              [sample(6), sample(1, drop="a2"), sample(3)]]
    out = os.path.join(data_dir, split)
    os.makedirs(out, exist_ok=True)
    for i, samples in enumerate(chunks):
        with open(os.path.join(out, f"batch_{i:05d}.pkl"), "wb") as fh:
            pickle.dump({"samples": samples, "batch_id": i, "n_samples": len(samples)}, fh)
