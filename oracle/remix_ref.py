"""CPU restatement of the reference's training-example remix (``experiments/remix_data.py``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``): only ``tests/``, ``smoke()`` and ``bench.py``'s
``cpu_baseline`` may import this.  Pinned by ``tests/golden/remix.npz`` (outputs of the reference's own
``RemixDataset`` on the synthetic cache of ``tests/golden/recipe.py``).

The reference draws and applies in one pass (``RemixDataset.__getitem__``, remix_data.py:218-299).
Here the same work is split in two so that the GPU path can be checked piecewise:

* ``draw(i)``  -- every random decision of one example, consuming ``default_rng((seed, epoch, i))`` in
  the reference's order (``Decisions``);
* ``apply(i, dec)`` -- the deterministic algebra given those decisions: fp16 -> fp32 noise + scaled,
  circularly shifted signals (fp32, summed in storage order), optional re-colouring, detector
  dropout fill, network SNR over kept detectors, label relabel and loudness re-sort.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

T_LEN = 16384                # remix_data.py:39
MAX_SIGNALS = 5              # remix_data.py:38
N_PARAMS = 11
IDX_DIST, IDX_TIME = 2, 8    # remix_data.py:40-41 (positions in PARAM_NAMES)
DETS = ("H1", "L1", "V1")
KEEP_CONFIGS = ((0,), (1,), (2,), (0, 1), (0, 2), (1, 2))       # remix_data.py:154


def loudness(m1, m2, d):
    """remix_data.py:44-46 -- chirp-mass^(5/6) / max(d, 1); evaluated in the dtype handed in."""
    mc = (m1 * m2) ** 0.6 / (m1 + m2) ** 0.2
    return mc ** (5.0 / 6.0) / max(d, 1.0)


@dataclass
class Decisions:
    noise_idx: int = -1                                   # pool row, or -1 when real noise is used
    real: Optional[List[Tuple[int, int, bool]]] = None    # per detector (segment, offset, flip)
    scale: List[float] = field(default_factory=list)      # accepted amplitude factor per signal
    shift: List[int] = field(default_factory=list)        # circular shift (samples) per signal
    keep: Tuple[int, ...] = (0, 1, 2)
    refill: dict = field(default_factory=dict)            # det -> N(0,1) array, or (segment, offset, flip)


class RemixRef:
    def __init__(self, cache_dir, time_shift_max=0.1, dist_scale_range=(0.75, 1.33), sample_rate=4096,
                 remix=True, seed=0, real_noise_dir=None, real_noise_prob=0.0, recolor_clamp=50.0,
                 det_dropout=0.0, return_asd_bands=False, psd_bands=16):
        self.noise = np.load(os.path.join(cache_dir, "noise.npy"), mmap_mode="r")
        self.signals = np.load(os.path.join(cache_dir, "signals.npy"), mmap_mode="r")
        self.params = np.load(os.path.join(cache_dir, "params.npy"), mmap_mode="r")
        with open(os.path.join(cache_dir, "events.json")) as fh:
            meta = json.load(fh)
        self.events, self.n_noise = meta["events"], meta["n_noise"]
        self.max_shift = int(time_shift_max * sample_rate)            # :145
        self.s_lo, self.s_hi = dist_scale_range
        self.remix, self.seed, self.epoch = remix, seed, 0
        self.det_dropout = float(det_dropout)
        self.return_asd_bands, self.psd_bands = bool(return_asd_bands), int(psd_bands)
        if self.return_asd_bands:                                     # :162-171
            freqs = np.fft.rfftfreq(T_LEN, 1.0 / sample_rate)
            edges = np.geomspace(20.0, sample_rate / 2.0, self.psd_bands + 1)
            self.band_bins = []
            for lo, hi in zip(edges[:-1], edges[1:]):
                sel = np.nonzero((freqs >= lo) & (freqs < hi))[0]
                self.band_bins.append(sel if sel.size else np.array([np.argmin(np.abs(freqs - lo))]))
        self.real_prob = float(real_noise_prob)
        self.bank = None
        if real_noise_dir and self.real_prob > 0.0:                   # :174-196
            self.bank, self.filters = {}, {}
            for d in DETS:
                design = np.load(os.path.join(real_noise_dir, f"design_asd_{d}.npy"))
                segs, filts = [], []
                for name in sorted(os.listdir(real_noise_dir)):
                    if not (name.startswith(f"{d}_") and name.endswith("_strain.npy")):
                        continue
                    asd_path = os.path.join(real_noise_dir, name.replace("_strain", "_asd"))
                    if not os.path.exists(asd_path):
                        continue
                    asd = np.load(asd_path).astype(np.float32)
                    filts.append(np.clip(design / np.maximum(asd, 1e-30), 1.0 / recolor_clamp,
                                         recolor_clamp).astype(np.float32))
                    segs.append(np.load(os.path.join(real_noise_dir, name)))
                if not segs:
                    raise ValueError(f"real-noise bank incomplete under {real_noise_dir}")
                self.bank[d], self.filters[d] = segs, filts

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return len(self.events)

    # ---- random decisions, in the reference's draw order ------------------------------------------
    def _draw_crop(self, rng, det):                                   # :206-216
        k = int(rng.integers(len(self.bank[det])))
        i0 = int(rng.integers(0, self.bank[det][k].shape[0] - T_LEN))
        return k, i0, bool(rng.uniform() < 0.5)

    def draw(self, i) -> Decisions:
        start, nsig = self.events[i]
        rng = np.random.default_rng((self.seed, self.epoch, i))       # :220
        dec = Decisions()
        use_real = self.bank is not None and rng.uniform() < self.real_prob      # :222
        if use_real:
            dec.real = [self._draw_crop(rng, d) for d in DETS]
        else:
            dec.noise_idx = int(rng.integers(self.n_noise)) if self.remix else i % self.n_noise   # :226
        for k in range(nsig):                                         # :232-249
            par = self.params[start + k]
            s, ds = 1.0, 0
            if self.remix:
                s = float(rng.uniform(self.s_lo, self.s_hi))
                d_new = par[IDX_DIST] / s
                if not (45.0 < d_new < 2100.0):
                    s = 1.0
                if abs(par[IDX_TIME]) < 1.45 and self.max_shift > 0:
                    ds = int(rng.integers(-self.max_shift, self.max_shift + 1))
            dec.scale.append(s)
            dec.shift.append(ds)
        if self.remix and self.det_dropout > 0.0 and rng.uniform() < self.det_dropout:   # :262-279
            dec.keep = KEEP_CONFIGS[int(rng.integers(len(KEEP_CONFIGS)))]
            for di in range(3):
                if di in dec.keep:
                    continue
                dec.refill[di] = self._draw_crop(rng, DETS[di]) if use_real \
                    else rng.standard_normal(T_LEN).astype(np.float32)
        return dec

    # ---- deterministic algebra --------------------------------------------------------------------
    def _crop(self, det, k, i0, flip):
        c = self.bank[det][k][i0:i0 + T_LEN].astype(np.float32)
        return -c[::-1].copy() if flip else c                         # :213-214

    def relabel(self, i, dec):
        """labels [5, 11] after distance / time relabel and loudness re-sort (:238-249, :288-291)."""
        start, nsig = self.events[i]
        rows = []
        for k in range(nsig):
            par = self.params[start + k].copy()
            if dec.scale[k] != 1.0:
                par[IDX_DIST] = par[IDX_DIST] / dec.scale[k]
            if dec.shift[k] != 0:
                par[IDX_TIME] += dec.shift[k] / 4096.0
            rows.append(par)
        rows.sort(key=lambda p: loudness(p[0], p[1], p[IDX_DIST]), reverse=True)
        pv = np.zeros((MAX_SIGNALS, N_PARAMS), dtype=np.float32)
        for k, par in enumerate(rows):
            pv[k] = par
        return pv

    def signal_sum(self, i, dec):
        """fp32 sum over the event's signals of  s_k * roll(sig_k, ds_k)  in storage order (:232-251)."""
        start, nsig = self.events[i]
        acc = np.zeros((3, T_LEN), dtype=np.float32)
        for k in range(nsig):
            sig = self.signals[start + k].astype(np.float32)
            if self.remix:
                sig = sig * np.float32(dec.scale[k])
                if dec.shift[k] != 0:
                    sig = np.roll(sig, dec.shift[k], axis=-1)
            acc += sig
        return acc

    def apply(self, i, dec):
        _, nsig = self.events[i]
        use_real = dec.real is not None
        if use_real:
            strain = np.stack([self._crop(d, *dec.real[di]) for di, d in enumerate(DETS)])
            filts = [self.filters[d][dec.real[di][0]] for di, d in enumerate(DETS)]
        else:
            strain = self.noise[dec.noise_idx].astype(np.float32)
            filts = None
        ssum = self.signal_sum(i, dec)
        if filts is not None:                                         # :253-258
            for di in range(3):
                ssum[di] = np.fft.irfft(np.fft.rfft(ssum[di]) * filts[di], n=T_LEN).astype(np.float32)
        strain = strain + ssum
        for di, fill in dec.refill.items():                           # :266-279
            strain[di] = self._crop(DETS[di], *fill) if use_real else fill
        snr = np.float32(np.sqrt((ssum[list(dec.keep)] ** 2).sum()))  # :286
        out = [strain, self.relabel(i, dec), nsig, snr]
        if self.return_asd_bands:                                     # :301-311
            ab = np.zeros((3, self.psd_bands), dtype=np.float32)
            if filts is not None:
                for di in dec.keep:
                    logf = np.log(np.maximum(filts[di], 1e-30))
                    ab[di] = [float(logf[sel].mean()) for sel in self.band_bins]
            out.append(ab)
        return tuple(out)

    def item(self, i):
        return self.apply(i, self.draw(i))
