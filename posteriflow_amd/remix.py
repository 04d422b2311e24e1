"""Training-example remix on the MI355X (reference: ``experiments/remix_data.py``).

The reference assembles one example per ``Dataset.__getitem__`` call on the CPU (memmap reads, numpy
roll / scale / FFT) behind a ``DataLoader``.  Here the whole cache (fp16 noise and signal pools, the
memmap layout of remix_data.py:49-111) lives in HBM and a batch of examples is assembled by ONE HIP
kernel (``pf_remix_forward``) from a *plan*: the per-example random decisions plus the relabelled
parameters.  Two ways to get a plan:

* ``reference_plan(indices)`` -- host side, consumes ``numpy.random.default_rng((seed, epoch, i))`` in
  the reference's order, so an example is the reference's example for the same (seed, epoch, i)
  (bit-identical strain and labels; used by the parity tests and by ``__getitem__``);
* ``device_plan(indices)``    -- every draw and the relabel / loudness re-sort as device tensor ops from a
  ``torch.Generator`` (same distributions, different stream): the training path, no host work per
  example.

``RemixDataset`` keeps the reference's constructor arguments and tuple layout
``(strain[3,T], params[5,11], n_signals, net_snr[, asd_bands])``; ``batch()`` is the same with a leading
batch dimension and device tensors.  The optional real-noise path (crops of a real-noise bank and the
re-colouring ``irfft(rfft(sig) * design/measured)``, remix_data.py:253-258) uses hipFFT through
``torch.fft`` on the signal sum the kernel returns.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib

T_LEN = 16384
MAX_SIGNALS = 5
PARAM_NAMES = ["mass_1", "mass_2", "luminosity_distance", "ra", "dec", "theta_jn", "psi", "phase",
               "geocent_time", "a1", "a2"]
IDX_DIST, IDX_TIME = PARAM_NAMES.index("luminosity_distance"), PARAM_NAMES.index("geocent_time")
_DETS = ("H1", "L1", "V1")
_KEEP = ((0,), (1,), (2,), (0, 1), (0, 2), (1, 2))        # proper subsets of detectors to keep (RD:154)
_D_LO, _D_HI, _T_EDGE = 45.0, 2100.0, 1.45                 # relabel guards (RD:238, :246)


def _loudness_np(p):
    """chirp-mass^(5/6) / max(d_L, 1) on one fp32 parameter row (RD:44-46)."""
    m1, m2, d = p[0], p[1], p[IDX_DIST]
    mc = (m1 * m2) ** 0.6 / (m1 + m2) ** 0.2
    return mc ** (5.0 / 6.0) / max(d, 1.0)


@dataclass
class RemixPlan:
    """Decisions for a batch of examples (device tensors unless noted)."""
    noise_row: torch.Tensor                   # int64 [B]; -1 where real noise is used
    sig_start: torch.Tensor                   # int64 [B]
    nsig: torch.Tensor                        # int32 [B]
    scale: torch.Tensor                       # f32 [B,5]
    shift: torch.Tensor                       # int32 [B,5]
    labels: torch.Tensor                      # f32 [B,5,11] relabelled + loudness-sorted
    keep: torch.Tensor                        # bool [B,3]
    fill_row: Optional[torch.Tensor] = None   # int32 [B,3] row of ``fill`` or -1
    fill: Optional[torch.Tensor] = None       # f32 [n,T]
    real: Optional[dict] = None               # real-noise decisions: mask[B], seg/off/flip [B,3], re_* for refills


class RemixDataset:
    def __init__(self, cache_dir: str, time_shift_max: float = 0.1, dist_scale_range: tuple = (0.75, 1.33),
                 sample_rate: int = 4096, remix: bool = True, seed: int = 0,
                 real_noise_dir: Optional[str] = None, real_noise_prob: float = 0.0,
                 recolor_clamp: float = 50.0, det_dropout: float = 0.0, return_asd_bands: bool = False,
                 psd_bands: int = 16, device="cuda"):
        # plans (decisions + labels) can be drawn anywhere; assemble() needs the GPU and raises off it
        self.device = torch.device(device)
        with open(os.path.join(cache_dir, "events.json")) as fh:
            meta = json.load(fh)
        self.n_noise = int(meta["n_noise"])
        ev = np.asarray(meta["events"], dtype=np.int64).reshape(-1, 2)
        self._ev_host = ev
        self.ev_start = torch.from_numpy(ev[:, 0].copy()).to(self.device)
        self.ev_count = torch.from_numpy(ev[:, 1].copy()).to(self.device)
        # the pools stay in HBM as stored (fp16): 98 KB per row
        self.noise = self._to_device(np.load(os.path.join(cache_dir, "noise.npy"), mmap_mode="r"))
        self.signals = self._to_device(np.load(os.path.join(cache_dir, "signals.npy"), mmap_mode="r"))
        self._params_host = np.array(np.load(os.path.join(cache_dir, "params.npy"), mmap_mode="r"), dtype=np.float32)
        self.params = torch.from_numpy(self._params_host).to(self.device)
        if self.noise.shape[1:] != (3, T_LEN) or self.signals.shape[1:] != (3, T_LEN):
            raise ValueError("cache pools must be [rows, 3, 16384] (remix_data.py:76-79)")
        self.shift_max = int(time_shift_max * sample_rate)
        self.s_lo, self.s_hi = (float(v) for v in dist_scale_range)
        self.remix, self.seed, self.epoch = bool(remix), seed, 0
        self.det_dropout = float(det_dropout)
        self.return_asd_bands, self.psd_bands = bool(return_asd_bands), int(psd_bands)
        self.real_noise_prob = float(real_noise_prob)
        self.bank = None
        if real_noise_dir and self.real_noise_prob > 0.0:
            self._load_bank(real_noise_dir, float(recolor_clamp), sample_rate)
        self._ws = None

    # ---- loading ----------------------------------------------------------------------------------
    def _to_device(self, mm, rows_per_copy: int = 4096):
        """memmap -> HBM in slices (a 22 GB pool must not be read into host RAM whole)."""
        out = torch.empty(mm.shape, dtype=torch.float16, device=self.device)
        for i in range(0, mm.shape[0], rows_per_copy):
            out[i:i + rows_per_copy].copy_(torch.from_numpy(np.array(mm[i:i + rows_per_copy])))
        return out

    def _load_bank(self, bank_dir, clamp, sample_rate):
        """per detector: all segments back to back in one fp16 tensor + their recolour filters (RD:174-196)."""
        freqs = np.fft.rfftfreq(T_LEN, 1.0 / sample_rate)
        edges = np.geomspace(20.0, sample_rate / 2.0, self.psd_bands + 1)
        bins = []
        for lo, hi in zip(edges[:-1], edges[1:]):
            sel = np.nonzero((freqs >= lo) & (freqs < hi))[0]
            bins.append(sel if sel.size else np.array([np.argmin(np.abs(freqs - lo))]))
        self.bank = []
        for d in _DETS:
            design = np.load(os.path.join(bank_dir, f"design_asd_{d}.npy"))
            segs, filts = [], []
            for name in sorted(os.listdir(bank_dir)):
                if not (name.startswith(f"{d}_") and name.endswith("_strain.npy")):
                    continue
                asd_path = os.path.join(bank_dir, name.replace("_strain", "_asd"))
                if not os.path.exists(asd_path):
                    continue
                asd = np.load(asd_path).astype(np.float32)
                filts.append(np.clip(design / np.maximum(asd, 1e-30), 1.0 / clamp, clamp).astype(np.float32))
                segs.append(np.load(os.path.join(bank_dir, name)).astype(np.float16))
            if not segs:
                raise ValueError(f"real-noise bank incomplete under {bank_dir}: no segments for {d}")
            lens = np.array([s.shape[0] for s in segs], dtype=np.int64)
            base = np.concatenate([[0], np.cumsum(lens)[:-1]])
            logf = [np.log(np.maximum(f, 1e-30)) for f in filts]
            bands = np.array([[float(lf[sel].mean()) for sel in bins] for lf in logf], dtype=np.float32)
            self.bank.append(dict(
                data=torch.from_numpy(np.concatenate(segs)).to(self.device),
                base=torch.from_numpy(base).to(self.device), lens=torch.from_numpy(lens).to(self.device),
                lens_host=lens, filt=torch.from_numpy(np.stack(filts)).to(self.device),
                bands=torch.from_numpy(bands).to(self.device)))

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __len__(self):
        return self._ev_host.shape[0]

    # ---- plans --------------------------------------------------------------------------------------
    def reference_plan(self, indices) -> RemixPlan:
        """The reference's own decisions for examples ``indices`` of the current epoch: the generator
        ``default_rng((seed, epoch, i))`` is consumed in the order of RD:220-279."""
        idx = [int(i) for i in indices]
        n = len(idx)
        noise_row = np.full(n, -1, np.int64)
        scale = np.ones((n, MAX_SIGNALS), np.float32)
        shift = np.zeros((n, MAX_SIGNALS), np.int32)
        labels = np.zeros((n, MAX_SIGNALS, len(PARAM_NAMES)), np.float32)
        keep = np.ones((n, 3), bool)
        fill_row = np.full((n, 3), -1, np.int32)
        fills = []
        real = dict(mask=np.zeros(n, bool), seg=np.zeros((n, 3), np.int64), off=np.zeros((n, 3), np.int64),
                    flip=np.zeros((n, 3), bool), re_seg=np.zeros((n, 3), np.int64),
                    re_off=np.zeros((n, 3), np.int64), re_flip=np.zeros((n, 3), bool))

        def crop(rng, d):
            k = int(rng.integers(len(self.bank[d]["lens_host"])))
            return k, int(rng.integers(0, int(self.bank[d]["lens_host"][k]) - T_LEN)), bool(rng.uniform() < 0.5)

        for b, i in enumerate(idx):
            start, count = (int(v) for v in self._ev_host[i])
            rng = np.random.default_rng((self.seed, self.epoch, i))
            is_real = self.bank is not None and rng.uniform() < self.real_noise_prob
            if is_real:
                real["mask"][b] = True
                for d in range(3):
                    real["seg"][b, d], real["off"][b, d], real["flip"][b, d] = crop(rng, d)
            else:
                noise_row[b] = int(rng.integers(self.n_noise)) if self.remix else i % self.n_noise
            rows = []
            for k in range(count):
                par = self._params_host[start + k].copy()
                if self.remix:
                    s = float(rng.uniform(self.s_lo, self.s_hi))
                    d_new = par[IDX_DIST] / s
                    if _D_LO < d_new < _D_HI:
                        scale[b, k], par[IDX_DIST] = s, d_new
                    if abs(par[IDX_TIME]) < _T_EDGE and self.shift_max > 0:
                        ds = int(rng.integers(-self.shift_max, self.shift_max + 1))
                        shift[b, k] = ds
                        if ds != 0:
                            par[IDX_TIME] += ds / 4096.0
                rows.append(par)
            rows.sort(key=_loudness_np, reverse=True)
            for k, par in enumerate(rows):
                labels[b, k] = par
            if self.remix and self.det_dropout > 0.0 and rng.uniform() < self.det_dropout:
                kept = _KEEP[int(rng.integers(len(_KEEP)))]
                for d in range(3):
                    if d in kept:
                        continue
                    keep[b, d] = False
                    if is_real:
                        real["re_seg"][b, d], real["re_off"][b, d], real["re_flip"][b, d] = crop(rng, d)
                    else:
                        fill_row[b, d] = len(fills)
                        fills.append(rng.standard_normal(T_LEN).astype(np.float32))
        dev = self.device
        ev = self._ev_host[idx] if n else np.zeros((0, 2), np.int64)
        return RemixPlan(
            noise_row=torch.from_numpy(noise_row).to(dev), sig_start=torch.from_numpy(ev[:, 0].copy()).to(dev),
            nsig=torch.from_numpy(ev[:, 1].astype(np.int32)).to(dev), scale=torch.from_numpy(scale).to(dev),
            shift=torch.from_numpy(shift).to(dev), labels=torch.from_numpy(labels).to(dev),
            keep=torch.from_numpy(keep).to(dev),
            fill_row=torch.from_numpy(fill_row).to(dev) if fills else None,
            fill=torch.from_numpy(np.stack(fills)).to(dev) if fills else None,
            real={k: torch.from_numpy(v).to(dev) for k, v in real.items()} if real["mask"].any() else None)

    def device_plan(self, indices, generator: Optional[torch.Generator] = None) -> RemixPlan:
        """Same distributions as ``reference_plan`` with every draw, the relabel and the loudness re-sort
        done as device tensor ops (no host work per example, no sync)."""
        dev = self.device
        idx = torch.as_tensor(indices, dtype=torch.int64, device=dev)
        n = idx.numel()
        g = generator
        rand = lambda *shape: torch.rand(*shape, device=dev, generator=g)
        randint = lambda hi, *shape: torch.randint(0, hi, shape, device=dev, generator=g)
        start, count = self.ev_start[idx], self.ev_count[idx]
        ks = torch.arange(MAX_SIGNALS, device=dev)
        valid = ks[None, :] < count[:, None]
        rows = (start[:, None] + ks[None, :]).clamp_(max=self.params.shape[0] - 1)
        par = torch.where(valid[..., None], self.params[rows], torch.zeros((), device=dev))
        scale = torch.ones(n, MAX_SIGNALS, device=dev)
        shift = torch.zeros(n, MAX_SIGNALS, dtype=torch.int32, device=dev)
        is_real = torch.zeros(n, dtype=torch.bool, device=dev)
        if self.bank is not None:
            is_real = rand(n) < self.real_noise_prob
        noise_row = randint(self.n_noise, n) if self.remix else idx % self.n_noise
        noise_row = torch.where(is_real, torch.full_like(noise_row, -1), noise_row)
        if self.remix:
            s = self.s_lo + (self.s_hi - self.s_lo) * rand(n, MAX_SIGNALS)
            d_new = par[..., IDX_DIST] / s
            ok = valid & (d_new > _D_LO) & (d_new < _D_HI)
            scale = torch.where(ok, s, scale)
            par[..., IDX_DIST] = torch.where(ok, d_new, par[..., IDX_DIST])
            if self.shift_max > 0:
                ds = randint(2 * self.shift_max + 1, n, MAX_SIGNALS) - self.shift_max
                ds = torch.where(valid & (par[..., IDX_TIME].abs() < _T_EDGE), ds, torch.zeros_like(ds))
                shift = ds.to(torch.int32)
                par[..., IDX_TIME] = par[..., IDX_TIME] + ds.to(torch.float32) / 4096.0
        m1, m2, d = par[..., 0], par[..., 1], par[..., IDX_DIST]
        loud = ((m1 * m2) ** 0.6 / (m1 + m2) ** 0.2) ** (5.0 / 6.0) / d.clamp_min(1.0)
        loud = torch.where(valid, loud, torch.full_like(loud, -float("inf")))
        order = torch.sort(loud, dim=1, descending=True, stable=True).indices
        labels = torch.gather(par, 1, order[..., None].expand(-1, -1, par.shape[-1]))
        keep = torch.ones(n, 3, dtype=torch.bool, device=dev)
        fill_row = fill = None
        real = None
        if self.remix and self.det_dropout > 0.0:
            table = torch.tensor([[d in c for d in range(3)] for c in _KEEP], device=dev)
            drop = rand(n) < self.det_dropout
            keep = torch.where(drop[:, None], table[randint(len(_KEEP), n)], keep)
            gone = ~keep & ~is_real[:, None]
            fill_row = (torch.cumsum(gone.reshape(-1).to(torch.int32), 0) - 1).to(torch.int32).reshape(n, 3)
            fill_row = torch.where(gone, fill_row, torch.full_like(fill_row, -1))
            # upper bound on the number of dropped detectors without a device->host sync
            fill = torch.randn(2 * n, T_LEN, device=dev, generator=g)
        if self.bank is not None:
            real = dict(mask=is_real)
            for tag in ("", "re_"):
                seg = torch.stack([randint(len(bk["lens_host"]), n) for bk in self.bank], 1)
                span = torch.stack([bk["lens"][seg[:, d]] - T_LEN for d, bk in enumerate(self.bank)], 1)
                real[tag + "seg"] = seg
                real[tag + "off"] = (rand(n, 3) * span).long().clamp_(max=(span - 1).clamp_min(0))
                real[tag + "flip"] = rand(n, 3) < 0.5
        return RemixPlan(noise_row=noise_row, sig_start=start, nsig=count.to(torch.int32), scale=scale,
                         shift=shift, labels=labels, keep=keep, fill_row=fill_row, fill=fill, real=real)

    # ---- assembly (HIP) -----------------------------------------------------------------------------
    def _crops(self, which, seg, off, flip):
        """[n, T] fp32 real-noise crops of detector ``which``; a flipped crop is time-reversed and negated."""
        bk = self.bank[which]
        t = torch.arange(T_LEN, device=self.device)
        pos = torch.where(flip[:, None], T_LEN - 1 - t[None, :], t[None, :])
        x = bk["data"][bk["base"][seg][:, None] + off[:, None] + pos].float()
        return torch.where(flip[:, None], -x, x)

    def assemble(self, plan: RemixPlan):
        """(strain [B,3,T], labels [B,5,11], n_signals [B], net_snr [B][, asd_bands [B,3,bands]])."""
        dev = self.device
        if dev.type != "cuda":
            raise _lib.PfError(f"RemixDataset.assemble runs on the MI355X only (pools on {dev}); no CPU path")
        L = _lib.lib()
        n = plan.nsig.numel()
        strain = torch.empty(n, 3, T_LEN, dtype=torch.float32, device=dev)
        snr = torch.empty(n, dtype=torch.float32, device=dev)
        ssum = torch.empty_like(strain) if plan.real is not None else None
        need = L.pf_remix_workspace_bytes(n)
        if self._ws is None or self._ws.numel() * 8 < need:
            self._ws = torch.empty(max(need // 8, 1), dtype=torch.float64, device=dev)
        ptr = lambda t: 0 if t is None else t.data_ptr()
        tensors = [plan.noise_row.contiguous(), plan.sig_start.contiguous(), plan.nsig.contiguous(),
                   plan.scale.contiguous(), plan.shift.contiguous()]
        fill_row = None if plan.fill_row is None else plan.fill_row.contiguous()
        fill = None if plan.fill is None else plan.fill.contiguous()
        _lib.check(L.pf_remix_forward(
            self.noise.data_ptr(), self.noise.shape[0], self.signals.data_ptr(), self.signals.shape[0],
            *[t.data_ptr() for t in tensors], ptr(fill_row), ptr(fill), 0 if fill is None else fill.shape[0],
            n, strain.data_ptr(), ptr(ssum), snr.data_ptr(), self._ws.data_ptr(), self._ws.numel() * 8,
            torch.cuda.current_stream(dev).cuda_stream), "pf_remix_forward")
        asd = torch.zeros(n, 3, self.psd_bands, device=dev) if self.return_asd_bands else None
        if plan.real is not None:
            r = plan.real
            m = r["mask"]
            filt = torch.stack([bk["filt"][r["seg"][:, d]] for d, bk in enumerate(self.bank)], 1)     # [B,3,nf]
            recol = torch.fft.irfft(torch.fft.rfft(ssum) * filt, n=T_LEN)                              # RD:253-258
            crops = torch.stack([self._crops(d, r["seg"][:, d], r["off"][:, d], r["flip"][:, d]) for d in range(3)], 1)
            refill = torch.stack([self._crops(d, r["re_seg"][:, d], r["re_off"][:, d], r["re_flip"][:, d])
                                  for d in range(3)], 1)
            real_strain = torch.where(plan.keep[..., None], crops + recol, refill)
            strain = torch.where(m[:, None, None], real_strain, strain)
            real_snr = (recol.square().sum(-1) * plan.keep).sum(-1).sqrt()
            snr = torch.where(m, real_snr, snr)
            if asd is not None:
                bands = torch.stack([bk["bands"][r["seg"][:, d]] for d, bk in enumerate(self.bank)], 1)
                asd = torch.where((m[:, None] & plan.keep)[..., None], bands, asd)
        out = (strain, plan.labels, plan.nsig.to(torch.int64), snr)
        return out + (asd,) if asd is not None else out

    def batch(self, indices, exact: bool = False, generator: Optional[torch.Generator] = None):
        """A batch of remixed examples; ``exact=True`` reproduces the reference's examples for
        (seed, epoch, index), otherwise the decisions are drawn on the device."""
        plan = self.reference_plan(indices) if exact else self.device_plan(indices, generator)
        return self.assemble(plan)

    def __getitem__(self, i):
        return tuple(t[0] for t in self.batch([i], exact=True))


def synthetic_dataset(device="cuda", n_noise: int = 512, n_events: int = 512, seed: int = 0, **kw) -> RemixDataset:
    """A synthetic memmap cache in the reference's format (unit white noise rows, weak random "signals", 1-3 signals per
    event, labels inside the prior box of lean_npe.py:54-66) written to a temporary directory and loaded as a
    RemixDataset with its pools resident in HBM: the config-4 workload of bench.py / the tests (there is no network for
    the real 22 GB dataset).  ``n_events`` is also available as ``ds.n_events``."""
    import tempfile
    tmp = tempfile.mkdtemp(prefix="pf_remix_")
    rng = np.random.default_rng(seed)
    counts = rng.integers(1, 4, n_events)
    m = int(counts.sum())
    np.save(os.path.join(tmp, "noise.npy"), rng.standard_normal((n_noise, 3, T_LEN), dtype=np.float32).astype(np.float16))
    np.save(os.path.join(tmp, "signals.npy"),
            (0.1 * rng.standard_normal((m, 3, T_LEN), dtype=np.float32)).astype(np.float16))
    lo = np.array([5, 5, 100, 0, -1.5, 0, 0, 0, -1.2, 0, 0], np.float32)
    hi = np.array([80, 60, 1500, 6.2, 1.5, 3.1, 3.1, 6.2, 1.2, 1, 1], np.float32)
    np.save(os.path.join(tmp, "params.npy"), lo + (hi - lo) * rng.random((m, 11), dtype=np.float32))
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    with open(os.path.join(tmp, "events.json"), "w") as fh:
        json.dump({"n_noise": n_noise, "n_signals": m, "events": [[int(a), int(b)] for a, b in zip(starts, counts)]}, fh)
    ds = RemixDataset(tmp, seed=seed, device=device, **kw)
    ds.n_events = n_events
    return ds
