#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference's
own classes in the build container (``/root/reference`` is read-only and never
travels to the GPU box; only the vectors this script writes are committed).

What can be pinned here: the reference's OWN arithmetic --
``ParamScaler`` (lean_npe.py:48-114), ``PSDScaledNormal`` (flows.py:28-109),
``LeanStrainEncoder`` (lean_npe.py:131-252), ``CoherentEncoder``
(coherent_encoder.py:42-123), ``MaskedContextLinear`` masks (flows.py:112-183).

What cannot: anything that executes ``nflows`` (absent from the image, see
SURVEY.md 8c).  ``ahsd.models.flows`` does ``from nflows... import`` at module
level, so this script registers EMPTY placeholder modules under the nflows
names -- names only, every placeholder raises if it is instantiated or called
-- purely so that the module-level imports resolve.  No nflows arithmetic is
emulated and no golden vector depends on it.

Run:  python tests/golden/make_golden.py      (from the repo root)
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipe  # noqa: E402

REF_SRC = "/root/reference/src"


def _register_nflows_placeholders():
    class _Absent:
        def __init__(self, *a, **k):
            raise RuntimeError("nflows is not installed: placeholder only")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("nflows", distributions=mod("nflows.distributions"))
    mod("nflows.flows", Flow=_Absent)
    mod("nflows.distributions.normal", StandardNormal=_Absent)
    mod("nflows.transforms", CompositeTransform=_Absent)
    mod("nflows.transforms.autoregressive",
        MaskedPiecewiseRationalQuadraticAutoregressiveTransform=_Absent)
    mod("nflows.transforms.permutations", ReversePermutation=_Absent)


def main():
    _register_nflows_placeholders()
    sys.path.insert(0, REF_SRC)
    torch.set_num_threads(4)
    from ahsd.models.lean_npe import LeanStrainEncoder, ParamScaler, PARAM_NAMES
    from ahsd.models.coherent_encoder import CoherentEncoder
    from ahsd.models.flows import PSDScaledNormal, MaskedContextLinear

    out = {}

    # ---- ParamScaler -------------------------------------------------------
    p = recipe.physical_params(64, seed=11)
    for tag, pre in (("", False), ("_premerger", True)):
        sc = ParamScaler(PARAM_NAMES, premerger=pre)
        y = sc.normalize(p)
        g = torch.Generator().manual_seed(12)
        raw = (torch.rand(64, 11, generator=g) * 2 - 1) * 3.5   # beyond [-1,1]: wrap / clamp
        out[f"scaler_norm{tag}"] = y.numpy()
        out[f"scaler_denorm{tag}"] = sc.denormalize(raw).numpy()
        out[f"scaler_wrap{tag}"] = sc.wrap(raw).numpy()
        out[f"scaler_lo{tag}"] = sc.lo.numpy()
        out[f"scaler_hi{tag}"] = sc.hi.numpy()
    out["scaler_phys_in"] = p.numpy()
    out["scaler_raw_in"] = raw.numpy()

    # ---- PSDScaledNormal ---------------------------------------------------
    g = torch.Generator().manual_seed(21)
    z = torch.randn(32, 11, generator=g) * 2
    ls = torch.randn(32, 11, generator=g) * 0.3
    base = PSDScaledNormal(shape=[11])
    out["base_z"] = z.numpy()
    out["base_ls"] = ls.numpy()
    out["base_logp_zero"] = base.log_prob(z, torch.zeros_like(z)).numpy()
    out["base_logp_ls"] = base.log_prob(z, ls).numpy()

    # ---- MaskedContextLinear masks ------------------------------------------
    hidden_deg = torch.arange(256) % 10 + 1          # upstream rule for D = 11
    for full in (True, False):
        m = MaskedContextLinear(11, 24, hidden_deg, full_context=full)
        out[f"mcl_mask_full{int(full)}"] = m.mask.numpy().astype(np.uint8)

    # ---- LeanStrainEncoder (3 detectors, and single detector) ---------------
    enc_golden = {}
    for tag, ndet, psd in (("det3", 3, 0), ("det1", 1, 0), ("det3_psd", 3, 16)):
        enc = LeanStrainEncoder(n_detectors=ndet, psd_bands=psd).eval()
        shapes = {k: v.shape for k, v in enc.state_dict().items() if k != "pos.pe"}
        sd = recipe.fill_state_dict(shapes, seed=100 + ndet + psd)
        missing = enc.load_state_dict(sd, strict=False)
        assert missing.missing_keys == ["pos.pe"], missing
        strain = recipe.strain_batch(4, ndet, seed=7)
        asd = None
        if psd:
            g = torch.Generator().manual_seed(8)
            asd = torch.randn(4, ndet, psd, generator=g) * 0.3
        with torch.no_grad():
            feats, clean = enc._compute_feats(strain, asd)
            ctx = enc(strain, asd)
            w = enc.n_energy_windows
            win = clean[:, :, : (clean.shape[-1] // w) * w].reshape(4, ndet, w, -1)
            log_energy = torch.log((win ** 2).mean(dim=-1) + 1e-8)
            stem_out = enc.stem(torch.asinh(clean).reshape(4 * ndet, 1, -1))  # [B*D,192,61]
            # per-stage stem activations of sequence 0 (for the HIP conv kernels)
            h = torch.asinh(clean).reshape(4 * ndet, 1, -1)[:2]
            stages = []
            for layer in enc.stem:
                h = layer(h)
                if isinstance(layer, torch.nn.GELU):
                    stages.append(h)
        enc_golden[f"{tag}_ctx"] = ctx.numpy()
        enc_golden[f"{tag}_feats"] = feats.numpy()
        enc_golden[f"{tag}_log_energy"] = log_energy.numpy()
        enc_golden[f"{tag}_stem_out"] = stem_out[:2].numpy()
        enc_golden[f"{tag}_stage0"] = stages[0][:, :, ::16].numpy()   # subsampled in time
        enc_golden[f"{tag}_stage1"] = stages[1][:, :, ::4].numpy()
        enc_golden[f"{tag}_stage2"] = stages[2].numpy()
        if asd is not None:
            enc_golden[f"{tag}_asd"] = asd.numpy()

    # ---- CoherentEncoder ------------------------------------------------------
    enc = CoherentEncoder(context_dim=256, psd_bands=16).eval()
    skip = {"pos.pe", "Bsum", "bcount", "lags_norm"}
    shapes = {k: v.shape for k, v in enc.state_dict().items() if k not in skip}
    sd = recipe.fill_state_dict(shapes, seed=200)
    missing = enc.load_state_dict(sd, strict=False)
    assert sorted(missing.missing_keys) == sorted(skip), missing
    strain = recipe.strain_batch(4, 3, seed=9)
    g = torch.Generator().manual_seed(10)
    asd = torch.randn(4, 3, 16, generator=g) * 0.3
    with torch.no_grad():
        clean = torch.nan_to_num(strain, nan=0.0, posinf=100.0, neginf=-100.0).clamp(-100.0, 100.0)
        enc_golden["coh_rel"] = enc._geometry_rel(clean).numpy()
        enc_golden["coh_ctx"] = enc(strain, asd).numpy()
        enc_golden["coh_asd"] = asd.numpy()
        enc_golden["coh_band"] = np.array([enc.band_lo, enc.Nf, enc.maxlag], dtype=np.int64)

    np.savez_compressed(os.path.join(HERE, "own_code_small.npz"), **out)
    np.savez_compressed(os.path.join(HERE, "encoder.npz"),
                        **{k: v.astype(np.float32) if v.dtype == np.float32 else v
                           for k, v in enc_golden.items()})
    for f in ("own_code_small.npz", "encoder.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
