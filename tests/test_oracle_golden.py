"""Pin the oracle's restatement of the reference's OWN code against the golden
vectors that tests/golden/make_golden.py produced by running the reference
(ParamScaler, PSDScaledNormal, LeanStrainEncoder, CoherentEncoder, masks)."""
import numpy as np
import pytest
import torch

import recipe
from oracle import lean_ref
from oracle.flow_ref import PSDScaledNormalRef

RTOL, ATOL = 1e-5, 1e-6   # 1e-5 rel / 1e-6 abs fp32


def close(a, b, rtol=RTOL, atol=ATOL):
    a = a.numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("pre", [False, True])
def test_param_scaler(golden_small, pre):
    tag = "_premerger" if pre else ""
    sc = lean_ref.ParamScalerRef(premerger=pre)
    p = torch.from_numpy(golden_small["scaler_phys_in"])
    raw = torch.from_numpy(golden_small["scaler_raw_in"])
    close(sc.lo, golden_small[f"scaler_lo{tag}"])
    close(sc.hi, golden_small[f"scaler_hi{tag}"])
    close(sc.normalize(p), golden_small[f"scaler_norm{tag}"])
    close(sc.denormalize(raw), golden_small[f"scaler_denorm{tag}"])
    close(sc.wrap(raw), golden_small[f"scaler_wrap{tag}"])


def test_psd_scaled_normal(golden_small):
    base = PSDScaledNormalRef([11])
    z = torch.from_numpy(golden_small["base_z"])
    ls = torch.from_numpy(golden_small["base_ls"])
    close(base.log_prob(z, torch.zeros_like(z)), golden_small["base_logp_zero"])
    close(base.log_prob(z, ls), golden_small["base_logp_ls"])
    with pytest.raises(ValueError):
        base.log_prob(z, ls[:, :5])


def _enc_weights(ndet, psd):
    shapes = {}
    E = 192
    for i, (ci, co, k, _) in enumerate(lean_ref.STEM):
        shapes[f"stem.{2*i}.weight"] = (co, ci, k)
        shapes[f"stem.{2*i}.bias"] = (co,)
    shapes["detector_embed.weight"] = (ndet, E)
    for l in range(3):
        p = f"fusion.layers.{l}."
        shapes[p + "self_attn.in_proj_weight"] = (3 * E, E)
        shapes[p + "self_attn.in_proj_bias"] = (3 * E,)
        shapes[p + "self_attn.out_proj.weight"] = (E, E)
        shapes[p + "self_attn.out_proj.bias"] = (E,)
        shapes[p + "linear1.weight"] = (4 * E, E)
        shapes[p + "linear1.bias"] = (4 * E,)
        shapes[p + "linear2.weight"] = (E, 4 * E)
        shapes[p + "linear2.bias"] = (E,)
        for n in ("norm1", "norm2"):
            shapes[p + n + ".weight"] = (E,)
            shapes[p + n + ".bias"] = (E,)
    shapes["pool_queries"] = (8, E)
    shapes["pool_attn.in_proj_weight"] = (3 * E, E)
    shapes["pool_attn.in_proj_bias"] = (3 * E,)
    shapes["pool_attn.out_proj.weight"] = (E, E)
    shapes["pool_attn.out_proj.bias"] = (E,)
    shapes["energy_mlp.0.weight"] = (64, ndet * 16)
    shapes["energy_mlp.0.bias"] = (64,)
    shapes["energy_mlp.2.weight"] = (64, 64)
    shapes["energy_mlp.2.bias"] = (64,)
    nd = 0
    if psd:
        shapes["noise_mlp.0.weight"] = (64, ndet * psd)
        shapes["noise_mlp.0.bias"] = (64,)
        shapes["noise_mlp.2.weight"] = (32, 64)
        shapes["noise_mlp.2.bias"] = (32,)
        nd = 32
    shapes["out_proj.0.weight"] = (512, 8 * E + 64 + nd)
    shapes["out_proj.0.bias"] = (512,)
    shapes["out_proj.2.weight"] = (256, 512)
    shapes["out_proj.2.bias"] = (256,)
    return shapes


@pytest.mark.parametrize("tag,ndet,psd", [("det3", 3, 0), ("det1", 1, 0), ("det3_psd", 3, 16)])
def test_lean_strain_encoder(golden_encoder, tag, ndet, psd):
    torch.set_num_threads(4)
    w = recipe.fill_state_dict(_enc_weights(ndet, psd), seed=100 + ndet + psd)
    strain = recipe.strain_batch(4, ndet, seed=7)
    asd = torch.from_numpy(golden_encoder[f"{tag}_asd"]) if psd else None
    with torch.no_grad():
        clean = lean_ref.sanitize_strain(strain)
        close(lean_ref.window_log_energy(clean), golden_encoder[f"{tag}_log_energy"])
        x = torch.asinh(clean).reshape(4 * ndet, 1, -1)
        out, stages = lean_ref.stem_forward(w, x[:2], return_stages=True)
        close(stages[0][:, :, ::16], golden_encoder[f"{tag}_stage0"], 1e-4, 1e-5)
        close(stages[1][:, :, ::4], golden_encoder[f"{tag}_stage1"], 1e-4, 1e-5)
        close(stages[2], golden_encoder[f"{tag}_stage2"], 1e-4, 1e-5)
        close(out, golden_encoder[f"{tag}_stem_out"], 1e-4, 1e-5)
        feats, _ = lean_ref.encoder_features(w, strain, asd, psd_bands=psd)
        close(feats, golden_encoder[f"{tag}_feats"], 1e-4, 2e-5)
        close(lean_ref.out_proj(w, feats), golden_encoder[f"{tag}_ctx"], 1e-4, 2e-5)


def test_coherent_encoder(golden_encoder):
    torch.set_num_threads(4)
    shapes = _enc_weights(3, 16)
    shapes["geom_mlp.0.weight"] = (128, 201)
    shapes["geom_mlp.0.bias"] = (128,)
    shapes["geom_mlp.2.weight"] = (128, 128)
    shapes["geom_mlp.2.bias"] = (128,)
    shapes["geom_to_tokens.weight"] = (4 * 192, 128)
    shapes["geom_to_tokens.bias"] = (4 * 192,)
    w = recipe.fill_state_dict(shapes, seed=200)
    geom = lean_ref.CoherentGeometry()
    assert [geom.band_lo, geom.Nf, geom.maxlag] == list(golden_encoder["coh_band"])
    strain = recipe.strain_batch(4, 3, seed=9)
    asd = torch.from_numpy(golden_encoder["coh_asd"])
    with torch.no_grad():
        rel = geom.rel(lean_ref.sanitize_strain(strain))
        assert rel.shape == (4, 201)
        close(rel, golden_encoder["coh_rel"], 1e-4, 1e-5)
        ctx = lean_ref.coherent_encoder_forward(w, strain, asd, geom)
        close(ctx, golden_encoder["coh_ctx"], 1e-4, 2e-5)
