"""LeanNPE glue: ParamScaler / encoders against the golden vectors made by the reference's own
classes (CPU, tensor ops), and -- on the GPU -- LeanNPE.nll / sample_posterior / batch_nll
against the oracle composition."""
import numpy as np
import pytest
import torch

import recipe
from oracle import lean_ref
from posteriflow_amd import npe


def test_param_scaler_matches_reference_golden(golden_small):
    p = torch.from_numpy(golden_small["scaler_phys_in"]); raw = torch.from_numpy(golden_small["scaler_raw_in"])
    for tag, pre in (("", False), ("_premerger", True)):
        sc = npe.ParamScaler(premerger=pre)
        np.testing.assert_allclose(sc.normalize(p).numpy(), golden_small[f"scaler_norm{tag}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(sc.denormalize(raw).numpy(), golden_small[f"scaler_denorm{tag}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(sc.wrap(raw).numpy(), golden_small[f"scaler_wrap{tag}"], rtol=1e-5, atol=1e-6)
    assert set(npe.ParamScaler.RANGES) == set(npe.PARAM_NAMES) and set(npe.ParamScaler.CIRCULAR) == {"ra", "phase", "psi"}


def _load(enc, seed, skip=("pos.pe",)):
    shapes = {k: v.shape for k, v in enc.state_dict().items() if k not in skip}
    sd = recipe.fill_state_dict(shapes, seed=seed)
    missing = enc.load_state_dict(sd, strict=False)
    assert sorted(missing.missing_keys) == sorted(skip)
    for p in enc.parameters():
        p.requires_grad_(False)
    return enc.eval()


@pytest.mark.parametrize("tag,ndet,psd", [("det3", 3, 0), ("det1", 1, 0), ("det3_psd", 3, 16)])
def test_lean_strain_encoder_golden_cpu(golden_encoder, tag, ndet, psd):
    torch.set_num_threads(4)
    enc = _load(npe.LeanStrainEncoder(n_detectors=ndet, psd_bands=psd), 100 + ndet + psd)
    with pytest.raises(RuntimeError):
        enc(torch.zeros(1, ndet, 16384))            # product path: no CPU fallback
    enc._allow_tensor_op_stem = True                # wiring / state_dict check on the CPU only
    strain = recipe.strain_batch(4, ndet, seed=7)
    asd = torch.from_numpy(golden_encoder[f"{tag}_asd"]) if psd else None
    with torch.no_grad():
        ctx = enc(strain, asd)
    np.testing.assert_allclose(ctx.numpy(), golden_encoder[f"{tag}_ctx"], rtol=1e-4, atol=2e-5)


def test_coherent_encoder_golden_cpu(golden_encoder):
    torch.set_num_threads(4)
    enc = _load(npe.CoherentEncoder(context_dim=256, psd_bands=16), 200, skip=("pos.pe", "Bsum", "bcount", "lags_norm"))
    enc._allow_tensor_op_stem = True
    assert [enc.band_lo, enc.Nf, enc.maxlag] == list(golden_encoder["coh_band"])
    strain = recipe.strain_batch(4, 3, seed=9)
    with torch.no_grad():
        ctx = enc(strain, torch.from_numpy(golden_encoder["coh_asd"]))
    np.testing.assert_allclose(ctx.numpy(), golden_encoder["coh_ctx"], rtol=1e-4, atol=2e-5)


def test_deepcopy_of_flattened_modules():
    """copy.deepcopy (EMA copies, snapshots) of a model whose parameters were flattened -- the sub-modules hold
    non-leaf views of the flat leaf, which torch refuses to deep-copy"""
    import copy
    from posteriflow_amd import LeanNPE, NSFPosteriorFlow
    flow = NSFPosteriorFlow(11, 288, 64, 2, 8, 5.0, use_masked_context=False).flatten_parameters()
    twin = copy.deepcopy(flow)
    assert twin._theta is not flow._theta and torch.equal(twin._theta, flow._theta)
    net = twin._ar_transforms[0].autoregressive_net
    assert net.initial_layer.weight.data_ptr() == twin._theta.data_ptr()            # the copy's views are views of ITS leaf
    with torch.no_grad():
        twin._theta.add_(1.0)
    assert not torch.equal(twin._theta, flow._theta)
    assert set(twin.state_dict()) == set(flow.state_dict())
    model = LeanNPE(flow_layers=2, flow_hidden=64).flatten_parameters()
    ema = copy.deepcopy(model)
    assert ema.encoder._theta is not model.encoder._theta
    assert ema.encoder.stem[0].weight.data_ptr() == ema.encoder._theta.data_ptr()
    assert model.encoder.stem[0].weight.data_ptr() == model.encoder._theta.data_ptr()   # the original keeps its views
    a, b = model.state_dict(), ema.state_dict()
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    copy.deepcopy(LeanNPE(flow_layers=2, flow_hidden=64))                              # not flattened: unchanged behaviour


def test_state_dict_layout_matches_reference_names():
    m = npe.LeanNPE(flow_layers=2)
    keys = set(m.state_dict())
    for k in ("encoder.stem.0.weight", "encoder.stem.6.bias", "encoder.detector_embed.weight", "encoder.pos.pe",
              "encoder.fusion.layers.2.self_attn.in_proj_weight", "encoder.fusion.layers.0.norm1.weight",
              "encoder.pool_queries", "encoder.pool_attn.out_proj.weight", "encoder.energy_mlp.2.bias",
              "encoder.out_proj.2.weight", "rank_embed.weight", "flow.temperature", "flow._ar_perm",
              "flow.transform._transforms.0._permutation",
              "flow.transform._transforms.1.autoregressive_net.blocks.1.linear_layers.0.mask",
              "flow.flow._transform._transforms.3.autoregressive_net.final_layer.weight"):
        assert k in keys, k
    assert sum(p.numel() for p in m.encoder.parameters()) == 2642336          # SURVEY 8c probe
    assert not m.flow.temperature.requires_grad and m.flow._tail_bound == 5.0
    assert sum(p.numel() for p in npe.CoherentEncoder(context_dim=256, psd_bands=16).parameters()) == 2805376


@pytest.mark.gpu
@pytest.mark.parametrize("tag,ndet", [("det3", 3), ("det1", 1)])
def test_hip_stem_against_reference_golden(golden_encoder, tag, ndet):
    """pf_embed_stem_forward (through LeanStrainEncoder._stem_hip) vs the reference's own stem /
    energy-window outputs.  fp32 mode: 2e-5 abs (values O(1)); bf16 mode: 3e-2 (bf16 operands, K = 512)."""
    enc = _load(npe.LeanStrainEncoder(n_detectors=ndet), 100 + ndet).cuda()
    strain = recipe.strain_batch(4, ndet, seed=7).cuda()
    want_tok = torch.from_numpy(golden_encoder[f"{tag}_stem_out"]).transpose(1, 2)      # [2, 61, 192]
    with torch.no_grad():
        tok, le = enc._stem_hip(strain)
        assert tok.shape == (4 * ndet, 61, 192) and le.shape == (4, ndet, 16)
        np.testing.assert_allclose(le.cpu().numpy(), golden_encoder[f"{tag}_log_energy"], rtol=1e-5, atol=1e-5)
        err = (tok[:2].cpu() - want_tok).abs().max().item()
        print(f"\n[{tag}] fp32 stem max abs err {err:.2e}")
        assert err < 2e-5
        ctx = enc(strain)
        np.testing.assert_allclose(ctx.cpu().numpy(), golden_encoder[f"{tag}_ctx"], rtol=2e-3, atol=1e-3)
        enc.precision = "bf16"
        tok16, _ = enc._stem_hip(strain)
        err16 = (tok16[:2].cpu() - want_tok).abs().max().item()
        print(f"[{tag}] bf16 stem max abs err {err16:.2e}")
        assert err16 < 3e-2
    with pytest.raises(RuntimeError):
        enc._stem_hip(strain.cpu())
    # differentiable call: interim tensor-op stem under autograd, same value as the HIP stem
    enc.precision = "fp32"
    for p in enc.parameters():
        p.requires_grad_(True)
    ctx_g = enc(strain)
    assert ctx_g.requires_grad and torch.allclose(ctx_g.detach(), ctx, rtol=1e-3, atol=1e-3)
    ctx_g.square().sum().backward()
    assert enc.stem[0].weight.grad is not None and torch.isfinite(enc.stem[0].weight.grad).all()


@pytest.mark.gpu
def test_lean_npe_nll_and_sampling_gpu(golden_encoder):
    from helpers import oracle_state_for_product
    from oracle.flow_ref import NSFPosteriorFlowRef
    torch.manual_seed(0)
    model = npe.LeanNPE(flow_layers=3)
    _load(model.encoder, 103)
    ref_flow = NSFPosteriorFlowRef(11, 288, 256, 3, 16, 5.0, temperature_scale=1.0)
    model.flow.load_state_dict(oracle_state_for_product(ref_flow))
    model = model.to("cuda").eval()
    strain = recipe.strain_batch(4, 3, seed=7)
    phys = recipe.physical_params(8, seed=11)[:4]
    rank = torch.tensor([0, 1, 0, 2])
    with torch.no_grad():
        ctx = model.encode(strain.cuda())
        # stem output agrees with the golden to 1e-6 on the GPU; the fusion transformer runs on the
        # device BLAS / SDPA path this round and lands within 4e-4 abs (scale 4.5) of the CPU golden
        np.testing.assert_allclose(ctx.cpu().numpy(), golden_encoder["det3_ctx"], rtol=2e-3, atol=1e-3)
        nll = model.nll(strain.cuda(), phys.cuda(), rank.cuda()).cpu()
        # oracle composition: reference-pinned context -> rank embedding -> scaler -> flow
        full = torch.cat([torch.from_numpy(golden_encoder["det3_ctx"]), model.rank_embed.weight.cpu()[rank]], dim=1)
        y = lean_ref.ParamScalerRef().normalize(phys)
        want = ref_flow.compute_psd_aware_nll(y, full, torch.zeros_like(y))
        assert ((nll - want).abs() / want.abs().clamp_min(1)).max() < 2e-3      # encoder runs TF32-free fp32 on GPU
        draws = model.sample_posterior(strain.cuda(), rank=0, n_samples=64)
        assert draws.shape == (4, 64, 11) and torch.isfinite(draws).all()
        lo, hi = npe.ParamScaler().denormalize(torch.tensor([[-1.0] * 11, [1.0] * 11]))
        assert (draws.cpu() >= lo - 1e-3).all() and (draws.cpu() <= hi * (1 + 1e-5) + 1e-5).all()
        # batch_nll: one flattened flow call == the reference's per-rank loop
        params = torch.stack([recipe.physical_params(8, seed=20 + r)[4:] for r in range(5)], dim=1)   # [4,5,11]
        nsig = torch.tensor([1, 3, 5, 2])
        got = npe.batch_nll(model, strain.cuda(), params.cuda(), nsig.cuda()).item()
        ctx_cpu = ctx.cpu()
        ref_nll = lambda p, r, c: ref_flow.compute_psd_aware_nll(
            lean_ref.ParamScalerRef().normalize(p), torch.cat([c, model.rank_embed.weight.cpu()[r]], dim=1),
            torch.zeros_like(p))
        want_b = lean_ref.batch_nll_ref(ref_nll, ctx_cpu, params, nsig).item()
        assert abs(got - want_b) / abs(want_b) < 1e-3
        # static row cap (11 existing pairs of 20): the same mean from 12 flow rows instead of 20; a cap that is too
        # small is reported, not silently absorbed
        got_c = npe.batch_nll(model, strain.cuda(), params.cuda(), nsig.cuda(), row_cap=12).item()
        assert abs(got_c - want_b) / abs(want_b) < 1e-3 and int(npe.batch_nll.last_overflow) == 0
        npe.batch_nll(model, strain.cuda(), params.cuda(), nsig.cuda(), row_cap=8)
        assert int(npe.batch_nll.last_overflow) == 3
        # "exact": only the 11 existing pairs go through the flow (what the reference's per-rank loop evaluates); the
        # gradient of the context agrees with the padded form's
        got_e = npe.batch_nll(model, strain.cuda(), params.cuda(), nsig.cuda(), row_cap="exact").item()
        assert abs(got_e - want_b) / abs(want_b) < 1e-3
        with pytest.raises(ValueError):
            npe.batch_nll(model, strain.cuda(), params.cuda(), nsig.cuda(), row_cap="all")


@pytest.mark.gpu
def test_inference_sampling_loop_gpu():
    """pipeline.py:57-76, 161-186 counterpart: physical-units log q against the oracle's closed
    form on the same draws, railing mask, mass ordering, chunking invariance of shapes."""
    from helpers import oracle_state_for_product
    from oracle.flow_ref import NSFPosteriorFlowRef
    from posteriflow_amd import inference
    torch.manual_seed(0)
    model = npe.LeanNPE(flow_layers=2)
    _load(model.encoder, 103)
    ref_flow = NSFPosteriorFlowRef(11, 288, 256, 2, 16, 5.0, temperature_scale=1.0)
    model.flow.load_state_dict(oracle_state_for_product(ref_flow))
    model = model.to("cuda").eval()
    strain = recipe.strain_batch(1, 3, seed=7).cuda()
    out = inference.sample_event(model, strain, num_samples=5000, rank=1, seed=3, batch_size=2048)
    s, logq = out["samples"], out["logq"]
    assert s.shape == (5000, 11) and s.dtype == torch.float64 and logq.shape == (5000,)
    assert (s[:, 0] >= s[:, 1]).all()                                           # m1 >= m2
    lo, hi = npe.ParamScaler().denormalize(torch.tensor([[-1.0] * 11, [1.0] * 11])).double()
    assert (s.cpu() >= lo - 1e-6).all() and (s.cpu() <= hi * (1 + 1e-6)).all()
    assert 0.0 <= out["boundary_railing_frac"].item() <= 1.0
    # log q check on a fresh set of normalised points (no mass swap involved)
    g = torch.Generator().manual_seed(5)
    y = (torch.rand(64, 11, generator=g) * 1.9 - 0.95)
    full = torch.cat([out["context"].cpu(), model.rank_embed.weight.cpu()[torch.tensor([1])]], dim=1)
    got = inference.log_prob_physical(model, y.cuda(), full.cuda()).cpu()
    neg = ref_flow.compute_psd_aware_nll(y, full.expand(64, -1), torch.zeros_like(y))
    want = lean_ref.log_prob_physical(neg, y, lean_ref.ParamScalerRef())
    assert ((got - want).abs() / want.abs().clamp_min(1)).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_encoder_and_npe_with_an_empty_batch(precision):
    """zero events: the encoder returns [0, context_dim], LeanNPE.nll an empty loss vector -- no kernel is launched with
    an empty grid"""
    from posteriflow_amd import LeanNPE
    model = LeanNPE(flow_layers=2).cuda().eval()
    model.encoder.precision = model.flow.precision = precision
    with torch.no_grad():
        ctx = model.encode(torch.empty(0, 3, 16384, device="cuda"))
        assert ctx.shape == (0, model.context_dim)
        strain = torch.empty(0, 3, 16384, device="cuda")
        nll = model.nll(strain, torch.empty(0, 11, device="cuda"), torch.empty(0, dtype=torch.long, device="cuda"), context=ctx)
        assert nll.shape == (0,)


def test_coherent_geometry_plan_is_the_band_membership():
    """the HIP geometry kernel takes the 16 log-spaced bands (coherent_encoder.py:60-66) as contiguous ranges of the kept rfft
    bins: the plan derived from the Bsum buffer partitions [0, Nf), agrees with the membership matrix, and refuses a
    membership that is not a partition into intervals (tensor ops then)"""
    enc = npe.CoherentEncoder(context_dim=256, psd_bands=16)
    edges = enc._geometry_plan()
    assert edges is not None and len(edges) == 17 and edges[0] == 0 and edges[-1] == enc.Nf == 4016
    assert [enc.band_lo, enc.maxlag] == [80, 122]
    m = enc.Bsum.numpy()
    for b in range(16):
        want = np.zeros(enc.Nf, np.float32)
        want[edges[b]:edges[b + 1]] = 1.0
        assert np.array_equal(m[b], want) and edges[b + 1] - edges[b] == int(enc.bcount[b])
    enc2 = npe.CoherentEncoder(context_dim=256, psd_bands=16)
    enc2.Bsum[3, 10] = 1.0                                    # bin 10 in two bands: not a partition
    assert enc2._geometry_plan() is None
