"""The strain embedding's HIP training path end to end (pf_embed_train_forward / _backward behind LeanStrainEncoder):
forward against the reference-made golden contexts, every gradient against float64 autograd through the CPU oracle
(oracle/lean_ref.py, itself pinned by those goldens), with and without dropout -- nn.Dropout replaced in the oracle by the
factors the kernels' counter hash produced (oracle/enc_dropout.py), since nn.Dropout's stream is not reproducible across
devices."""
import math

import numpy as np
import pytest
import torch

import recipe
from oracle import lean_ref
from oracle.enc_dropout import factors, seed32
from posteriflow_amd import npe

pytestmark = pytest.mark.gpu


def _load(enc, seed, skip=("pos.pe",)):
    shapes = {k: v.shape for k, v in enc.state_dict().items() if k not in skip}
    sd = recipe.fill_state_dict(shapes, seed=seed)
    missing = enc.load_state_dict(sd, strict=False)
    assert sorted(missing.missing_keys) == sorted(skip)
    return enc


@pytest.mark.parametrize("tag,ndet,psd", [("det3", 3, 0), ("det1", 1, 0), ("det3_psd", 3, 16)])
def test_fp32_parity_mode_runs_the_hip_path_and_matches_the_golden(golden_encoder, tag, ndet, psd):
    """the fp32 mode no longer evaluates nn.TransformerEncoder: stem, Transformer layers and pooling are fp32-MFMA kernels.
    Against the context the REFERENCE's own LeanStrainEncoder produced: 1e-5 relative (+ 2e-5 abs on values of scale ~4)."""
    enc = _load(npe.LeanStrainEncoder(n_detectors=ndet, psd_bands=psd), 100 + ndet + psd).cuda().eval()
    enc.precision = "fp32"
    strain = recipe.strain_batch(4, ndet, seed=7).cuda()
    asd = torch.from_numpy(golden_encoder[f"{tag}_asd"]).cuda() if psd else None
    called = []
    enc.fusion.register_forward_hook(lambda *a: called.append(1))
    with torch.no_grad():
        ctx = enc(strain, asd)
    assert not called, "nn.TransformerEncoder was evaluated"
    want = golden_encoder[f"{tag}_ctx"]
    err = np.abs(ctx.cpu().numpy() - want).max()
    print(f"\n[{tag}] fp32 HIP training-path forward vs reference golden: max abs {err:.2e} (scale {np.abs(want).max():.2f})")
    np.testing.assert_allclose(ctx.cpu().numpy(), want, rtol=1e-5, atol=2e-5)
    # bf16 training-path forward (the one a training step runs) against the same golden
    enc.precision = "bf16"
    for p in enc.parameters():
        p.requires_grad_(True)
    ctx16 = enc(strain, asd)
    assert ctx16.requires_grad
    e16 = np.abs(ctx16.detach().cpu().numpy() - want).max() / np.abs(want).max()
    print(f"[{tag}] bf16 training-path forward: {e16:.2e} of scale")
    assert e16 < 4e-2


def _oracle_grads(enc, strain, weight, drop=None, asd=None, psd=0):
    w = {k: v.detach().cpu().double().requires_grad_(v.dtype.is_floating_point) for k, v in enc.state_dict().items() if k != "pos.pe"}
    feats, _ = lean_ref.encoder_features(w, strain.cpu().double(), asd, psd_bands=psd, drop=drop)
    ctx = lean_ref.out_proj(w, feats)
    (ctx * weight.cpu().double()).sum().backward()
    return ctx.detach(), {k: v.grad for k, v in w.items() if v.requires_grad and v.grad is not None}


def _drop_factors(p, seed64, B, T):
    s = seed32(seed64)
    out = {}
    for l in range(3):
        out[(l, "attn")] = torch.from_numpy(factors(p, s, 4 * l + 0, B * 6 * T * 192)).reshape(B, 6, T, 192)[..., :T].double()
        out[(l, "res1")] = torch.from_numpy(factors(p, s, 4 * l + 1, B * T * 192)).reshape(B, T, 192).double()
        out[(l, "ffn")] = torch.from_numpy(factors(p, s, 4 * l + 2, B * T * 768)).reshape(B, T, 768).double()
        out[(l, "res2")] = torch.from_numpy(factors(p, s, 4 * l + 3, B * T * 192)).reshape(B, T, 192).double()
    return out


@pytest.mark.parametrize("ndet,train", [(3, False), (3, True), (1, True)])
def test_every_encoder_gradient_matches_float64_autograd_through_the_oracle(ndet, train):
    """fp32 mode: d(sum ctx * w) / d(parameter) for all 66 parameter tensors of LeanStrainEncoder (stem, embeddings, the
    three Transformer layers, pool, MLPs) within 1e-3 of each tensor's largest entry; train=True:
    train() mode, dropout 0.05 at the four sites of every layer, the oracle multiplying the kernels' own factors."""
    torch.manual_seed(3)
    enc = _load(npe.LeanStrainEncoder(n_detectors=ndet), 100 + ndet).cuda()
    enc.precision = "fp32"
    enc.train(train)
    B, T = 2, 61 * ndet
    strain = recipe.strain_batch(B, ndet, seed=7).cuda()
    weight = torch.randn(B, 256, generator=torch.Generator().manual_seed(1)).cuda()
    ctx = enc(strain)
    (ctx * weight).sum().backward()
    drop = None
    if train:
        seed = enc._train_state["last_seed"]
        assert seed is not None
        drop = _drop_factors(0.05, seed, B, T)
        zero = float((drop[(0, "res1")] == 0).double().mean())
        assert 0.03 < zero < 0.07
    want_ctx, want = _oracle_grads(enc, strain, weight, drop)
    e_ctx = ((ctx.detach().cpu().double() - want_ctx).abs().max() / want_ctx.abs().max()).item()
    print(f"\n[encoder gradients ndet={ndet} train={train}] forward rel {e_ctx:.2e}")
    assert e_ctx < 2e-5
    worst = (0.0, "")
    n = 0
    for name, p in enc.named_parameters():
        assert p.grad is not None and name in want, name
        g, w = p.grad.detach().cpu().double(), want[name]
        rel = ((g - w).abs().max() / w.abs().max().clamp_min(1e-30)).item()
        worst = max(worst, (rel, name))
        n += 1
        assert rel < 1e-5, (name, rel)          # measured: 2.65e-6 on the worst tensor
    print(f"  {n} parameter tensors, worst relative error {worst[0]:.2e} ({worst[1]})")
    # the pool's query rows: gradient through the host-side projection only
    assert enc.pool_attn.in_proj_weight.grad[:192].abs().max() > 0


def test_bf16_training_gradients_follow_the_fp32_oracle():
    """the throughput mode a trainer runs in: bf16 activations / MFMA operands, fp32 accumulation, weight gradients in fp32.
    Per parameter tensor the cosine with float64 autograd through the fp32 oracle, dropout off (its factors act on
    different roundings): > 0.98 for every tensor, > 0.995 for the whole gradient."""
    torch.manual_seed(3)
    enc = _load(npe.LeanStrainEncoder(n_detectors=3), 103).cuda().eval()
    enc.precision = "bf16"
    for p in enc.parameters():
        p.requires_grad_(True)
    B = 4
    strain = recipe.strain_batch(B, 3, seed=7).cuda()
    weight = torch.randn(B, 256, generator=torch.Generator().manual_seed(1)).cuda()
    (enc(strain) * weight).sum().backward()
    _, want = _oracle_grads(enc, strain, weight)
    cos = lambda a, b: torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
    got_all, want_all, worst = [], [], (1.0, "")
    for name, p in enc.named_parameters():
        g, w = p.grad.detach().cpu().double(), want[name]
        assert torch.isfinite(g).all(), name
        got_all.append(g.flatten()), want_all.append(w.flatten())
        worst = min(worst, (cos(g, w), name))
    total = cos(torch.cat(got_all), torch.cat(want_all))
    print(f"\n[bf16 encoder gradients] whole-gradient cosine {total:.5f}, worst tensor {worst[0]:.5f} ({worst[1]})")
    assert total > 0.995 and worst[0] > 0.98


def test_coherent_encoder_trains_through_the_hip_path(golden_encoder):
    """CoherentEncoder (4 geometry tokens prepended, 187 tokens): fp32 forward vs the reference-made golden, gradients of the
    geometry branch (through grad_extra_tokens) and of the Transformer vs float64 autograd through the oracle"""
    enc = _load(npe.CoherentEncoder(context_dim=256, psd_bands=16), 200, skip=("pos.pe", "Bsum", "bcount", "lags_norm")).cuda().eval()
    enc.precision = "fp32"
    strain = recipe.strain_batch(4, 3, seed=9).cuda()
    asd = torch.from_numpy(golden_encoder["coh_asd"]).cuda()
    for p in enc.parameters():
        p.requires_grad_(True)
    ctx = enc(strain, asd)
    np.testing.assert_allclose(ctx.detach().cpu().numpy(), golden_encoder["coh_ctx"], rtol=2e-4, atol=5e-5)
    weight = torch.randn(4, 256, generator=torch.Generator().manual_seed(2)).cuda()
    (ctx * weight).sum().backward()
    w = {k: v.detach().cpu().double().requires_grad_(v.dtype.is_floating_point) for k, v in enc.state_dict().items()
         if k not in ("pos.pe", "Bsum", "bcount", "lags_norm")}
    # coherent_encoder_forward (CE:118-123) step by step: the FFT geometry features are fp32 by construction (CE:93),
    # everything downstream in float64
    geom = lean_ref.CoherentGeometry()
    clean = lean_ref.sanitize_strain(strain.cpu())
    gfeat = lean_ref._mlp2(w, "geom_mlp", geom.rel(clean).double())
    gtok = torch.nn.functional.linear(gfeat, w["geom_to_tokens.weight"], w["geom_to_tokens.bias"]).reshape(-1, 4, 192)
    feats, _ = lean_ref.encoder_features(w, clean.double(), asd.cpu().double(), extra_tokens=gtok, psd_bands=16)
    out = lean_ref.out_proj(w, feats)
    (out * weight.cpu().double()).sum().backward()
    for name in ("geom_to_tokens.weight", "geom_mlp.0.weight", "fusion.layers.0.self_attn.in_proj_weight", "stem.0.weight",
                 "pool_queries", "detector_embed.weight"):
        g, ww = dict(enc.named_parameters())[name].grad.cpu().double(), w[name].grad
        rel = ((g - ww).abs().max() / ww.abs().max()).item()
        print(f"[coherent] {name}: {rel:.2e}")
        assert rel < 2e-3, (name, rel)


def test_flat_parameter_mode_of_the_encoder():
    """LeanStrainEncoder.flatten_parameters(): one leaf for the 46 HIP-trained tensors.  Same output, the leaf's gradient
    equals the per-tensor gradients laid end to end, state_dict keys and values unchanged, loading works both ways."""
    from posteriflow_amd import npe, _enc_train
    torch.manual_seed(3)
    a = npe.LeanStrainEncoder().cuda().train()
    b = npe.LeanStrainEncoder().cuda().train()
    b.load_state_dict(a.state_dict())
    b.flatten_parameters()
    assert set(a.state_dict()) == set(b.state_dict()) and all(torch.equal(v, b.state_dict()[k]) for k, v in a.state_dict().items())
    assert len(list(b.parameters())) < 20 < len(list(a.parameters()))
    for m in (a, b):
        m.precision = "bf16"
    strain = torch.randn(6, 3, 16384, device="cuda")
    outs = []
    for m in (a, b):
        torch.manual_seed(11)                                  # the dropout seed is drawn from torch's CPU generator
        y = m(strain)
        y.square().mean().backward()
        outs.append(y.detach())
    assert torch.equal(outs[0], outs[1])
    want = torch.cat([p.grad.reshape(-1) for p in _enc_train.train_parameters(a)])
    got = b._theta.grad
    assert got.shape == want.shape
    assert (got - want).abs().max() <= 1e-6 * want.abs().max() + 1e-12      # (float atomics: the order of additions differs)
    for (_, pa), (_, pb) in zip(sorted((n, p) for n, p in a.named_parameters() if n.startswith(("energy", "out_proj", "pool_q", "detector"))),
                                sorted((n, p) for n, p in b.named_parameters() if n.startswith(("energy", "out_proj", "pool_q", "detector")))):
        assert (pa.grad - pb.grad).abs().max() <= 1e-5 * pa.grad.abs().max() + 1e-12
    c = npe.LeanStrainEncoder().cuda()
    c.load_state_dict(b.state_dict())                          # flat -> plain
    assert torch.equal(c.stem[0].weight, a.stem[0].weight)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_no_grad_call_takes_the_forward_only_workspace_and_gives_the_same_context(precision, monkeypatch):
    """a call without gradients of the training-path embedding (fp32 parity mode, or train() mode) uses the
    forward-only workspace -- the layers share one set of activation buffers, none of the backward's temporaries exist -- in
    chunks of events.  Same context as the differentiable call, bit for bit (same kernels, same order); a strain that requires
    a gradient is refused instead of silently getting none."""
    from posteriflow_amd import npe, _enc_train, _lib
    import ctypes as C
    torch.manual_seed(5)
    enc = npe.LeanStrainEncoder().cuda().eval()          # eval: no dropout, the two calls are comparable
    enc.precision = precision
    strain = torch.randn(11, 3, 16384, device="cuda")
    if precision == "bf16":
        enc.train()                                       # bf16 + eval would take the fused inference kernels
        for m in enc.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        for l in enc.fusion.layers:
            l.self_attn.dropout = 0.0
    seen = []
    real = _lib.lib().pf_embed_train_workspace_bytes

    class Spy:                                            # records (forward_only, events) of every workspace query
        def __call__(self, desc, n):
            seen.append((int(desc._obj.forward_only), int(n)))
            return real(desc, n)
    monkeypatch.setattr(_lib.lib(), "pf_embed_train_workspace_bytes", Spy(), raising=False)
    want = enc(strain)                                    # parameters require grad: the differentiable route
    assert want.requires_grad and seen and all(f == 0 for f, _ in seen)
    seen.clear()
    monkeypatch.setattr(_enc_train, "FWD_ONLY_CHUNK", 4)
    with torch.no_grad():
        got = enc(strain)
    assert [n for _, n in seen] == [4, 4, 3] and all(f == 1 for f, _ in seen), seen
    assert not got.requires_grad and torch.equal(got, want.detach())
    for p in enc.parameters():                            # frozen parameters under grad mode: forward-only as well
        p.requires_grad_(False)
    seen.clear()
    got2 = enc(strain)
    assert all(f == 1 for f, _ in seen) and torch.equal(got2, want.detach())
    with pytest.raises(NotImplementedError):
        enc(strain.clone().requires_grad_(True))
