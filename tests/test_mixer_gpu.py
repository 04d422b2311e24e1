"""The fused HIP token mixer (pf_embed_fusion_forward: 3 Transformer layers + pool attention) against the
oracle's explicit Transformer arithmetic (oracle/lean_ref.py, itself pinned by the reference's golden
encoder outputs): once with the kernel's operand rounding mirrored (bf16 round trip on every
matrix-product operand; tight) and once against plain fp32 (bf16-sized tolerance)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import recipe                                            # noqa: E402
from oracle import lean_ref                              # noqa: E402

pytestmark = pytest.mark.gpu
bf16_rt = lambda t: t.to(torch.bfloat16).float()


def _encoder(seed=103):
    from posteriflow_amd import npe
    enc = npe.LeanStrainEncoder().eval()
    shapes = {k: v.shape for k, v in enc.state_dict().items() if k != "pos.pe"}
    enc.load_state_dict(recipe.fill_state_dict(shapes, seed=seed), strict=False)
    return enc


def _oracle_mix(w, tok, rnd):
    out = lean_ref.fusion_forward(w, tok, rnd=rnd)
    E = tok.shape[-1]
    q = w["pool_queries"].unsqueeze(0).expand(tok.shape[0], -1, -1)
    in_w, in_b = w["pool_attn.in_proj_weight"], w["pool_attn.in_proj_bias"]
    # heads of the pool attention BEFORE its out-projection (what the kernel returns); the query side
    # is input-independent and stays fp32 until it becomes a matrix-product operand
    qp = (F_linear(q, in_w[:E], in_b[:E]) / math.sqrt(32)).reshape(-1, 8, 6, 32).transpose(1, 2)
    k = F_linear(rnd(out), rnd(in_w[E:2 * E]), in_b[E:2 * E]).reshape(-1, tok.shape[1], 6, 32).transpose(1, 2)
    v = F_linear(rnd(out), rnd(in_w[2 * E:]), in_b[2 * E:]).reshape(-1, tok.shape[1], 6, 32).transpose(1, 2)
    return out, lean_ref._softmax_weighted(rnd(qp) @ rnd(k).transpose(-1, -2), v, rnd).transpose(1, 2).reshape(-1, 8, E)


def F_linear(x, w, b):
    return torch.nn.functional.linear(x, w, b)


def _run_kernel(enc, tok, tok_bias=None):
    from posteriflow_amd import _lib
    n_events, T, E = tok.shape
    enc = enc.cuda()
    enc.precision = "bf16"
    x = tok.cuda().contiguous()
    enc._mix_hip(x[:1].clone())                       # builds the packed parameter block
    packed = enc.__dict__["_mixer_state"]["packed"]
    q = ((enc.pool_queries @ enc.pool_attn.in_proj_weight[:E].t() + enc.pool_attn.in_proj_bias[:E]) / math.sqrt(32)).contiguous()
    pooled = torch.empty(n_events, 8, E, device="cuda")
    tb = None if tok_bias is None else tok_bias.cuda().contiguous()
    _lib.check(_lib.lib().pf_embed_fusion_forward(packed.data_ptr(), x.data_ptr(), T, 0 if tb is None else tb.data_ptr(),
                                                  q.data_ptr(), n_events, pooled.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), "fusion")
    return x.cpu(), pooled.cpu()


def _silence(enc, keep):
    """zero the output projections of every block except ``keep`` ('attn' | 'ffn' | 'both') of layer 0, so
    that the kernel's output is exactly that block's result (the other blocks add 0)."""
    with torch.no_grad():
        for li, layer in enumerate(enc.fusion.layers):
            if li > 0 or keep == "ffn":
                layer.self_attn.out_proj.weight.zero_(), layer.self_attn.out_proj.bias.zero_()
            if li > 0 or keep == "attn":
                layer.linear2.weight.zero_(), layer.linear2.bias.zero_()


@pytest.mark.parametrize("keep", ["attn", "ffn", "both"])
def test_one_layer_is_exact_up_to_rounding_boundary_flips(keep):
    """with the operand rounding mirrored on the CPU a single block agrees to fp32 accumulation noise in
    the bulk (median); the maximum is a bf16 rounding-boundary flip of an intermediate (one bf16 ulp of one
    operand).  This is the test that pins the kernel's arithmetic."""
    enc = _encoder()
    _silence(enc, keep)
    w = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    tok = torch.randn(3, 183, 192, generator=torch.Generator().manual_seed(1)) * 0.7
    with torch.no_grad():
        out_emu, pool_emu = _oracle_mix(w, tok, bf16_rt)
    bias = torch.randn(183, 192, generator=torch.Generator().manual_seed(2)) * 0.3 if keep == "both" else None
    got_out, got_pool = _run_kernel(enc, tok if bias is None else tok - bias, bias)     # token_bias is added on load
    assert (got_out - tok).abs().median() > 0.05                    # the block did something
    d_out, d_pool = (got_out - out_emu).abs(), (got_pool - pool_emu).abs()
    assert d_out.median() < 2e-6 and d_out.max() < 1.5e-2, (d_out.median(), d_out.max())
    assert d_pool.median() < 1e-4 and d_pool.max() < 3e-3, (d_pool.median(), d_pool.max())


@pytest.mark.parametrize("n_events,T", [(5, 183), (3, 187), (2, 61), (1, 192), (2, 17), (3, 1)])
def test_fused_mixer_matches_oracle_transformer(n_events, T):
    """all three layers: rounding flips of layer 1 spread through the LayerNorms and attentions of layers
    2-3, so kernel and rounding-mirrored oracle are two realisations of the same bf16 noise: the kernel
    must be as close to the fp32 oracle as the mirrored oracle is, and close to the mirrored one."""
    enc = _encoder()
    w = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    tok = torch.randn(n_events, T, 192, generator=torch.Generator().manual_seed(T)) * 0.7
    with torch.no_grad():
        out_emu, pool_emu = _oracle_mix(w, tok, bf16_rt)
        out_f32, pool_f32 = _oracle_mix(w, tok, lambda t: t)
    got_out, got_pool = _run_kernel(enc, tok)
    assert torch.isfinite(got_out).all() and torch.isfinite(got_pool).all()
    for got, emu, f32 in ((got_out, out_emu, out_f32), (got_pool, pool_emu, pool_f32)):
        scale = f32.abs().max()
        noise_med, noise_max = (emu - f32).abs().median(), (emu - f32).abs().max()
        assert (got - f32).abs().median() < 1.3 * noise_med and (got - f32).abs().max() < 2.0 * noise_max
        assert (got - emu).abs().median() < 1e-3 * scale and (got - emu).abs().max() < 1e-2 * scale
        assert (got - f32).abs().max() < 4e-2 * scale


def test_encoder_bf16_uses_the_hip_mixer_and_agrees_with_fp32_mode():
    enc = _encoder().cuda()
    strain = recipe.strain_batch(4, 3, seed=7).cuda()
    with torch.no_grad():
        enc.precision = "fp32"
        want = enc(strain)
        enc.precision = "bf16"
        got = enc(strain)
        assert "_mixer_state" in enc.__dict__                  # the fused kernel ran
        err = (got - want).abs().max() / want.abs().max()
        assert err < 4e-2, err
        gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder.npz"))["det3_ctx"]
        assert np.abs(got.cpu().numpy() - gold).max() / np.abs(gold).max() < 4e-2
        # training mode (dropout active) and differentiable calls stay on tensor ops
        enc.train()
        enc.__dict__.pop("_mixer_state")
        enc(strain)
        assert "_mixer_state" not in enc.__dict__
    with pytest.raises(NotImplementedError):
        from posteriflow_amd import _lib
        _lib.check(_lib.lib().pf_embed_fusion_forward(1, 1, 200, 0, 1, 1, 1, 0), "fusion")


@pytest.mark.parametrize("n_det", [1, 2])
def test_fewer_detectors_bf16_vs_fp32(n_det):
    """config 2 of BASELINE (single-detector strain): 61 / 122 tokens through the same kernels."""
    from posteriflow_amd import npe
    torch.manual_seed(3)
    enc = npe.LeanStrainEncoder(n_detectors=n_det).cuda().eval()
    strain = recipe.strain_batch(3, n_det, seed=11).cuda()
    with torch.no_grad():
        enc.precision = "fp32"
        want = enc(strain)
        enc.precision = "bf16"
        got = enc(strain)
    assert "_mixer_state" in enc.__dict__
    assert (got - want).abs().max() / want.abs().max() < 4e-2
