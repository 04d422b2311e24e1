"""The building blocks of the strain embedding's training path, each against the same op on device tensor ops / autograd
(fp32 reference of the same op), through the C ABI: pf_dense_nt, pf_dense_tn, pf_enc_ln_*, pf_enc_attn_*, pf_enc_pool_*.
fp32 mode (v_mfma_f32_16x16x4_f32): 1e-5-level agreement; bf16 mode: against the reference evaluated on bf16-rounded
operands (what the kernel multiplies), fp32 accumulation noise + one output rounding."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

PREC = {"fp32": 0, "bf16": 1}


def _L():
    from posteriflow_amd import _lib
    return _lib, _lib.lib()


def stream():
    return torch.cuda.current_stream().cuda_stream


def act_dtype(precision):
    return torch.bfloat16 if precision == "bf16" else torch.float32


def ptr(t):
    return 0 if t is None else t.data_ptr()


def pack_matrix(w, precision, mode=0):
    """W [N, K] (mode 0) or its transpose source [K, N] (mode 1) -> MFMA fragments"""
    lib, L = _L()
    n, k = (w.shape if mode == 0 else (w.shape[1], w.shape[0]))
    out = torch.empty(L.pf_dense_frag_bytes(PREC[precision], n, k), dtype=torch.uint8, device="cuda")
    src = w.contiguous().float()
    lib.check(L.pf_dense_pack_matrix(PREC[precision], src.data_ptr(), mode, src.shape[1], n, k, out.data_ptr(), stream()), "pack")
    return out


def dense_nt(precision, epi, a, w, bias=None, kc=None, out=None, **kw):
    lib, L = _L()
    m, k = a.shape
    n = w.shape[0]
    frags = pack_matrix(w, precision)
    args = lib.PfDenseArgs()
    args.A, args.M, args.rows_per_seq, args.a_seq_stride, args.lda = a.data_ptr(), m, max(m, 1), 0, k
    args.K, args.N, args.KC = k, n, kc or (k if k <= 256 else 192)
    args.wfrags, args.bias = frags.data_ptr(), ptr(bias)
    f32_out = epi == 2 or kw.get("out_f32", False)
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32 if f32_out else act_dtype(precision), device="cuda")
    args.out, args.o_seq_stride, args.ldo, args.o_valid_per_seq, args.x_seq_stride = out.data_ptr(), 0, n, 0, 0
    args.dact, args.resid, args.mul = ptr(kw.get("dact")), ptr(kw.get("resid")), ptr(kw.get("mul"))
    args.drop_p, args.seed, args.site = kw.get("drop_p", 0.0), kw.get("seed", 0), kw.get("site", 0)
    args.out_f32 = 1 if kw.get("out_f32", False) else 0
    lib.check(L.pf_dense_nt(PREC[precision], epi, C.byref(args), stream()), "pf_dense_nt")
    return out


def rnd(t, precision):
    return t.bfloat16().float() if precision == "bf16" else t


def tol(precision, scale=1.0):
    return (2e-2 if precision == "bf16" else 2e-5) * scale


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("m,k,n", [(300, 192, 576), (257, 192, 768), (130, 768, 192), (64, 576, 192), (1000, 384, 192), (5, 192, 192),
                                   (129, 256, 256), (77, 64, 32), (200, 256, 128)])
def test_dense_nt_plain_matches_matmul(precision, m, k, n):
    g = torch.Generator().manual_seed(m + k + n)
    a = torch.randn(m, k, generator=g).cuda()
    w = (torch.randn(n, k, generator=g) / math.sqrt(k)).cuda()
    b = torch.randn(n, generator=g).cuda()
    got = dense_nt(precision, 0, a.to(act_dtype(precision)).contiguous(), w, b).float()
    want = rnd(a, precision).double() @ rnd(w, precision).double().t() + b.double()
    err = (got.double() - want).abs().max().item()
    print(f"\n[dense_nt {precision} {m}x{k}x{n}] max err {err:.2e}")
    assert err < tol(precision, want.abs().max().item())
    # fp32 output in either precision
    got32 = dense_nt(precision, 0, a.to(act_dtype(precision)).contiguous(), w, b, out_f32=True)
    assert got32.dtype == torch.float32 and (got32.double() - want).abs().max().item() < 2e-5 * want.abs().max().item() * (k ** 0.5)


def test_dense_nt_64_row_strips_at_the_default_threshold():
    """Round 4: from 16 384 rows the bf16 strip GEMM with the plain / GELU epilogue runs 64-row strips (three workgroups per CU).
    A ragged M just above the threshold through the DEFAULT choice (no $PF_DENSE_BM): plain (single chunk and chunked K) and
    GELU + derivative with dropout, against float64 matmul of the rounded operands; the same call below the threshold (128-row
    strips) must give bit-identical rows -- a row's result does not depend on the strip it sits in."""
    import os
    from oracle.enc_dropout import factors
    assert "PF_DENSE_BM" not in os.environ
    precision = "bf16"
    g = torch.Generator().manual_seed(64)
    m = 16384 + 77
    for k, n in ((192, 576), (768, 192)):
        a = torch.randn(m, k, generator=g).cuda()
        w = (torch.randn(n, k, generator=g) / math.sqrt(k)).cuda()
        b = torch.randn(n, generator=g).cuda()
        ad = a.to(act_dtype(precision)).contiguous()
        got = dense_nt(precision, 0, ad, w, b).float()
        want = rnd(a, precision).double() @ rnd(w, precision).double().t() + b.double()
        err = (got.double() - want).abs().max().item()
        print(f"\n[dense_nt 64-row strips {m}x{k}x{n}] max err {err:.2e}")
        assert err < tol(precision, want.abs().max().item())
        small = dense_nt(precision, 0, ad[:1000].contiguous(), w, b).float()            # 128-row strips
        assert torch.equal(small, got[:1000])
    k, n, p = 192, 768, 0.25
    a = torch.randn(m, k, generator=g).cuda()
    w = (torch.randn(n, k, generator=g) / math.sqrt(k)).cuda()
    b = torch.randn(n, generator=g).cuda()
    ad = a.to(act_dtype(precision)).contiguous()
    dact = torch.empty(m, n, dtype=act_dtype(precision), device="cuda")
    got = dense_nt(precision, 1, ad, w, b, dact=dact, drop_p=p, seed=77, site=5).float().double()
    fac = torch.from_numpy(factors(p, 77, 5, m * n)).reshape(m, n).cuda().double()
    x = (rnd(a, precision).double() @ rnd(w, precision).double().t() + b.double()).requires_grad_(True)
    y = F.gelu(x)
    (dy,) = torch.autograd.grad(y.sum(), x)
    e1, e2 = (got - y.detach() * fac).abs().max().item(), (dact.float().double() - dy * fac).abs().max().item()
    print(f"[gelu 64-row strips p={p}] out {e1:.2e} dact {e2:.2e}")
    assert e1 < tol(precision, 4.0) and e2 < tol(precision, 2.0)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_dense_nt_epilogues(precision):
    from oracle.enc_dropout import factors
    g = torch.Generator().manual_seed(3)
    m, k, n = 333, 192, 768
    a = torch.randn(m, k, generator=g).cuda()
    w = (torch.randn(n, k, generator=g) / math.sqrt(k)).cuda()
    b = torch.randn(n, generator=g).cuda()
    ad = a.to(act_dtype(precision)).contiguous()
    pre = (rnd(a, precision).double() @ rnd(w, precision).double().t() + b.double())
    # GELU + its derivative, with and without dropout
    for p in (0.0, 0.25):
        dact = torch.empty(m, n, dtype=act_dtype(precision), device="cuda")
        got = dense_nt(precision, 1, ad, w, b, dact=dact, drop_p=p, seed=77, site=5).float().double()
        fac = torch.from_numpy(factors(p, 77, 5, m * n)).reshape(m, n).cuda().double()
        x = pre.clone().requires_grad_(True)
        y = F.gelu(x)
        (dy,) = torch.autograd.grad(y.sum(), x)
        e1, e2 = (got - y.detach() * fac).abs().max().item(), (dact.float().double() - dy * fac).abs().max().item()
        print(f"\n[gelu {precision} p={p}] out {e1:.2e} dact {e2:.2e}")
        assert e1 < tol(precision, 4.0) and e2 < tol(precision, 2.0)
        if p > 0:
            assert abs((fac == 0).double().mean().item() - p) < 0.01
    # residual (+ dropout), fp32 out
    n2 = 192
    w2 = (torch.randn(n2, k, generator=g) / math.sqrt(k)).cuda()
    b2 = torch.randn(n2, generator=g).cuda()
    res = torch.randn(m, n2, generator=g).cuda()
    for p in (0.0, 0.1):
        got = dense_nt(precision, 2, ad, w2, b2, resid=res, drop_p=p, seed=5, site=9).double()
        fac = torch.from_numpy(factors(p, 5, 9, m * n2)).reshape(m, n2).cuda().double()
        want = res.double() + fac * (rnd(a, precision).double() @ rnd(w2, precision).double().t() + b2.double())
        assert (got - want).abs().max().item() < (2e-4 if precision == "fp32" else 2e-2)
    # multiply (GELU backward / transposed convolution)
    mul = torch.randn(m, n, generator=g).cuda().to(act_dtype(precision)).contiguous()
    got = dense_nt(precision, 3, ad, w, None, mul=mul).float().double()
    want = (rnd(a, precision).double() @ rnd(w, precision).double().t()) * mul.float().double()
    assert (got - want).abs().max().item() < tol(precision, 8.0)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_dense_nt_overlapping_windows_is_a_convolution(precision):
    """sequence-strided rows with ld < K: conv1d over position-major activations, and the transposed convolution
    (data gradient) over a zero-padded gradient image with the o_valid limit"""
    lib, L = _L()
    g = torch.Generator().manual_seed(11)
    n_seq, cin, cout, kw, s, lin = 5, 32, 64, 16, 4, 203
    lout = (lin - kw) // s + 1
    x = torch.randn(n_seq, cin, lin, generator=g).cuda()
    w = (torch.randn(cout, cin, kw, generator=g) / math.sqrt(cin * kw)).cuda()
    b = torch.randn(cout, generator=g).cuda()
    xr, wr = rnd(x, precision), rnd(w, precision)
    want = F.conv1d(xr.double(), wr.double(), b.double(), stride=s)                    # [n, cout, lout]
    xp = x.transpose(1, 2).contiguous().to(act_dtype(precision))                       # position-major [n][lin][cin]
    wm = w.permute(0, 2, 1).reshape(cout, kw * cin).contiguous()                       # im2col order: tap * cin + ch
    frags = pack_matrix(wm, precision)
    out = torch.empty(n_seq, lout, cout, dtype=act_dtype(precision), device="cuda")
    a = lib.PfDenseArgs()
    a.A, a.M, a.rows_per_seq, a.a_seq_stride, a.lda = xp.data_ptr(), n_seq * lout, lout, lin * cin, s * cin
    a.K, a.N, a.KC = kw * cin, cout, 256
    a.wfrags, a.bias, a.out, a.o_seq_stride, a.ldo = frags.data_ptr(), b.data_ptr(), out.data_ptr(), lout * cout, cout
    lib.check(L.pf_dense_nt(PREC[precision], 0, C.byref(a), stream()), "conv as dense_nt")
    err = (out.float().double().transpose(1, 2) - want).abs().max().item()
    print(f"\n[conv via dense_nt {precision}] {err:.2e}")
    assert err < tol(precision, 4.0)
    # transposed convolution: dX[n, ci, q] from G [n, cout, lout]
    G = torch.randn(n_seq, cout, lout, generator=g).cuda()
    Gr = rnd(G, precision)
    want_dx = torch.nn.grad.conv1d_input((n_seq, cin, lin), wr.double(), Gr.double(), stride=s)    # [n, cin, lin]
    r = kw // s
    rows_dx = -(-lin // s)
    rows_pad = rows_dx + r - 1
    gpad = torch.zeros(n_seq, rows_pad, cout, dtype=act_dtype(precision), device="cuda")
    gpad[:, r - 1:r - 1 + lout] = G.transpose(1, 2).to(act_dtype(precision))
    # P[t cin + ci][u cout + co] = W[co][ci][s (r - 1 - u) + t]
    wt = torch.empty(s * cin, r * cout, device="cuda")
    for t in range(s):
        for u in range(r):
            wt[t * cin:(t + 1) * cin, u * cout:(u + 1) * cout] = w[:, :, s * (r - 1 - u) + t].t()
    frags = pack_matrix(wt, precision)
    dx = torch.full((n_seq, lin, cin), 7.0, dtype=act_dtype(precision), device="cuda")
    a = lib.PfDenseArgs()
    a.A, a.M, a.rows_per_seq, a.a_seq_stride, a.lda = gpad.data_ptr(), n_seq * rows_dx, rows_dx, rows_pad * cout, cout
    a.K, a.N, a.KC = r * cout, s * cin, 256
    a.wfrags, a.out, a.o_seq_stride, a.ldo, a.o_valid_per_seq = frags.data_ptr(), dx.data_ptr(), lin * cin, s * cin, lin * cin
    lib.check(L.pf_dense_nt(PREC[precision], 0, C.byref(a), stream()), "transposed conv as dense_nt")
    err = (dx.float().double().transpose(1, 2) - want_dx).abs().max().item()
    print(f"[transposed conv via dense_nt {precision}] {err:.2e}")
    assert err < tol(precision, 8.0)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("m,n1,n2", [(1000, 192, 768), (333, 576, 192), (64, 768, 192), (5000, 32, 64), (130, 384, 192), (97, 8, 8)])
def test_dense_tn_is_the_weight_gradient(precision, m, n1, n2):
    lib, L = _L()
    g = torch.Generator().manual_seed(m + n1)
    G = torch.randn(m, n1, generator=g).cuda()
    A = torch.randn(m, n2, generator=g).cuda()
    Gd, Ad = G.to(act_dtype(precision)).contiguous(), A.to(act_dtype(precision)).contiguous()
    dW = torch.zeros(n1, n2, device="cuda")
    db = torch.zeros(n1, device="cuda")
    a = lib.PfDenseTnArgs()
    a.G, a.g_seq_stride, a.ldg, a.A, a.a_seq_stride, a.lda = Gd.data_ptr(), 0, n1, Ad.data_ptr(), 0, n2
    a.M, a.rows_per_seq, a.N1, a.N2, a.dW, a.ldw, a.db, a.splits = m, m, n1, n2, dW.data_ptr(), n2, db.data_ptr(), 0
    lib.check(L.pf_dense_tn(PREC[precision], C.byref(a), stream()), "pf_dense_tn")
    want = rnd(G, precision).double().t() @ rnd(A, precision).double()
    e1 = (dW.double() - want).abs().max().item() / want.abs().max().item()
    e2 = (db.double() - rnd(G, precision).double().sum(0)).abs().max().item()
    print(f"\n[dense_tn {precision} {m}x{n1}x{n2}] rel {e1:.2e} bias {e2:.2e}")
    assert e1 < 1e-5 and e2 < 1e-3 * math.sqrt(m)
    # accumulates (the caller zeroes): a second call doubles
    lib.check(L.pf_dense_tn(PREC[precision], C.byref(a), stream()), "pf_dense_tn")
    assert (dW.double() - 2 * want).abs().max().item() / want.abs().max().item() < 2e-5


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_dense_tn_convolution_weight_gradient(precision):
    lib, L = _L()
    g = torch.Generator().manual_seed(2)
    n_seq, cin, cout, kw, s, lin = 7, 32, 64, 16, 4, 203
    lout = (lin - kw) // s + 1
    x = torch.randn(n_seq, cin, lin, generator=g).cuda()
    G = torch.randn(n_seq, cout, lout, generator=g).cuda()
    want = torch.nn.grad.conv1d_weight(rnd(x, precision).double(), (cout, cin, kw), rnd(G, precision).double(), stride=s)
    xp = x.transpose(1, 2).contiguous().to(act_dtype(precision))
    # the gradient rows sit in a padded image with an offset, as the stem's backward keeps them
    gp = torch.zeros(n_seq, lout + 5, cout, dtype=act_dtype(precision), device="cuda")
    gp[:, 3:3 + lout] = G.transpose(1, 2).to(act_dtype(precision))
    dW = torch.zeros(cout, cin, kw, device="cuda")
    db = torch.zeros(cout, device="cuda")
    a = lib.PfDenseTnArgs()
    a.G, a.g_seq_stride, a.ldg = gp.data_ptr() + 3 * cout * gp.element_size(), (lout + 5) * cout, cout
    a.A, a.a_seq_stride, a.lda = xp.data_ptr(), lin * cin, s * cin
    a.M, a.rows_per_seq, a.N1, a.N2, a.dW, a.ldw, a.conv_cin, a.conv_kw, a.db, a.splits = (
        n_seq * lout, lout, cout, kw * cin, dW.data_ptr(), cin * kw, cin, kw, db.data_ptr(), 3)
    lib.check(L.pf_dense_tn(PREC[precision], C.byref(a), stream()), "pf_dense_tn conv")
    err = (dW.double() - want).abs().max().item() / want.abs().max().item()
    print(f"\n[conv weight gradient via dense_tn {precision}] rel {err:.2e}")
    assert err < 1e-5
    assert (db.double() - rnd(G, precision).double().sum((0, 2))).abs().max().item() < 1e-3


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_layernorm_forward_backward(precision):
    from oracle.enc_dropout import factors
    lib, L = _L()
    g = torch.Generator().manual_seed(5)
    m = 1037
    x = (torch.randn(m, 192, generator=g) * 2 + 0.3).cuda()
    gamma, beta = (torch.rand(192, generator=g) + 0.5).cuda(), torch.randn(192, generator=g).cuda()
    y = torch.empty(m, 192, dtype=act_dtype(precision), device="cuda")
    mean, rstd = torch.empty(m, device="cuda"), torch.empty(m, device="cuda")
    a = lib.PfLnArgs()
    a.x, a.gamma, a.beta, a.M, a.y, a.mean, a.rstd = x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), m, y.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    lib.check(L.pf_enc_ln_forward(PREC[precision], C.byref(a), stream()), "ln fwd")
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    want = F.layer_norm(xr, (192,), gr, br, 1e-5)
    err = (y.float().double() - want.detach()).abs().max().item()
    print(f"\n[layernorm {precision}] fwd {err:.2e}")
    assert err < (3e-2 if precision == "bf16" else 1e-5)
    dy = torch.randn(m, 192, generator=g).cuda()
    dres = torch.randn(m, 192, generator=g).cuda()
    dyd = dy.to(act_dtype(precision)).contiguous()
    gx, gg, gb = torch.autograd.grad(want, (xr, gr, br), rnd(dy, precision).double())
    for p in (0.0, 0.2):
        dx = torch.empty(m, 192, device="cuda")
        gout = torch.empty(m, 192, dtype=act_dtype(precision), device="cuda")
        dg, db = torch.zeros(192, device="cuda"), torch.zeros(192, device="cuda")
        a.dy, a.dres, a.dx, a.gout, a.dgamma, a.dbeta = dyd.data_ptr(), dres.data_ptr(), dx.data_ptr(), gout.data_ptr(), dg.data_ptr(), db.data_ptr()
        a.drop_p, a.seed, a.site = p, 31, 2
        lib.check(L.pf_enc_ln_backward(PREC[precision], C.byref(a), stream()), "ln bwd")
        fac = torch.from_numpy(factors(p, 31, 2, m * 192)).reshape(m, 192).cuda().double()
        e = [(dx.double() - (gx + dres.double())).abs().max().item(), (dg.double() - gg).abs().max().item() / gg.abs().max().item(),
             (db.double() - gb).abs().max().item() / gb.abs().max().item(),
             (gout.float().double() - (gx + dres.double()) * fac).abs().max().item()]
        print(f"[layernorm {precision} p={p}] dx {e[0]:.2e} dgamma {e[1]:.2e} dbeta {e[2]:.2e} gout {e[3]:.2e}")
        assert e[0] < 1e-4 and e[1] < 1e-5 and e[2] < 1e-5 and e[3] < (5e-2 if precision == "bf16" else 1e-4)


def _attn_reference(qkv, T, B, fac=None):
    """torch reference on float64: [B*T, 576] -> out [B*T, 192], with optional dropout factors [B, 6, T, T]"""
    q, k, v = (qkv[:, i * 192:(i + 1) * 192].reshape(B, T, 6, 32).transpose(1, 2) for i in range(3))
    s = q @ k.transpose(-1, -2) / math.sqrt(32.0)
    p = torch.softmax(s, dim=-1)
    lse = torch.logsumexp(s, dim=-1)
    if fac is not None:
        p = p * fac
    return (p @ v).transpose(1, 2).reshape(B * T, 192), lse


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("T,B,p", [(183, 3, 0.0), (187, 2, 0.1), (61, 4, 0.0), (17, 2, 0.3), (192, 1, 0.0)])
def test_self_attention_forward_backward(precision, T, B, p):
    from oracle.enc_dropout import factors
    lib, L = _L()
    g = torch.Generator().manual_seed(T + B)
    qkv = torch.randn(B * T, 576, generator=g).cuda()
    qd = qkv.to(act_dtype(precision)).contiguous()
    out = torch.empty(B * T, 192, dtype=act_dtype(precision), device="cuda")
    lse = torch.empty(B, 6, T, device="cuda")
    a = lib.PfAttnArgs()
    a.qkv, a.B, a.T, a.out, a.lse, a.drop_p, a.seed, a.site = qd.data_ptr(), B, T, out.data_ptr(), lse.data_ptr(), p, 1234, 4
    lib.check(L.pf_enc_attn_forward(PREC[precision], C.byref(a), stream()), "attn fwd")
    fac = torch.from_numpy(factors(p, 1234, 4, B * 6 * T * 192)).reshape(B, 6, T, 192)[..., :T].cuda().double() if p > 0 else None
    xr = rnd(qkv, precision).double().requires_grad_(True)
    want, want_lse = _attn_reference(xr, T, B, fac)
    e1 = (out.float().double() - want.detach()).abs().max().item()
    e2 = (lse.double() - want_lse.detach()).abs().max().item()
    print(f"\n[attention {precision} T={T} B={B} p={p}] out {e1:.2e} lse {e2:.2e}")
    assert e1 < (3e-2 if precision == "bf16" else 2e-5) and e2 < (1e-2 if precision == "bf16" else 2e-5)
    dout = torch.randn(B * T, 192, generator=g).cuda()
    dd = dout.to(act_dtype(precision)).contiguous()
    dqkv = torch.empty(B * T, 576, dtype=act_dtype(precision), device="cuda")
    a.dout, a.dqkv = dd.data_ptr(), dqkv.data_ptr()
    lib.check(L.pf_enc_attn_backward(PREC[precision], C.byref(a), stream()), "attn bwd")
    (gw,) = torch.autograd.grad(want, xr, rnd(dout, precision).double())
    e3 = (dqkv.float().double() - gw).abs().max().item() / gw.abs().max().item()
    print(f"[attention {precision}] dqkv rel {e3:.2e}")
    assert e3 < (3e-2 if precision == "bf16" else 2e-5)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("T,B", [(183, 3), (187, 2), (61, 5)])
def test_attention_pool_forward_backward(precision, T, B):
    lib, L = _L()
    g = torch.Generator().manual_seed(T)
    kv = torch.randn(B * T, 384, generator=g).cuda()
    q = (torch.randn(8, 192, generator=g) / math.sqrt(32.0)).cuda()
    kd = kv.to(act_dtype(precision)).contiguous()
    pooled = torch.empty(B, 8, 192, device="cuda")
    a = lib.PfPoolArgs()
    a.kv, a.q, a.B, a.T, a.pooled = kd.data_ptr(), q.data_ptr(), B, T, pooled.data_ptr()
    lib.check(L.pf_enc_pool_forward(PREC[precision], C.byref(a), stream()), "pool fwd")
    kr = rnd(kv, precision).double().requires_grad_(True)
    qr = q.double().requires_grad_(True)
    k_, v_ = (kr[:, i * 192:(i + 1) * 192].reshape(B, T, 6, 32).transpose(1, 2) for i in range(2))        # [B, 6, T, 32]
    qh = qr.reshape(8, 6, 32).transpose(0, 1)                                                               # [6, 8, 32]
    pr = torch.softmax(qh[None] @ k_.transpose(-1, -2), dim=-1)                                              # [B, 6, 8, T]
    want = (pr @ v_).transpose(1, 2).reshape(B, 8, 192)
    e1 = (pooled.double() - want.detach()).abs().max().item()
    dp = torch.randn(B, 8, 192, generator=g).cuda()
    dkv = torch.empty(B * T, 384, dtype=act_dtype(precision), device="cuda")
    dq = torch.zeros(8, 192, device="cuda")
    a.dpooled, a.dkv, a.dq = dp.data_ptr(), dkv.data_ptr(), dq.data_ptr()
    lib.check(L.pf_enc_pool_backward(PREC[precision], C.byref(a), stream()), "pool bwd")
    gk, gq = torch.autograd.grad(want, (kr, qr), dp.double())
    e2 = (dkv.float().double() - gk).abs().max().item() / gk.abs().max().item()
    e3 = (dq.double() - gq).abs().max().item() / gq.abs().max().item()
    print(f"\n[pool {precision} T={T}] pooled {e1:.2e} dkv rel {e2:.2e} dq rel {e3:.2e}")
    assert e1 < 2e-5 and e2 < (1e-2 if precision == "bf16" else 2e-5) and e3 < 2e-5
