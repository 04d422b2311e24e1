"""The [B, .]-sized dense layers on the hand-written GEMMs (posteriflow_amd/_dense.py): forward and every gradient against
float64 tensor arithmetic, for the shapes the strain embedding's head uses (reference src/ahsd/models/lean_npe.py:181-197,
242-252; coherent_encoder.py:73-77, 120-121) and ragged ones; the context projections of the incremental inverse
(pf_flow_ctx_project_rows)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # (rows, K, N, bias)
    (1000, 48, 64, True),        # energy_mlp.0 / noise_mlp.0: K padded to 64
    (1024, 64, 32, True),        # noise_mlp.2: N padded to 64
    (1024, 1600, 512, True),     # out_proj.0: 25 k-chunks, two column groups, split reduction
    (4096, 1632, 512, True),     # out_proj.0 with psd bands: K padded to 1664
    (333, 512, 256, True),       # out_proj.2, ragged rows
    (8, 192, 192, True),         # the pool's query projection: 8 rows
    (8 * 77, 192, 192, True),    # the pool's output projection
    (300, 201, 128, True),       # geom_mlp.0: K padded to 256
    (300, 128, 768, False),      # geom_to_tokens-like, 48 tiles in one pass, no bias
    (1, 288, 96, True),          # one row
]


@pytest.mark.parametrize("rows,k,n,has_bias", SHAPES)
def test_linear_forward_and_gradients_match_float64(rows, k, n, has_bias):
    from posteriflow_amd import _dense
    torch.manual_seed(rows + k + n)
    dev = torch.device("cuda")
    x = torch.randn(rows, k, device=dev, requires_grad=True)
    w = (torch.randn(n, k, device=dev) / k ** 0.5).requires_grad_(True)
    b = torch.randn(n, device=dev, requires_grad=True) if has_bias else None
    state = {}
    y = _dense.linear(x, w, b, state, "t")
    g = torch.randn_like(y)
    grads = torch.autograd.grad(y, [x, w] + ([b] if has_bias else []), g)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    bd = b.detach().double().requires_grad_(True) if has_bias else None
    yd = xd @ wd.t() + (bd if has_bias else 0.0)
    want = torch.autograd.grad(yd, [xd, wd] + ([bd] if has_bias else []), g.double())
    scale = lambda t: max(1.0, t.abs().max().item())
    err_y = (y.double() - yd).abs().max().item() / scale(yd)
    assert y.shape == (rows, n) and err_y < 2e-6, err_y          # fp32 MFMA, fp32 accumulation (measured <= 5e-7)
    for name, got, ref in zip(("dx", "dw", "db"), grads, want):
        err = (got.double() - ref).abs().max().item() / scale(ref)
        assert got.shape == ref.shape and err < 4e-6, (name, err)
    # second call: the packed fragments are reused; after an in-place weight update they are rebuilt
    packed = state["t"][1]
    _dense.linear(x, w, b, state, "t")
    assert state["t"][1] is packed
    with torch.no_grad():
        w.mul_(2.0)
    y2 = _dense.linear(x, w, b, state, "t")
    assert state["t"][1] is not packed
    yd2 = x.detach().double() @ w.detach().double().t() + (b.detach().double() if has_bias else 0.0)
    assert (y2.double() - yd2).abs().max().item() / scale(yd2) < 2e-6


def test_linear_bf16_operands_and_leading_dimensions():
    from posteriflow_amd import _dense
    torch.manual_seed(3)
    dev = torch.device("cuda")
    x = torch.randn(5, 7, 288, device=dev)
    w = torch.randn(160, 288, device=dev) / 17.0
    b = torch.randn(160, device=dev)
    y = _dense.linear(x, w, b, {}, "t", precision="bf16")
    ref = x.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().t() + b.double()     # the kernel's operand rounding
    assert y.shape == (5, 7, 160) and y.dtype == torch.float32
    assert (y.double() - ref).abs().max().item() < 2e-5
    assert _dense.linear(x[:0], w, b, {}, "t").shape == (0, 7, 160)
    with pytest.raises(Exception):
        _dense.linear(x.cpu(), w.cpu(), b.cpu(), {}, "t")


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("rows,c,units", [(1, 288, 8 * 3 * 256), (70, 288, 2 * 3 * 64), (200, 40, 3 * 128)])
def test_context_projection_rows_match_tensor_arithmetic(precision, rows, c, units):
    """pf_flow_ctx_project_rows = the operand of pf_flow_inverse_inc: [rows][units] raw affine values"""
    from posteriflow_amd import _lib
    torch.manual_seed(rows + c)
    dev = torch.device("cuda")
    L = _lib.lib()
    prec = _lib.PRECISIONS[precision]
    kw = 32 if precision == "bf16" else 16
    kpad = -(-c // kw) * kw
    w = torch.randn(units, c, device=dev) / c ** 0.5
    b = torch.randn(units, device=dev)
    ctx = torch.randn(rows, c, device=dev)
    wp = torch.zeros(units, kpad, device=dev)
    wp[:, :c] = w
    frags = torch.empty(L.pf_dense_frag_bytes(prec, units, kpad), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(L.pf_dense_pack_matrix(prec, wp.data_ptr(), 0, kpad, units, kpad, frags.data_ptr(), s), "pack")
    out = torch.full((rows, units), float("nan"), device=dev)
    _lib.check(L.pf_flow_ctx_project_rows(prec, frags.data_ptr(), b.data_ptr(), ctx.data_ptr(), rows, c, units, out.data_ptr(), s), "proj")
    if precision == "bf16":
        ref = ctx.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().t() + b.double()
    else:
        ref = ctx.double() @ w.double().t() + b.double()
    err = (out.double() - ref).abs().max().item()
    assert err < 1e-5, err
    assert L.pf_flow_ctx_project_rows(prec, frags.data_ptr(), b.data_ptr(), ctx.data_ptr(), rows, c, units + 8, out.data_ptr(), s) == _lib.PF_ERR_BAD_ARG
    assert L.pf_flow_ctx_project_rows(prec, None, b.data_ptr(), ctx.data_ptr(), rows, c, units, out.data_ptr(), s) == _lib.PF_ERR_BAD_ARG
    assert L.pf_flow_ctx_project_rows(prec, None, None, None, 0, c, units, None, s) == _lib.PF_OK
