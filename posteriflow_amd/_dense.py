"""``y = x W^T + b`` for the [B, .]-sized dense layers of the path, on the hand-written GEMMs (no vendor library).

The strain embedding's small dense layers -- ``energy_mlp`` / ``noise_mlp`` / ``out_proj`` of ``LeanStrainEncoder``
(reference ``src/ahsd/models/lean_npe.py:181-197, 242-252``), the attention pool's output projection (``:228-233``),
``geom_mlp`` / ``geom_to_tokens`` of ``CoherentEncoder`` (``src/ahsd/models/coherent_encoder.py:73-77, 120-121``) -- and the
context projections of the incremental inverse run through the same two kernels as the encoder's training path:

* forward and data gradient: ``pf_dense_nt`` (strip GEMM, ``csrc/pf_dense.hip``) over the weight packed as MFMA fragments
  (``pf_dense_pack_matrix``; the plain form for the forward, the transposed form for ``dX = G W``), packed once per weight
  update;
* weight / bias gradient: ``pf_dense_tn`` (``dW += G^T x``, ``db += sum G``, float atomics into zeroed fp32 buffers).

Shapes the kernels do not take directly are zero-padded here: the reduction length to a multiple of 64 (the k-chunk staged in
LDS), the output width to a multiple of 64 (it is the reduction length of the data gradient); a reduction that needs several
chunks is limited to 256 output units per launch, so wider outputs go out in column groups.  Few rows with a long reduction
(``out_proj``: 1600 -> 512 at 1024 rows = 8 strips) are split over workgroups (``k_splits``: partial sums added into an
output that already holds the bias).

Activations (GELU) stay elementwise tensor ops between the linears.  Everything here is device work; the module raises off
the GPU."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib

_KC_MAX = {_lib.PF_PREC_F32: 192, _lib.PF_PREC_BF16: 384}      # k-chunk (128 rows x KC operands in LDS, + staging, <= 160 KB)
_NG = 256                                                       # output units per launch of a chunked reduction


def _ceil(v: int, m: int) -> int:
    return -(-v // m) * m


def _stream(dev) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _kc(kp: int, prec: int) -> int:
    """largest multiple of 64 that divides kp and fits the LDS image"""
    best = 64
    for kc in range(64, min(kp, _KC_MAX[prec]) + 1, 64):
        if kp % kc == 0:
            best = kc
    return best


def _pad_cols(t: torch.Tensor, width: int, dtype) -> torch.Tensor:
    """[M, width] contiguous copy of t [M, k <= width] in ``dtype``, zero-filled on the right (no copy when nothing changes)"""
    if t.shape[1] == width and t.dtype == dtype and t.is_contiguous():
        return t
    if t.shape[1] == width:
        return t.to(dtype).contiguous()
    out = torch.zeros(t.shape[0], width, dtype=dtype, device=t.device)
    out[:, : t.shape[1]] = t
    return out


class _Packed:
    """MFMA fragments of one weight matrix [N, K] (zero-padded to [Np, Kp]): forward groups of <= 256 output units and the
    transposed form for the data gradient, rebuilt when the weight changes."""

    def __init__(self, weight: torch.Tensor, prec: int, want_transposed: bool):
        L, dev = _lib.lib(), weight.device
        n, k = weight.shape
        self.n, self.k, self.np_, self.kp = n, k, _ceil(n, 64), _ceil(k, 64)
        self.prec = prec
        wp = torch.zeros(self.np_, self.kp, dtype=torch.float32, device=dev)
        wp[:n, :k] = weight.detach().float()
        self.kc_f = _kc(self.kp, prec)
        self.kc_t = _kc(self.np_, prec)
        s = _stream(dev)

        def pack(src, mode, ld, N, K):
            nbytes = L.pf_dense_frag_bytes(prec, N, K)
            if nbytes < 0:
                raise NotImplementedError(f"pf_dense_frag_bytes({N}, {K})")
            out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(L.pf_dense_pack_matrix(prec, src.data_ptr(), mode, ld, N, K, out.data_ptr(), s), "pf_dense_pack_matrix")
            return out

        # forward: out[:, g0:g0+ng] = x_p W_p[g0:g0+ng]^T -- one group when the reduction is a single chunk
        gw = self.np_ if self.kp == self.kc_f else _NG
        self.fwd = [(g0, min(gw, self.np_ - g0), pack(wp[g0:], 0, self.kp, min(gw, self.np_ - g0), self.kp))
                    for g0 in range(0, self.np_, gw)]
        # data gradient: dX[:, g0:g0+ng] = G_p (W_p[:, g0:g0+ng]): the packed matrix is W_p^T restricted to those columns
        self.bwd = None
        if want_transposed:
            gw = self.kp if self.np_ == self.kc_t else _NG
            self.bwd = [(g0, min(gw, self.kp - g0), pack(wp[:, g0:], 1, self.kp, min(gw, self.kp - g0), self.np_))
                        for g0 in range(0, self.kp, gw)]
        self._keep = wp     # (pack kernels are asynchronous: the padded copy must outlive them; it is small)


def _packed(state: Dict, name: str, weight: torch.Tensor, prec: int, want_transposed: bool) -> _Packed:
    key = (_lib.param_epoch(), weight.device, prec, weight._version, weight.data_ptr(), tuple(weight.shape))
    ent = state.get(name)
    if ent is None or ent[0] != key or (want_transposed and ent[1].bwd is None):
        ent = (key, _Packed(weight, prec, want_transposed))
        state[name] = ent
    return ent[1]


def _nt(prec: int, a_t: torch.Tensor, K: int, kc: int, frags: torch.Tensor, N: int, bias_ptr: int, out: torch.Tensor,
        col0: int, k_splits: int) -> None:
    """out[:, col0:col0+N] (fp32, row stride out.shape[1]) (+)= a_t[:, :K] . W^T (+ bias)"""
    a = _lib.PfDenseArgs()
    M = a_t.shape[0]
    a.A, a.M, a.rows_per_seq, a.a_seq_stride, a.lda = a_t.data_ptr(), M, max(M, 1), 0, a_t.shape[1]
    a.K, a.N, a.KC = K, N, kc
    a.wfrags, a.bias = frags.data_ptr(), bias_ptr
    a.out, a.o_seq_stride, a.ldo = out.data_ptr() + 4 * col0, 0, out.shape[1]
    a.out_f32, a.k_splits = 1, k_splits
    _lib.check(_lib.lib().pf_dense_nt(prec, _lib.PF_EPI_PLAIN, C.byref(a), _stream(a_t.device)), "pf_dense_nt")


def _splits(m: int, n_groups: int, nchunks: int) -> int:
    """workgroups along the reduction: enough to reach ~one round of the chip when the strips alone do not"""
    strips = -(-m // 128) * n_groups
    if nchunks <= 1 or strips >= 128:
        return 1
    return max(1, min(nchunks, 256 // strips))


def _gemm(prec: int, a_t: torch.Tensor, groups, K: int, kc: int, n_out: int, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """[M, n_out] fp32 = a_t . W^T (+ bias) over the packed column groups"""
    M, dev = a_t.shape[0], a_t.device
    ks = _splits(M, len(groups), K // kc)
    if ks > 1:          # partial sums are ADDED into the output: it starts as the bias (or zero)
        out = torch.zeros(M, n_out, dtype=torch.float32, device=dev) if bias is None else bias.expand(M, n_out).contiguous()
    else:
        out = torch.empty(M, n_out, dtype=torch.float32, device=dev)
    for g0, ng, frags in groups:
        _nt(prec, a_t, K, kc, frags, ng, 0 if (bias is None or ks > 1) else bias.data_ptr() + 4 * g0, out, g0, ks)
    return out


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, state, name, prec):
        need_bwd = any(ctx.needs_input_grad[:3])
        pk = _packed(state, name, weight, prec, need_bwd)
        adt = torch.bfloat16 if prec == _lib.PF_PREC_BF16 else torch.float32
        xp = _pad_cols(x.detach(), pk.kp, adt)
        bp = None
        if bias is not None:
            bp = torch.zeros(pk.np_, dtype=torch.float32, device=x.device)
            bp[: pk.n] = bias.detach().float()
        out = _gemm(prec, xp, pk.fwd, pk.kp, pk.kc_f, pk.np_, bp)
        ctx.pk, ctx.has_bias, ctx.adt, ctx.prec = pk, bias is not None, adt, prec
        ctx.save_for_backward(xp)
        return out[:, : pk.n] if pk.n != pk.np_ else out

    @staticmethod
    def backward(ctx, g):
        pk, prec = ctx.pk, ctx.prec
        (xp,) = ctx.saved_tensors
        L, dev = _lib.lib(), g.device
        gp = _pad_cols(g, pk.np_, ctx.adt)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _gemm(prec, gp, pk.bwd, pk.np_, pk.kc_t, pk.kp, None)
            gx = gx[:, : pk.k] if pk.k != pk.kp else gx
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw = torch.zeros(pk.n, pk.k, dtype=torch.float32, device=dev)
            gb = torch.zeros(pk.n, dtype=torch.float32, device=dev) if ctx.has_bias else None
            t = _lib.PfDenseTnArgs()
            M = gp.shape[0]
            t.G, t.g_seq_stride, t.ldg = gp.data_ptr(), 0, pk.np_
            t.A, t.a_seq_stride, t.lda = xp.data_ptr(), 0, pk.kp
            t.M, t.rows_per_seq, t.N1, t.N2 = M, max(M, 1), pk.np_, pk.kp
            t.dW, t.ldw, t.db, t.splits = gw.data_ptr(), pk.k, 0 if gb is None else gb.data_ptr(), 0
            t.n1_rows, t.n2_cols = pk.n, pk.k
            _lib.check(L.pf_dense_tn(prec, C.byref(t), _stream(dev)), "pf_dense_tn")
        return gx, gw, gb, None, None, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], state: Dict, name: str,
           precision: str = "fp32") -> torch.Tensor:
    """``x [..., K] @ weight[N, K]^T + bias`` -> fp32 ``[..., N]`` on the HIP GEMMs, differentiable in x, weight and bias.
    ``state``: a dict owned by the calling module (packed fragments live there, keyed by ``name``); ``precision``: the
    operand type ("bf16": operands rounded to bf16, fp32 accumulation and output)."""
    if x.device.type != "cuda":
        raise _lib.PfError(f"posteriflow_amd dense layers run on the MI355X only (input on {x.device}); no CPU fallback")
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if x2.shape[0] == 0:
        return x.new_zeros(*lead, weight.shape[0], dtype=torch.float32)
    y = _LinearFn.apply(x2, weight, bias, state, name, _lib.PRECISIONS[precision])
    return y.reshape(*lead, weight.shape[0])


def sequential(mod: torch.nn.Sequential, x: torch.Tensor, state: Dict, name: str) -> torch.Tensor:
    """an ``nn.Sequential`` of Linear / elementwise modules with its Linears on the HIP GEMMs (fp32 operands)"""
    for i, m in enumerate(mod):
        if isinstance(m, torch.nn.Linear):
            x = linear(x, m.weight, m.bias, state, f"{name}.{i}")
        else:
            x = m(x)
    return x
