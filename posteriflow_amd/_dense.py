"""``y = x W^T + b`` for the [B, .]-sized dense layers of the path, on the hand-written GEMMs (no vendor library).

The strain embedding's small dense layers -- ``energy_mlp`` / ``noise_mlp`` / ``out_proj`` of ``LeanStrainEncoder``
(reference ``src/ahsd/models/lean_npe.py:181-197, 242-252``), the attention pool's output projection (``:228-233``),
``geom_mlp`` / ``geom_to_tokens`` of ``CoherentEncoder`` (``src/ahsd/models/coherent_encoder.py:73-77, 120-121``) -- and the
context projections of the incremental inverse run through the same two kernels as the encoder's training path:

* forward and data gradient: ``pf_dense_nt`` (strip GEMM, ``csrc/pf_dense.hip``) over the weight packed as MFMA fragments
  (``pf_dense_pack_matrix``; the plain form for the forward, the transposed form for ``dX = G W``), packed once per weight
  update;
* weight / bias gradient: ``pf_dense_tn`` (``dW += G^T x``, ``db += sum G``, float atomics into zeroed fp32 buffers).

Shapes the kernels do not take directly are zero-padded here: the reduction length to a multiple of 64 that splits into few
k-chunks (the chunk staged in LDS), the output width likewise (it is the reduction length of the data gradient).  One launch
per GEMM: the strip kernel divides wide outputs and few-row problems into column groups over its grid (``n_group``, chosen by
the library), so the forward is deterministic -- no split reduction, no atomics (``out_proj``: 1600 -> 512 at 1024 rows = 8
strips x 8 column groups).

Activations (GELU) stay elementwise tensor ops between the linears.  Everything here is device work; the module raises off
the GPU."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib

_KC_MAX = {_lib.PF_PREC_F32: 192, _lib.PF_PREC_BF16: 256}      # k-chunk: the strip kernel stages <= 256 operands per row, and
#                                                                 128 rows x KC fp32 + the epilogue's staging must fit 160 KB


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


def _padded(k: int, prec: int) -> int:
    """reduction length the operands are zero-padded to: a multiple of 64 that splits into few k-chunks (a chunk costs a
    staging pass and two barriers, a padded column a wasted multiply: 1600 -> 1728 = 9 x 192 rather than 25 x 64)"""
    best, best_cost = None, None
    for kp in range(_ceil(k, 64), _ceil(k, 64) + 256, 64):
        cost = kp + 32 * (kp // _kc(kp, prec))
        if best is None or cost < best_cost:
            best, best_cost = kp, cost
    return best


def _pad_cols(t: torch.Tensor, width: int, dtype) -> torch.Tensor:
    """[M, width] contiguous copy of t [M, k <= width] in ``dtype``, zero-filled on the right (no copy when nothing changes)"""
    if t.shape[1] == width and t.dtype == dtype and t.is_contiguous():
        return t
    if t.shape[1] == width:
        return t.to(dtype).contiguous()
    return torch.nn.functional.pad(t.to(dtype), (0, width - t.shape[1]))       # one kernel


class _Packed:
    """MFMA fragments of one weight matrix [N, K] (zero-padded to [Np, Kp]): the plain form for the forward and the
    transposed form for the data gradient, rebuilt when the weight changes."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], prec: int, want_transposed: bool):
        L, dev = _lib.lib(), weight.device
        n, k = weight.shape
        self.n, self.k, self.np_, self.kp = n, k, _padded(n, prec), _padded(k, prec)
        self.prec = prec
        self.kc_f = _kc(self.kp, prec)
        self.kc_t = _kc(self.np_, prec)
        w = weight.detach()
        if w.dtype != torch.float32 or w.stride(1) != 1:
            w = w.float().contiguous()
        one = L.pf_dense_frag_bytes(prec, self.np_, self.kp)
        if one < 0:
            raise NotImplementedError(f"pf_dense_frag_bytes({self.np_}, {self.kp})")
        # ONE launch: the forward form and (behind it) the transposed form, zero-extended inside the pack kernel
        buf = torch.empty(one * (2 if want_transposed else 1), dtype=torch.uint8, device=dev)
        _lib.check(L.pf_dense_pack_linear(prec, w.data_ptr(), w.stride(0), n, k, self.np_, self.kp, 1 if want_transposed else 0,
                                          buf.data_ptr(), _stream(dev)), "pf_dense_pack_linear")
        self.fwd = buf[:one]                                                     # W_p   [Np][Kp]: forward
        self.bwd = buf[one:] if want_transposed else None                        # W_p^T [Kp][Np]: dX = G W
        self._keep = w      # (the pack kernel is asynchronous)
        self.bias = None
        if bias is not None:
            b = bias.detach().float()
            self.bias = b if self.np_ == n else torch.nn.functional.pad(b, (0, self.np_ - n))


def _packed(state: Dict, name: str, weight: torch.Tensor, bias: Optional[torch.Tensor], prec: int,
            want_transposed: bool) -> _Packed:
    key = (_lib.param_epoch(), weight.device, prec, weight._version, weight.data_ptr(), tuple(weight.shape),
           None if bias is None else (bias._version, bias.data_ptr()))
    ent = state.get(name)
    if ent is None or ent[0] != key or (want_transposed and ent[1].bwd is None):
        ent = (key, _Packed(weight, bias, prec, want_transposed))
        state[name] = ent
    return ent[1]


def _gemm(prec: int, a_t: torch.Tensor, frags: torch.Tensor, K: int, kc: int, n_out: int, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """[M, n_out] fp32 = a_t[:, :K] . W^T (+ bias): one pf_dense_nt launch"""
    M, dev = a_t.shape[0], a_t.device
    out = torch.empty(M, n_out, dtype=torch.float32, device=dev)
    a = _lib.PfDenseArgs()
    a.A, a.M, a.rows_per_seq, a.a_seq_stride, a.lda = a_t.data_ptr(), M, max(M, 1), 0, a_t.shape[1]
    a.K, a.N, a.KC = K, n_out, kc
    a.wfrags, a.bias = frags.data_ptr(), 0 if bias is None else bias.data_ptr()
    a.out, a.o_seq_stride, a.ldo, a.out_f32 = out.data_ptr(), 0, n_out, 1
    _lib.check(_lib.lib().pf_dense_nt(prec, _lib.PF_EPI_PLAIN, C.byref(a), _stream(dev)), "pf_dense_nt")
    return out


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, state, name, prec):
        need_bwd = any(ctx.needs_input_grad[:3])
        pk = _packed(state, name, weight, bias, prec, need_bwd)
        adt = torch.bfloat16 if prec == _lib.PF_PREC_BF16 else torch.float32
        xp = _pad_cols(x.detach(), pk.kp, adt)
        out = _gemm(prec, xp, pk.fwd, pk.kp, pk.kc_f, pk.np_, pk.bias)
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
