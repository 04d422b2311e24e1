"""ctypes binding of libpfhip.so (the C ABI in include/pf_hip.h).

The product path has no CPU fallback: if the shared library is missing this
module raises on first use, and the wrappers raise if tensors are not on the GPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpfhip.so")
if os.environ.get("PF_LIBPFHIP"):          # kernel experiments: an alternative build of the same library
    LIB_PATH = os.environ["PF_LIBPFHIP"]

PF_OK, PF_ERR_BAD_ARG, PF_ERR_UNSUPPORTED, PF_ERR_HIP = 0, -1, -2, -3
PF_PREC_F32, PF_PREC_BF16 = 0, 1
PF_FLAG_HOIST_CTX = 1
PF_FLAG_MASKED_CONTEXT = 2
PF_FLAG_WIDE = 4
PF_FLAG_BWD = 8
PF_FLAG_GENERIC = 16
PF_REDUCE_SLOTS = 16          # include/pf_hip.h
PF_EPI_PLAIN, PF_EPI_GELU, PF_EPI_RESID, PF_EPI_MUL = 0, 1, 2, 3
PRECISIONS = {"fp32": PF_PREC_F32, "f32": PF_PREC_F32, "bf16": PF_PREC_BF16}


class PfFlowBwdChainArgs(C.Structure):
    """include/pf_hip.h PfFlowBwdChainArgs (device pointers as integers)."""
    _fields_ = [("batch", C.c_int64)] + [(n, C.c_void_p) for n in (
        "WfT", "W2T", "W1T", "W0T", "U", "params", "hs", "t1s", "t2s", "gates", "pc", "g_z", "g_lad", "g_nll", "nll_z", "log_sigma",
        "Gp", "Gh0", "Gt1", "Gt2", "Gc", "g_x", "drop")] + [("compact", C.c_uint32), ("drop_scale", C.c_float),
                                                           ("packed", C.c_void_p), ("gp_ld", C.c_uint32)]


class PfFlowReevalArgs(C.Structure):
    """include/pf_hip.h PfFlowReevalArgs"""
    _fields_ = [("batch", C.c_int64)] + [(n, C.c_void_p) for n in (
        "packed", "U", "ctx", "hs", "t1s", "t2s", "gates", "pc", "h2", "params", "drop")] + [("compact", C.c_uint32)]


class PfEmbedTrainDesc(C.Structure):
    """include/pf_hip.h PfEmbedTrainDesc"""
    _fields_ = [("precision", C.c_int32), ("n_detectors", C.c_int32), ("n_extra_tokens", C.c_int32), ("training", C.c_int32),
                ("dropout_p", C.c_float), ("forward_only", C.c_int32), ("dropout_seed", C.c_uint64)]


class PfDenseArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("M", C.c_int64), ("rows_per_seq", C.c_int64), ("a_seq_stride", C.c_int64), ("lda", C.c_int32),
                ("K", C.c_int32), ("N", C.c_int32), ("KC", C.c_int32), ("wfrags", C.c_void_p), ("bias", C.c_void_p),
                ("out", C.c_void_p), ("o_seq_stride", C.c_int64), ("ldo", C.c_int32), ("o_valid_per_seq", C.c_int64),
                ("x_seq_stride", C.c_int64), ("dact", C.c_void_p), ("resid", C.c_void_p), ("mul", C.c_void_p),
                ("drop_p", C.c_float), ("seed", C.c_uint32), ("site", C.c_uint32), ("out_f32", C.c_int32),
                ("a_chunk_stride", C.c_int64), ("a_slab_chunks", C.c_int32), ("k_splits", C.c_int32), ("n_group", C.c_int32)]


class PfDenseTnArgs(C.Structure):
    _fields_ = [("G", C.c_void_p), ("g_seq_stride", C.c_int64), ("ldg", C.c_int32), ("A", C.c_void_p), ("a_seq_stride", C.c_int64),
                ("lda", C.c_int32), ("M", C.c_int64), ("rows_per_seq", C.c_int64), ("N1", C.c_int32), ("N2", C.c_int32),
                ("dW", C.c_void_p), ("ldw", C.c_int32), ("conv_cin", C.c_int32), ("conv_kw", C.c_int32), ("db", C.c_void_p),
                ("splits", C.c_int32), ("batch", C.c_int32), ("g_batch_stride", C.c_int64), ("a_batch_stride", C.c_int64),
                ("w_batch_stride", C.c_int64), ("b_batch_stride", C.c_int64), ("n1_rows", C.c_int32), ("n2_cols", C.c_int32), ("mask", C.c_void_p)]


class PfLnArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("M", C.c_int64), ("y", C.c_void_p),
                ("mean", C.c_void_p), ("rstd", C.c_void_p), ("dy", C.c_void_p), ("dres", C.c_void_p), ("dx", C.c_void_p),
                ("gout", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("drop_p", C.c_float), ("seed", C.c_uint32),
                ("site", C.c_uint32)]


class PfAttnArgs(C.Structure):
    _fields_ = [("qkv", C.c_void_p), ("B", C.c_int64), ("T", C.c_int32), ("out", C.c_void_p), ("lse", C.c_void_p),
                ("drop_p", C.c_float), ("seed", C.c_uint32), ("site", C.c_uint32), ("dout", C.c_void_p), ("dqkv", C.c_void_p)]


class PfPoolArgs(C.Structure):
    _fields_ = [("kv", C.c_void_p), ("q", C.c_void_p), ("B", C.c_int64), ("T", C.c_int32), ("pooled", C.c_void_p),
                ("dpooled", C.c_void_p), ("dkv", C.c_void_p), ("dq", C.c_void_p)]


class PfGeomArgs(C.Structure):
    _fields_ = [("clean", C.c_void_p), ("batch", C.c_int64), ("n_det", C.c_int32), ("band_lo", C.c_int32), ("nf", C.c_int32),
                ("n_bands", C.c_int32), ("maxlag", C.c_int32), ("band_edge", C.c_int32 * 17), ("twiddle", C.c_void_p),
                ("spec", C.c_void_p), ("etot", C.c_void_p), ("rel", C.c_void_p), ("sanitize", C.c_int32)]


class PfFlowDesc(C.Structure):
    _fields_ = [
        ("features", C.c_int32), ("context_features", C.c_int32),
        ("hidden_features", C.c_int32), ("num_bins", C.c_int32),
        ("num_layers", C.c_int32), ("num_blocks", C.c_int32),
        ("tail_bound", C.c_float), ("min_bin_width", C.c_float),
        ("min_bin_height", C.c_float), ("min_derivative", C.c_float),
        ("precision", C.c_int32), ("reserved", C.c_int32),
    ]


# every symbol include/pf_hip.h declares: (restype, argtypes)
_P = C.POINTER(PfFlowDesc)
SYMBOLS = {
    "pf_flow_raw_param_count": (C.c_int64, [_P]),
    "pf_flow_packed_bytes": (C.c_int64, [_P]),
    "pf_flow_pack_map_len": (C.c_int64, [_P]),
    "pf_flow_build_pack_map": (C.c_int, [_P, C.c_void_p]),
    "pf_flow_pack": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_flow_workspace_bytes": (C.c_int64, [_P, C.c_int64]),
    "pf_flow_forward": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int64, C.c_void_p]),
    "pf_flow_forward_train": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.c_void_p]),
    "pf_flow_forward_train_dropout": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_float, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p]),
    "pf_flow_dropout_mask": (C.c_int, [_P, C.c_float, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p]),
    "pf_flow_forward_reduce": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_void_p]),
    "pf_flow_rqs_backward": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_flow_backward_chain": (C.c_int, [_P, C.POINTER(PfFlowBwdChainArgs), C.c_void_p]),
    "pf_flow_reevaluate": (C.c_int, [_P, C.POINTER(PfFlowReevalArgs), C.c_void_p]),
    "pf_flow_ctx_transposed_bytes": (C.c_int64, [_P]),
    "pf_flow_pack_ctx_transposed": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_flow_inverse": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int64, C.c_void_p]),
    "pf_pack_bf16_frags": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pf_flow_inc_layer_bytes": (C.c_int64, [_P]),
    "pf_dense_pack_linear": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p]),
    "pf_diag_stream_ingest": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pf_flow_ctx_project_rows": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                           C.c_void_p, C.c_void_p]),
    "pf_flow_inverse_inc": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_embed_stem_raw_param_count": (C.c_int64, []),
    "pf_embed_stem_packed_bytes": (C.c_int64, [C.c_int32]),
    "pf_embed_stem_pack_map_len": (C.c_int64, [C.c_int32]),
    "pf_embed_stem_build_pack_map": (C.c_int, [C.c_int32, C.c_void_p]),
    "pf_embed_stem_pack": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_embed_stem_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int64]),
    "pf_embed_stem_forward": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.c_void_p]),
    "pf_embed_fusion_raw_param_count": (C.c_int64, []),
    "pf_embed_fusion_packed_bytes": (C.c_int64, []),
    "pf_embed_fusion_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_embed_fusion_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_void_p]),
    "pf_embed_train_raw_param_count": (C.c_int64, []),
    "pf_embed_train_packed_bytes": (C.c_int64, [C.c_int32]),
    "pf_embed_train_pack": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_embed_train_workspace_bytes": (C.c_int64, [C.POINTER(PfEmbedTrainDesc), C.c_int64]),
    "pf_embed_train_forward": (C.c_int, [C.POINTER(PfEmbedTrainDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pf_embed_train_backward": (C.c_int, [C.POINTER(PfEmbedTrainDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_dense_frag_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "pf_dense_pack_matrix": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pf_dense_nt": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(PfDenseArgs), C.c_void_p]),
    "pf_dense_tn": (C.c_int, [C.c_int32, C.POINTER(PfDenseTnArgs), C.c_void_p]),
    "pf_dropout_factor": (C.c_float, [C.c_float, C.c_uint32, C.c_uint32, C.c_uint32]),
    "pf_enc_ln_forward": (C.c_int, [C.c_int32, C.POINTER(PfLnArgs), C.c_void_p]),
    "pf_enc_ln_backward": (C.c_int, [C.c_int32, C.POINTER(PfLnArgs), C.c_void_p]),
    "pf_enc_attn_forward": (C.c_int, [C.c_int32, C.POINTER(PfAttnArgs), C.c_void_p]),
    "pf_enc_attn_backward": (C.c_int, [C.c_int32, C.POINTER(PfAttnArgs), C.c_void_p]),
    "pf_enc_pool_forward": (C.c_int, [C.c_int32, C.POINTER(PfPoolArgs), C.c_void_p]),
    "pf_enc_pool_backward": (C.c_int, [C.c_int32, C.POINTER(PfPoolArgs), C.c_void_p]),
    "pf_remix_workspace_bytes": (C.c_int64, [C.c_int64]),
    "pf_remix_forward": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_void_p]),
    "pf_geom_twiddles": (C.c_int, [C.c_void_p]),
    "pf_geom_features": (C.c_int, [C.POINTER(PfGeomArgs), C.c_void_p]),
    "pf_last_error": (C.c_char_p, []),
    "pf_version": (C.c_char_p, []),
    "pf_flow_rows_per_workgroup": (C.c_int32, [_P, C.c_int64]),
    "pf_flow_issued_flop_per_row": (C.c_int64, [_P]),
    "pf_flow_forward_kernel_name": (C.c_char_p, [_P, C.c_int64]),
}

_lib = None


class PfError(RuntimeError):
    pass


def lib():
    """Load libpfhip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PfError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
                "posteriflow_amd/csrc`). posteriflow_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)   # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != PF_OK:
        msg = lib().pf_last_error().decode()
        if rc == PF_ERR_BAD_ARG:
            raise ValueError(f"{what}: {msg}")
        if rc == PF_ERR_UNSUPPORTED:
            raise NotImplementedError(f"{what}: {msg}")
        raise PfError(f"{what}: HIP error: {msg}")


# ---- parameter epoch -----------------------------------------------------------------------------------------------------
# The packed-weight caches are keyed on the parameters' version counters.  torch's fused optimisers (AdamW(fused=True),
# torch._fused_adamw_) update parameters WITHOUT bumping those counters (measured: leaf and view versions 0 -> 0 across
# opt.step()), so a cache keyed on versions alone would keep serving the weights of before the step.  Every optimiser step
# anywhere in the process therefore advances a global epoch that is part of every key.  Weights edited through ``.data`` or
# ``torch.no_grad`` tricks that bypass version counting need ``invalidate_packed_caches()``.
_PARAM_EPOCH = [0]


def param_epoch() -> int:
    return _PARAM_EPOCH[0]


def invalidate_packed_caches() -> None:
    """call after changing parameters in a way autograd's version counters do not see"""
    _PARAM_EPOCH[0] += 1


def _install_optimizer_hook() -> None:
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
        register_optimizer_step_post_hook(lambda opt, args, kwargs: invalidate_packed_caches())
    except Exception:                                  # pragma: no cover -- torch without global optimiser hooks
        pass


_install_optimizer_hook()
