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
PRECISIONS = {"fp32": PF_PREC_F32, "f32": PF_PREC_F32, "bf16": PF_PREC_BF16}


class PfFlowBwdChainArgs(C.Structure):
    """include/pf_hip.h PfFlowBwdChainArgs (device pointers as integers)."""
    _fields_ = [("batch", C.c_int64)] + [(n, C.c_void_p) for n in (
        "WfT", "W2T", "W1T", "W0T", "U", "params", "hs", "t1s", "t2s", "gates", "pc", "g_z", "g_lad", "g_nll", "nll_z", "log_sigma",
        "Gp", "Gh0", "Gt1", "Gt2", "Gc", "g_x", "drop")] + [("compact", C.c_uint32), ("drop_scale", C.c_float),
                                                           ("packed", C.c_void_p)]


class PfFlowReevalArgs(C.Structure):
    """include/pf_hip.h PfFlowReevalArgs"""
    _fields_ = [("batch", C.c_int64)] + [(n, C.c_void_p) for n in (
        "packed", "U", "ctx", "hs", "t1s", "t2s", "gates", "pc", "h2", "params", "drop")] + [("compact", C.c_uint32)]


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
    "pf_flow_inverse": (C.c_int, [_P, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int64, C.c_void_p]),
    "pf_pack_bf16_frags": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pf_flow_inc_layer_bytes": (C.c_int64, [_P]),
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
    "pf_remix_workspace_bytes": (C.c_int64, [C.c_int64]),
    "pf_remix_forward": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_void_p]),
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
