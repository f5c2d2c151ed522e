"""ctypes binding of libmftx.so (include/mftx.h).  There is no CPU fallback: if
the HIP library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libmftx.so"

MAX_CANDIDATES = 16
NUM_RAFT_WEIGHTS = 34
LOOKUP_CONVC1_WEIGHT_BYTES = 393216
FLOW_BRANCH_WEIGHT_BYTES = 352256
FLOW_HEAD_WEIGHT_BYTES = 32768
OU_HEADS_WTILE_BYTES = 6635520     # MFTX_OU_HEADS_WTILE_BYTES
SPLIT_LIMIT = 65504.0      # MFTX_SPLIT_LIMIT: operands of the split arithmetic must stay below it in magnitude


class ConvDesc(C.Structure):
    _fields_ = [("a0", C.c_void_p), ("lda0", C.c_int), ("c0", C.c_int),
                ("a1", C.c_void_p), ("lda1", C.c_int), ("c1", C.c_int),
                ("wpk", C.c_void_p), ("bias", C.c_void_p),
                ("out", C.c_void_p), ("ldo", C.c_int),
                ("P", C.c_int), ("h", C.c_int), ("w", C.c_int),
                ("N", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("act", C.c_int), ("out_scale", C.c_float),
                ("addend", C.c_void_p), ("ld_addend", C.c_int),
                ("stride", C.c_int), ("hin", C.c_int), ("win", C.c_int), ("pad_y", C.c_int), ("pad_x", C.c_int),
                ("residual_mode", C.c_int), ("arith", C.c_int), ("a_split", C.c_int), ("out_split", C.c_int)]


_PP = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/mftx.h one to one
SIGNATURES = {
    "mftx_version": (C.c_int, []),
    "mftx_last_error_string": (C.c_char_p, []),
    "mftx_profile_begin": (C.c_int, []),
    "mftx_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]),
    "mftx_corr_pyramid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mftx_corr_pyramid_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mftx_corr_pyramid_layout": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "mftx_corr_lookup": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p]),
    "mftx_fmap_pyramid": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] * 4),
    "mftx_corr_lookup_ondemand": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p, C.c_int, C.c_void_p]),
    "mftx_raft_set_ondemand": (C.c_int, [C.c_void_p, C.c_int]),
    "mftx_raft_workspace_bytes_for": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mftx_conv2d": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "mftx_conv2d_tile": (C.c_int, [C.POINTER(ConvDesc), C.c_int, C.c_void_p]),
    "mftx_encoder_set_split_weights": (C.c_int, [C.c_void_p, _PP, C.c_int]),
    "mftx_raft_set_split_weights": (C.c_int, [C.c_void_p, _PP, C.c_int]),
    "mftx_raft_arith": (C.c_int, [C.c_void_p]),
    "mftx_raft_set_lookup_fused": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mftx_raft_set_flow_fused": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mftx_raft_set_tile_weights": (C.c_int, [C.c_void_p, _PP, C.c_int]),
    "mftx_raft_set_flow_head": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mftx_raft_set_ou_heads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mftx_pack_ou_heads_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mftx_ou_heads": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int, C.c_void_p]),
    "mftx_pack_flow_head_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mftx_flow_head": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 8),
    "mftx_pack_tile_conv_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mftx_tile_conv2d": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p]),
    "mftx_pack_flow_branch_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mftx_flow_branch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mftx_raft_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mftx_raft_set_coords_trace": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mftx_gru_half": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mftx_tile_conv_fills_chip": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "mftx_raft_clear_graphs": (C.c_int, [C.c_void_p]),
    "mftx_raft_set_nonfinite_counter": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mftx_raft_graph_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "mftx_pack_lookup_convc1_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mftx_corr_lookup_convc1": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p, C.c_void_p, C.c_void_p,
                                                                            C.c_int, C.c_int, C.c_void_p]),
    "mftx_split_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "mftx_count_not_below": (C.c_int, [C.c_void_p, C.c_longlong, C.c_float, C.c_void_p, C.c_void_p]),
    "mftx_raft_create": (C.c_int, [_PP, C.c_int, C.POINTER(C.c_void_p)]),
    "mftx_raft_destroy": (None, [C.c_void_p]),
    "mftx_raft_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "mftx_raft_workspace_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.c_int]),
    "mftx_raft_workspace_layout_for": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.c_int]),
    "mftx_raft_refine": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "mftx_raft_refine_gather": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          _PP, _PP, _PP, _PP, C.c_void_p,
                                          C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_size_t, C.c_void_p]),
    "mftx_encoder_create": (C.c_int, [_PP, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mftx_encoder_destroy": (None, [C.c_void_p]),
    "mftx_encoder_set_graph": (C.c_int, [C.c_void_p, C.c_int]),
    "mftx_encoder_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "mftx_encoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    "mftx_convex_upsample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 7
                             + [C.c_void_p] * 5),
    "mftx_chain": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int] + [C.c_void_p] * 4),
    "mftx_warp_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mftx_select": (C.c_int, [C.c_int, _PP, _PP, _PP, C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 5),
    "mftx_chain_select": (C.c_int, [C.c_int] + [_PP] * 6 + [C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 5),
    "mftx_chain_select_packed": (C.c_int, [C.c_int] + [_PP] * 4 + [C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 5),
    "mftx_quantize_workspace_bytes": (C.c_size_t, []),
    "mftx_quantize_u16": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mftx_dequantize_u16": (C.c_int, [C.c_void_p, C.c_longlong, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "mftx_png_unfilter": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mftx_copy_bytes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
}

_lib = None


class MftxError(RuntimeError):
    pass


def load():
    """Load libmftx.so (built by ``__graft_entry__.build()`` / ``make -C mft_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("MFTX_LIB", LIB_PATH))
    if not path.exists():
        raise MftxError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback for the MFT hot path)")
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().mftx_last_error_string().decode(errors="replace")
        raise MftxError(f"{what} failed ({code}): {msg}")


def ptr_array(ptrs):
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    return C.cast(arr, _PP), arr
