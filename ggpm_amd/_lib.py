"""ctypes binding of libggpm_hip.so (include/ggpm_hip.h).  Fails loudly: there is no CPU fallback.

The library is built in-tree by ``python -m ggpm_amd.build`` (``__graft_entry__.build()``).  Loading it does
not need a GPU (hipcc cross-compiles; the CPU tests check that every declared symbol is exported), calling
a compute entry point does.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_size_t, c_void_p, POINTER

from . import build as _build

_LIB = None

# name -> (restype, [argtypes]); mirrors include/ggpm_hip.h one to one
P = c_void_p
I = c_int
SIGNATURES = {
    "ggpm_version": (I, []),
    "ggpm_error_string": (c_char_p, [I]),
    "ggpm_padded_hidden": (I, [I]),
    "ggpm_padded_to_csr": (I, [P, I, I, P, P, P]),
    "ggpm_csr_transpose": (I, [P, P, I, I, P, P, P, P]),
    "ggpm_extract_column": (I, [P, I, I, I, P, P]),
    "ggpm_gemm_workspace_bytes": (c_size_t, [I, I, I]),
    "ggpm_gemm": (I, [I, I, I, I, I, P, I, P, I, P, I, I, P, I, I, I, P, c_size_t, P]),
    "ggpm_gemm_grouped": (I, [I, I, I, I, I, I, P, P]),                   # problems: ggpm_gemm_problem[count]
    "ggpm_gemm_grouped_splitk_workspace_bytes": (c_size_t, [I, I, I, I]),
    "ggpm_gemm_grouped_splitk": (I, [I, I, I, I, I, I, P, P, c_size_t, P]),
    "ggpm_gemm_ksegments": (I, [I, I, I, I, P, P, P, P, P, P, I, I, P, I, I, I, P]),
    "ggpm_gemm_tn_bf16": (I, [I, I, I, P, I, P, I, P, I, P, c_size_t, P]),
    "ggpm_gemm_tn_bf16_applies": (I, [I, I, I]),
    "ggpm_colsum": (I, [P, I, I, I, P, P, P]),
    "ggpm_act_backward": (I, [P, P, I, I, I, I, I, P, P]),
    "ggpm_segment_sum": (I, [P, I, P, P, I, I, P, I, I, I, P]),
    "ggpm_gather_rows": (I, [P, I, P, I, I, P, I, I, I, P]),
    "ggpm_scatter_rows": (I, [P, I, P, I, I, P, I, I, P]),
    "ggpm_adam_step": (I, [P, P, P, P, c_size_t, c_float, c_float, c_float, c_float, c_float, I, P]),
    "ggpm_onehot": (I, [P, I, I, P, I, I, I, P]),
    "ggpm_embed_graph": (I, [P, I, P, I, I, I, I, P, I, P, I, P]),
    "ggpm_level_gate_dtype": (I, [I]),
    "ggpm_level_bf16_storage": (I, [I, I]),
    "ggpm_backward_skip_x_sums": (None, [I]),
    "ggpm_gru_backward_stashes": (I, [P, I, I, I, POINTER(c_void_p), POINTER(c_void_p)]),
    "ggpm_lstm_backward_stashes": (I, [P, I, I, I, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p)]),
    "ggpm_sum_slots": (I, [P, I, c_size_t, P, P]),
    "ggpm_gru_pack_floats": (c_size_t, [I]),
    "ggpm_gru_forward": (I, [I, I, I, P, P, P, P, I, P, I, P, P, I, P, P, P, P, P, P, P, P, P, P, I, P]),
    "ggpm_gru_backward_workspace_bytes": (c_size_t, [I, I, I]),
    "ggpm_gru_backward": (I, [I, I, I, P, P, I, P, I, P, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P,
                              P, I, P, I, P, P, I, P, c_size_t, I, P]),
    "ggpm_gru_weight_grads": (I, [I, I, I, P, P, P, P, c_size_t, P, I, P, I, P, P, I, P]),
    "ggpm_gru_sparse_forward": (I, [I, I, I, P, P, P, P, P, P, I, P, I, P, P, I, P, P, P, P, P, P, P, P, P, P, I, P]),
    "ggpm_gru_sparse_backward": (I, [I, I, I, P, P, P, I, P, I, P, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P,
                                     P, I, P, I, P, P, I, P, c_size_t, P]),
    "ggpm_backward_defer_stash": (None, [P, P, P, P]),
    "ggpm_level_prefer_narrow": (None, [I]),
    "ggpm_forward_gather_state": (None, [P, P, P]),
    "ggpm_backward_scatter_state": (None, [P, P, P]),
    "ggpm_weights_packed": (None, [I]),
    "ggpm_weight_grads_stacked_workspace_bytes": (c_size_t, [I, I]),
    "ggpm_gru_weight_grads_stacked": (I, [I, I, I, P, P, P, P, P, P, P, I, P, I, P, P, I, P, c_size_t, P]),
    "ggpm_lstm_weight_grads_stacked": (I, [I, I, I, P, P, P, P, P, P, P, I, P, I, P, I, P, I, P, c_size_t, P]),
    "ggpm_decode_steps_forward": (I, [P, P, P, P, P, P, P, P, P, c_size_t, P, P, P]),
    "ggpm_decode_steps_backward": (I, [P, P, P, P, P, P, P, P, c_size_t, P, P, P, P, c_size_t, P, P, P, c_size_t, P, P]),
    "ggpm_decode_steps_forward_async": (I, [P, P, P, P, P, P, P, P, P, c_size_t, P, P, P]),
    "ggpm_decode_steps_backward_async": (I, [P, P, P, P, P, P, P, P, c_size_t, P, P, P, P, c_size_t, P, P, P, c_size_t, P, P]),
    "ggpm_decode_join": (I, []),
    "ggpm_lstm_pack_floats": (c_size_t, [I]),
    "ggpm_lstm_forward": (I, [I, I, I, P, P, P, P, P, I, P, I, P, I, P, I, P, P, P, P, P, P, P, P, P, P, P, I, P]),
    "ggpm_lstm_backward_workspace_bytes": (c_size_t, [I, I, I]),
    "ggpm_lstm_backward": (I, [I, I, I, P, P, I, P, I, P, I, P, I, P, P, P, P, P, P, P, P, P, P, P, P, P,
                               P, P, P, P, P, I, P, I, P, I, P, I, P, c_size_t, I, P]),
    "ggpm_lstm_weight_grads": (I, [I, I, I, P, P, P, c_size_t, P, I, P, I, P, I, P, I, P]),
    "ggpm_lstm_sparse_forward": (I, [I, I, I, P, P, P, P, P, P, P, P, I, P, I, P, I, P, I, P, P, P, P, P, P, P, P, P, P,
                                     P, I, P]),
    "ggpm_lstm_sparse_backward": (I, [I, I, I, P, P, P, I, P, I, P, I, P, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P,
                                      P, P, P, P, P, P, P, I, P, I, P, I, P, I, P, c_size_t, P]),
    "ggpm_rsample_forward": (I, [P, P, P, I, I, P, P, P]),
    "ggpm_rsample_backward": (I, [P, P, P, P, P, I, I, P, P, P]),
    "ggpm_softmax_ce": (I, [P, I, I, I, P, I, P, P, P, P, I, P, P, P]),
    "ggpm_bce_logits": (I, [P, P, I, P, P, P, P]),
    "ggpm_scale_rows": (I, [P, I, I, I, P, P]),
    "ggpm_head_accuracies": (I, [P, P, P, P, I, P, I, P, I, P, I, I, I, I, P, P]),
    "ggpm_dropout": (I, [P, I, I, I, ctypes.c_float, ctypes.c_uint, ctypes.c_uint, I, P]),
    "ggpm_encoder_saved_bytes": (c_size_t, [P]),
    "ggpm_encoder_work_bytes": (c_size_t, [P]),
    "ggpm_encoder_forward": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, c_size_t, P, P, P, P, P, P]),
    "ggpm_encoder_backward": (I, [P, P, P, P, P, c_size_t, P, P, P, P, P, P, P, P, P, c_size_t, I, P, P]),
    "ggpm_tree_level_saved_floats": (c_size_t, [P]),
    "ggpm_tree_level_work_bytes": (c_size_t, [P]),
    "ggpm_tree_level_forward": (I, [P, P, c_size_t, P, P]),
    "ggpm_tree_level_backward": (I, [P, P, P, P, P, P, c_size_t, P, P]),
    "ggpm_linear_wgrads_batch": (I, [I, P, P, c_size_t, P, P]),
    "ggpm_timing_enable": (I, [I]),
    "ggpm_timing_collect": (I, [I, POINTER(c_int), POINTER(c_double), POINTER(c_double)]),
    # host-only decode schedule (csrc/schedule.hip): in: ggpm_sched_in*
    "ggpm_schedule_build": (P, [P]),
    "ggpm_schedule_get": (I, [P, c_char_p, POINTER(c_void_p), POINTER(ctypes.c_int64), POINTER(c_int), POINTER(c_int),
                              POINTER(ctypes.c_int64)]),
    "ggpm_schedule_pack": (I, [P, I, POINTER(c_void_p), POINTER(ctypes.c_int64)]),
    "ggpm_schedule_names": (I, [P, c_char_p, ctypes.c_int64]),
    "ggpm_schedule_directory": (I, [P, P, ctypes.c_int64]),
    "ggpm_schedule_free": (None, [P]),
}


def lib_path() -> str:
    # GGPM_LIB_PATH: an alternative build of the same sources (A/B runs of compile-time tuning switches)
    return os.environ.get("GGPM_LIB_PATH") or _build.LIB_PATH


def load(build_if_missing: bool = True):
    """Load (building first if the .so is absent) and type every entry point."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the ONE HIP runtime in the process (the
    # stream and every device pointer we are handed come from it).  Importing torch first makes our
    # library's libamdhip64.so.7 dependency resolve to the copy torch already loaded; the other order
    # leaves two runtimes in one process and every launch fails.
    import torch  # noqa: F401
    path = lib_path()
    if not os.path.exists(path):
        if not build_if_missing:
            raise RuntimeError("ggpm_amd: %s is missing; run `python -m ggpm_amd.build` (no CPU fallback exists)" % path)
        _build.build(verbose=False)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header / library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().ggpm_error_string(code).decode()
        raise RuntimeError("ggpm_amd: %s failed: %s (code %d)" % (what, msg, code))


_ARRAY_TYPES: dict = {}


def array_type(base, n: int):
    """``base * n``, remembered.  ctypes only keeps array types weakly: once the last instance is gone the type (a class, i.e.
    a reference cycle) waits for Python's collector and the next ``base * n`` builds it again -- tens of microseconds on the
    thread that issues the launches, several times per training step."""
    t = _ARRAY_TYPES.get((base, n))
    if t is None:
        t = _ARRAY_TYPES[(base, n)] = base * n
    return t
