"""ctypes binding of libacmil_hip.so (C ABI: include/acmil_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C acmil_amd/csrc`.  There is NO
fallback: if the shared object is missing or a call returns an error code, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ACMIL_HIP_LIB", os.path.join(_HERE, "libacmil_hip.so"))  # env override: experiments only

OK = 0
ERRORS = {-1: "ACMIL_ERR_SHAPE", -2: "ACMIL_ERR_UNSUPPORTED", -3: "ACMIL_ERR_NULL", -4: "ACMIL_ERR_LAUNCH",
          -5: "ACMIL_ERR_ARCH"}
MODE_F32, MODE_F16X3, MODE_F16 = 0, 1, 2
MODES = {"fp32": MODE_F32, "f16x3": MODE_F16X3, "f16": MODE_F16}
DTYPE_F32, DTYPE_F16, DTYPE_BF16 = 0, 1, 2
MAX_TOKENS, MAX_TOKENS_FUSED, MAX_CLASSES = 16, 5, 16

_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t

# name -> (restype, argtypes); every symbol include/acmil_hip.h declares
SIGNATURES = {
    "acmil_version": (C.c_char_p, []),
    "acmil_check_device": (_i, []),
    "acmil_mfma_probe": (_i, [_i, _i, _vp, C.POINTER(C.c_longlong), _vp]),
    "acmil_ga_packed_bytes": (_sz, [_i] * 6),
    "acmil_ga_pack_weights": (_i, [_vp] * 7 + [C.POINTER(_vp), C.POINTER(_vp)] + [_vp] * 2 + [_i] * 6 + [_vp, _vp]),
    "acmil_ga_workspace_bytes": (_sz, [_i] * 6),
    "acmil_ga_workspace_init": (_i, [_vp, _vp]),
    "acmil_ga_forward": (_i, [_vp, _i, _i, _vp] + [_i] * 6 + [_vp] * 6 + [_i, _vp, _vp]),
    "acmil_ga_batch_workspace_bytes": (_sz, [_i, C.POINTER(_i)] + [_i] * 5),
    "acmil_ga_forward_batch": (_i, [_i, C.POINTER(_vp), C.POINTER(_i), _i, _vp] + [_i] * 6 + [C.POINTER(_vp)] + [_vp] * 4 +
                               [_i, _vp, _vp]),
    "acmil_ga_forward_guarded": (_i, [_i, C.POINTER(_vp), C.POINTER(_i), _i, _vp, _vp] + [_i] * 5 + [C.POINTER(_vp)] + [_vp] * 4 +
                                 [_i, _vp, _vp, _vp]),
    "acmil_ga_forward_guarded_wide_scratch_bytes": (_sz, [_i] * 6),
    "acmil_ga_rescore_fp32_cond_scratch_bytes": (_sz, [_i] * 4),
    "acmil_ga_rescore_fp32_cond": (_i, [_vp, _i, _i, _vp, _vp] + [_i] * 6 + [_vp] * 6),
    "acmil_ga_forward_guarded_wide": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "acmil_ga_pool": (_i, [_vp, _vp, _i, _vp] + [_i] * 6 + [_vp, _i] + [_vp] * 4 + [_i, _vp, _vp]),
    "acmil_ga_pool_group_workspace_bytes": (_sz, [_i] * 4),
    "acmil_ga_pool_group": (_i, [_vp, _vp, _i, _i, C.POINTER(_i), _vp] + [_i] * 6 + [_vp] * 4 + [_i, _vp, _vp]),
    "acmil_stkim_workspace_bytes": (_sz, [_i] * 3),
    "acmil_stkim_select": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "acmil_stkim_select_rng": (_i, [_vp, _i, _i, _i, _i, _vp, C.c_ulonglong, C.c_ulonglong, _vp, _vp, _vp, _vp]),
    "acmil_gemm_workspace_bytes": (_sz, [_i] * 4),
    "acmil_gemm_f32": (_i, [_i, _i, _i, _i, _i, C.c_float, _vp, _i, C.c_longlong, _vp, _i, _i, C.c_longlong, C.c_float,
                            _vp, _i, C.c_longlong, _vp, _i, _vp, _i, _vp, _vp]),
    "acmil_gemm_f16x3": (_i, [_i, _i, _i, _i, _i, C.c_float, _vp, _i, C.c_longlong, _vp, _i, _i, C.c_longlong, C.c_float,
                              _vp, _i, C.c_longlong, _vp, _i, _vp, _i, _vp, _vp]),
    "acmil_gemm_bf16x3": (_i, [_i, _i, _i, _i, _i, C.c_float, _vp, _i, C.c_longlong, _vp, _i, _i, C.c_longlong, C.c_float,
                               _vp, _i, C.c_longlong, _vp, _i, _vp, _i, _vp, _vp]),
    "acmil_linear_packed_bytes": (_sz, [_i, _i]),
    "acmil_linear_pack": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "acmil_linear_f16x3": (_i, [_vp, _i, _i, _i, C.c_longlong, _vp, _i, _vp, _i, C.c_float, _vp, C.c_longlong, _vp, _vp]),
    "acmil_gated_scores_packed": (_i, [_vp, _i, _i, _i, C.c_longlong, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "acmil_mha_workspace_bytes": (_sz, [_i] * 5),
    "acmil_mha_forward": (_i, [_vp] + [_i] * 5 + [_vp, _vp] + [C.POINTER(_vp)] * 4 + [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "acmil_gated_scores_workspace_bytes": (_sz, [_i] * 4),
    "acmil_gated_scores": (_i, [_vp] + [_i] * 4 + [_vp] * 6 + [_i, _vp, _vp, _vp]),
    "acmil_attn_pool_workspace_bytes": (_sz, [_i] * 3),
    "acmil_attn_pool": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "acmil_softmax_rows": (_i, [_vp, _vp, _i, _i, _vp]),
    "acmil_attn_row_stats": (_i, [_vp, _i, _i, _vp, _vp]),
    "acmil_attn_heatmap": (_i, [_vp, _i, _i, C.c_float, _vp, _vp, _vp]),
    "acmil_layernorm_fwd": (_i, [_vp, C.c_longlong, _i, _vp, _vp, C.c_float, _vp, _vp, _vp]),
    "acmil_layernorm_bwd_workspace_bytes": (_sz, [C.c_longlong, _i]),
    "acmil_layernorm_bwd": (_i, [_vp, _vp, _vp, _vp, C.c_longlong, _i, _vp, _vp, _vp, _vp, _vp]),
    "acmil_softmax_rows_bwd": (_i, [_vp, _vp, _vp, C.c_longlong, _i, _vp]),
    "acmil_seqconv": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "acmil_seqconv_bwd_w_workspace_bytes": (_sz, [_i, _i]),
    "acmil_seqconv_bwd_w": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "acmil_dwconv7": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "acmil_dwconv7_bwd_w_workspace_bytes": (_sz, [_i, _i]),
    "acmil_dwconv7_bwd_w": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "acmil_landmark_mean": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "acmil_landmark_mean_bwd": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "acmil_relu_bwd": (_i, [_vp, _vp, _vp, C.c_longlong, _vp]),
    "acmil_colsum_workspace_bytes": (_sz, [C.c_longlong, _i]),
    "acmil_colsum": (_i, [_vp, C.c_longlong, _i, _vp, _vp, _vp]),
    "acmil_gate_fwd": (_i, [_vp, _vp, C.c_longlong, _i, _vp]),
    "acmil_gate_bwd": (_i, [_vp, _vp, _vp, C.c_longlong, _i, _vp]),
    "acmil_adamw_step": (_i, [_vp, _vp, _vp, _vp, C.c_longlong, C.c_float, C.c_double, C.c_double, C.c_float, C.c_float,
                              C.c_longlong, _vp, _vp, _vp]),
    "acmil_adamw_step_report": (_i, [_vp, _vp, _vp, _vp, C.c_longlong, C.c_float, C.c_double, C.c_double, C.c_float, C.c_float,
                                     C.c_longlong, _vp, _vp, _vp, _vp]),
    "acmil_peer_publish": (_i, [_vp, _vp, C.c_longlong, C.POINTER(_vp), _i, _i, C.c_uint, _vp, _vp]),
    "acmil_adamw_step_peer": (_i, [_vp, _vp, _vp, C.c_longlong, C.POINTER(_vp), _vp, _i, _i, C.c_uint, C.c_double, _vp, C.c_float,
                                   C.c_double, C.c_double, C.c_float, C.c_float, C.c_longlong, _i, _vp, _vp, _vp, _vp]),
    "acmil_ga_loss_workspace_bytes": (_sz, [_i] * 2),
    "acmil_ga_loss": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "acmil_transmil_workspace_bytes": (_sz, [_i] * 4),
    "acmil_transmil_forward": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "acmil_transmil_forward_ex": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                       _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "acmil_ga_backward_workspace_bytes": (_sz, [_i] * 5),
    "acmil_ga_backward": (_i, [_vp, _i, _i] + [_vp] * 8 + [C.POINTER(_vp)] + [_vp] * 4 + [_vp] * 7 +
                          [C.POINTER(_vp), C.POINTER(_vp)] + [_vp, _vp] + [_i] * 6 + [_vp, _vp]),
    "acmil_ga_train_step_workspace_bytes": (_sz, [_i] * 6),
    # x, x_dtype, N, packed, repack | W1 Wv bv Wu bu Ww bw Wc[] bc[] Ws bs | dW1 dWv dbv dWu dbu dWw dbw dWc[] dbc[] dWs dbs |
    # D Di Da K C mode | label uniforms k_top m_mask | losses sub slide A_out topk midx guard_flag | workspace stream
    "acmil_ga_train_step": (_i, [_vp, _i, _i, _vp, _i] + [_vp] * 7 + [C.POINTER(_vp), C.POINTER(_vp), _vp, _vp] + [_vp] * 7 +
                            [C.POINTER(_vp), C.POINTER(_vp), _vp, _vp] + [_i] * 6 + [_vp, _vp, _i, _i] + [_vp] * 7 + [_vp, _vp]),
    "acmil_ga_train_step_rng": (_i, [_vp, _i, _i, _vp, _i] + [_vp] * 7 + [C.POINTER(_vp), C.POINTER(_vp), _vp, _vp] + [_vp] * 7 +
                            [C.POINTER(_vp), C.POINTER(_vp), _vp, _vp] + [_i] * 6 + [_vp, _vp, _i, _i] + [_vp] * 7 + [_vp, _vp] +
                                [C.c_ulonglong, C.c_ulonglong]),
    "acmil_ga_train_step_adamw": (_i, [_vp, _i, _i, _vp, _i] + [_vp] * 7 + [C.POINTER(_vp), C.POINTER(_vp), _vp, _vp] + [_vp] * 7 +
                                  [C.POINTER(_vp), C.POINTER(_vp), _vp, _vp] + [_i] * 6 + [_vp, _vp, _i, _i] + [_vp] * 7 + [_vp, _vp] +
                                  [C.c_ulonglong, C.c_ulonglong] +
                                  [_vp, C.c_longlong, _vp, _vp, C.c_float, C.c_double, C.c_double, C.c_float, C.c_float, C.c_longlong, _vp, _vp]),
    "acmil_ga_train_step_group_workspace_bytes": (_sz, [_i] * 7),
    # x, x_dtype, nbags, bag_rows (host), packed, repack | parameters | gradients | D Di Da K C mode | labels uniforms k_top m_mask |
    # losses sub slide A_out topk midx guard_flag | workspace stream | seed offset | adamw (struct acmil_adamw_args* or NULL)
    "acmil_ga_train_step_group": (_i, [_vp, _i, _i, C.POINTER(_i), _vp, _i] + [_vp] * 7 + [C.POINTER(_vp), C.POINTER(_vp), _vp, _vp] + [_vp] * 7 +
                                  [C.POINTER(_vp), C.POINTER(_vp), _vp, _vp] + [_i] * 6 + [_vp, _vp, _i, _i] + [_vp] * 7 + [_vp, _vp] +
                                  [C.c_ulonglong, C.c_ulonglong, _vp]),
    "acmil_ga_adamw_supported": (_i, [_vp] * 7 + [C.POINTER(_vp), C.POINTER(_vp), _vp, _vp] + [_vp] * 3 + [_i] * 6 + [_vp, C.c_longlong, _vp, _vp]),
    "acmil_ga_adamw_pack": (_i, [_vp] * 7 + [C.POINTER(_vp), C.POINTER(_vp), _vp, _vp] + [_vp] * 7 + [C.POINTER(_vp), C.POINTER(_vp), _vp, _vp] +
                            [_i] * 6 + [_vp] + [_vp, C.c_longlong, _vp, _vp, C.c_float, C.c_double, C.c_double, C.c_float, C.c_float, C.c_longlong,
                                                _vp, _vp, _vp, _vp]),
}



class AdamwArgs(C.Structure):
    """struct acmil_adamw_args (include/acmil_hip.h)"""
    _fields_ = [("flat_params", _vp), ("n_flat", C.c_longlong), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("lr", C.c_float),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_float), ("weight_decay", C.c_float), ("step", C.c_longlong),
                ("skipped", _vp), ("flag_report", _vp)]


_lib = None


def load() -> C.CDLL:
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libacmil_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C acmil_amd/csrc`; there is no CPU/PyTorch fallback for the aggregation path" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != OK:
        raise RuntimeError("%s failed: %s (%d)" % (what, ERRORS.get(rc, "unknown"), rc))
