"""ctypes binding of libtabgnn_hip.so (C ABI declared in include/tabgnn_hip.h).

The product path has NO fallback: if the library is missing or a kernel reports an error the
call raises.  torch is used for device memory (caching allocator), streams and distributed only.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TABGNN_LIB_PATH") or os.path.join(_HERE, "libtabgnn_hip.so")     # (override: kernel A/B builds)
ABI_VERSION = 7

_vp, _i32, _i64, _f32, _u32, _u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint32, C.c_uint64

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/tabgnn_hip.h one to one
SIGNATURES = {
    "tg_last_error": [],
    "tg_abi_version": [],
    "tg_device_check": [],
    "tg_ids_to_i32": [_vp, _i64, _i32, _vp, _vp, _vp],
    "tg_csr_workspace_ints": [_i64, _i32],
    "tg_csr_build": [_vp, _i64, _i32, _vp, _vp, _vp, _vp],
    "tg_encode_max_cols": [],
    "tg_encode_small_table_rows": [],
    "tg_encode_bwd_blocks": [],
    "tg_encode_fwd": [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "tg_encode_bwd": [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp],
    "tg_scatter_add_segments": [_vp, _vp, _i32, _i64, _vp],
    "tg_embed_grad_sorted": [_vp, _i64, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "tg_attn_fwd": [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _u64, _u32, _i32, _vp],
    "tg_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _u64, _u32, _i32, _vp],
    "tg_ln_partials_floats": [_i64, _i32],
    "tg_ln_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _f32, _f32, _f32, _u64, _u32, _i32, _vp],
    "tg_ln_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _f32, _f32, _u64, _u32,
                  _i32, _vp, _vp, _vp, _i32, _vp],
    "tg_ln_tail_ln_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _f32, _f32, _u64,
                          _u32, _vp, _i32, _vp],
    "tg_bn_partials_floats": [_i64, _i32],
    "tg_bn_act_res_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _f32, _i32, _f32,
                          _f32, _i64, _i32, _vp, _i32, _vp],
    "tg_bn_act_res_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _f32, _i64,
                          _i32, _vp, _i32, _vp],
    "tg_act_dropout_fwd": [_vp, _vp, _i64, _i32, _f32, _u64, _u32, _i32, _vp],
    "tg_act_dropout_bwd": [_vp, _vp, _vp, _i64, _i32, _f32, _u64, _u32, _i32, _vp],
    "tg_axpby": [_vp, _vp, _vp, _i64, _f32, _f32, _i32, _vp],
    "tg_head_mlp_supported": [_i32, _i32, _i32, _i32],
    "tg_head_mlp_partial_floats": [_i32, _i32, _i32, _i32],
    "tg_head_mlp_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _u64, _u32, _u32,
                        _i32, _vp],
    "tg_head_mlp_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32,
                        _i32, _f32, _u64, _u32, _u32, _i32, _vp],
    "tg_row_head_scale": [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _f32, _f32, _i32, _vp],
    "tg_gsampler_seedbit_bytes": [_i64],
    "tg_gsampler_workspace_bytes": [_i32, _i64],
    "tg_gsampler_draw": [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _vp, _i32, _u64, _i64, _vp, _vp, _vp, _vp],
    "tg_gsampler_emit": [_vp, _i64, _i32, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp],
    "tg_axpby2": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _i32, _vp],
    "tg_col_sum_workspace_floats": [_i64, _i32],
    "tg_col_sum": [_vp, _i64, _i32, _i64, _vp, _vp, _i32, _i32, _vp],
    "tg_cls_merge_fwd": [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp],
    "tg_gather_concat3": [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _i32, _vp, _i64,
                          _i32, _vp],
    "tg_segment_hub_ints": [_i64],
    "tg_segment_sum2": [_vp, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp],
    "tg_rows_add": [_vp, _vp, _vp, _i64, _i32, _i64, _i32, _i32, _vp],
    "tg_pna_aggregate_fwd": [_vp, _vp, _vp, _vp, _i32, _i32, _i64, _vp, _i32, _vp],
    "tg_pna_aggregate_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp],
    "tg_pna_aggregate_hubs": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp],
    "tg_pna_scale_combine_fwd": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    "tg_pna_scale_combine_bwd": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    "tg_pna_degree_scalers": [_vp, _vp, _vp, _i32, _vp],
    "tg_gemm_nt_scaled_bf16": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _i32, _vp],
    "tg_gemm_tn_scaled_bf16": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _i32, _vp],
    "tg_encode_ts_features": [_vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _vp],
    "tg_pna_fold_fwd": [_vp, _vp, _i32, _i32, _vp, _vp],
    "tg_pna_fold_bwd": [_vp, _vp, _vp, _i32, _i32, _vp, _vp],
    "tg_pna_fold_ws_floats": [_i32],
    "tg_gine_aggregate_fwd": [_vp, _vp, _vp, _vp, _vp, _f32, _vp, _i32, _i32, _vp, _i32, _vp],
    "tg_gine_message_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "tg_seed_pool_fwd": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "tg_seed_pool_inplace": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "tg_seed_pool_bwd": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "tg_gemm_tn_workspace_floats": [_i64, _i32, _i32],
    "tg_gemm_nt_supported": [_i64, _i32, _i32],
    "tg_gemm_nt_bf16": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _i32, _f32, _u64, _u32, _vp],
    "tg_gemm_nt_ln_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _f32, _f32, _u64, _u32, _vp],
    "tg_gemm_tn_bf16": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _i32, _vp],
    "tg_gemm_nt_gather3_bf16": [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _vp],
    "tg_gemm_tn_gather3_workspace_floats": [_i64, _i32],
    "tg_gemm_tn_gather3_bf16": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _vp],
    "tg_encoder_pack_bytes": [],
    "tg_encoder_stage_bytes": [],
    "tg_encoder_prm_floats": [],
    "tg_encoder_fused_supported": [_i32, _i32, _i32, _i32],
    "tg_encoder_pack": [_vp] * 17,
    "tg_encoder_pack_tiles": [_vp, _vp, _i32, _vp, _vp],
    "tg_encoder_bwd_ffn_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _f32, _f32, _u64,
                                _vp, _vp, _vp],
    "tg_encoder_bwd_attn_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _i64, _i32, _i32,
                                 _f32, _f32, _f32, _u64, _vp, _vp, _vp],
    "tg_encoder_ln_partial_blocks": [_i64, _i32],
    "tg_encoder_dw_blocks": [_i64, _i32],
    "tg_encoder_bwd_ffn_dw_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _f32, _f32, _u64, _vp, _vp, _vp, _vp,
                                   _vp],
    "tg_encoder_dw_reduce": [_vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp],
    "tg_encoder_ln_reduce": [_vp, _i64, _vp, _i32, _vp],
    "tg_encoder_fwd_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _u64, _vp, _vp],
    "tg_encoder_ffn_fwd_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _f32, _f32, _f32, _u64, _vp, _vp],
    "tg_weighted_ce_fwd": [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _vp],
    "tg_weighted_ce_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _vp],
    "tg_adam_step": [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _i32, _f32, _i32, _vp],
    "tg_cast_f32_to_bf16": [_vp, _vp, _i64, _vp],
    "tg_pna_post_fwd_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i64, _i64, _vp],
    "tg_pna_post_dagg_bf16": [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _i64, _vp],
    "tg_advance_step": [_vp, _f32, _f32, _f32, _vp],
    "tg_adam_step_dev": [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _vp, _f32, _i32, _vp],
    "tg_zero": [_vp, _i64, _vp],
    "tg_transpose_batched_bf16": [_vp, _vp, _vp, _i32, _vp],
}
_RESTYPES = {"tg_last_error": C.c_char_p, "tg_csr_workspace_ints": _i64, "tg_segment_hub_ints": _i64,
             "tg_gemm_tn_workspace_floats": _i64, "tg_col_sum_workspace_floats": _i64, "tg_head_mlp_partial_floats": _i64, "tg_gsampler_seedbit_bytes": _i64, "tg_gsampler_workspace_bytes": _i64, "tg_encoder_pack_bytes": _i64, "tg_encoder_stage_bytes": _i64, "tg_encoder_ln_partial_blocks": _i64, "tg_encoder_dw_blocks": _i64, "tg_encoder_prm_floats": _i64,
             "tg_pna_fold_ws_floats": _i64,
             "tg_gemm_tn_gather3_workspace_floats": _i64}


class EncCol(C.Structure):
    _fields_ = [("kind", _i32), ("out_col", _i32), ("src_col", _i32), ("rows", _i32), ("tab_off", _i32),
                ("acc_off", _i32), ("ts_slot", _i32), ("pad", _i32)]


class EncDesc(C.Structure):
    _fields_ = [("ncol", _i32), ("nts", _i32), ("col", EncCol * 16)]


class EncPtrs(C.Structure):
    _fields_ = [("num", _vp), ("nn", _i32), ("cat", _vp), ("nc", _i32), ("ts", _vp), ("nt", _i32), ("rel", _vp),
                ("nr", _i32), ("num_mean", _vp), ("num_std", _vp), ("num_w", _vp), ("num_b", _vp),
                ("cat_table", _vp), ("ts_min_year", _vp), ("ts_w", _vp), ("ts_b", _vp), ("rel_w", _vp),
                ("rel_b", _vp), ("row_ids", _vp)]


class Gather3(C.Structure):
    """tg_gather3: three 128-column bf16 row sources, optional int32 row indices, row pitches in elements."""
    _fields_ = [("src", _vp * 3), ("idx", _vp * 3), ("stride", _i64 * 3)]


class FoldParams(C.Structure):
    _fields_ = [(k, _vp) for k in ("P", "pb", "We", "be", "Qw", "qb", "Lw", "lb")]


class FoldOut(C.Structure):
    _fields_ = [(k, _vp) for k in ("w_msg", "b_msg", "w_x", "b_eff", "w_st", "w_msg_lp", "w_msg_lp_t", "w_x_lp",
                                   "w_x_lp_t", "w_cat", "wt_cat")]


class FoldGrads(C.Structure):
    _fields_ = [(k, _vp) for k in ("dw_msg", "db_msg", "dw_x", "db_eff", "dw_st")]


class FoldDParams(C.Structure):
    _fields_ = [(k, _vp) for k in ("dP", "dpb", "dWe", "dbe", "dQw", "dqb", "dLw", "dlb", "ws")] + [("accumulate", _i32)]


_lib = None


def load():
    """dlopen the in-tree library and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C models-for-relational-multimodal-data_amd` "
            "(or __graft_entry__.build()).  There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, C.c_int)
    if lib.tg_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libtabgnn_hip ABI {lib.tg_abi_version()} != expected {ABI_VERSION}")
    _lib = lib
    return lib


_TRACE = os.environ.get("TABGNN_TRACE_CALLS") == "1"      # debugging aid: name every launch and wait for it


def call(name, *args):
    lib = load()
    if _TRACE:
        print("tg call", name, flush=True)
        rc = getattr(lib, name)(*args)
        torch.cuda.synchronize()
        if rc != 0:
            raise RuntimeError(f"{name} failed ({rc}): {lib.tg_last_error().decode()}")
        return
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib, name)
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.tg_last_error().decode()}")


_FN = {}


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be contiguous CUDA(HIP) tensors."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("tabgnn_amd ops need tensors on the MI355X (got a CPU tensor); there is no CPU path")
    if not t.is_contiguous():
        raise RuntimeError("tabgnn_amd ops need contiguous tensors")
    return t.data_ptr()


def stream():
    """Raw handle of torch's current HIP stream on the current device (the fast C getter: torch.cuda.current_stream()
    costs ~10 us per call, which is most of a launch in the small-batch regime)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def dt(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    raise RuntimeError(f"unsupported activation dtype {t.dtype} (float32 or bfloat16)")
