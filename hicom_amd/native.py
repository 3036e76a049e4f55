"""ctypes binding of libhicom_hip.so (the C ABI declared in include/hicom_hip.h).

The HIP library is the product: there is NO CPU or PyTorch fallback.  `lib()` raises if the
shared object is missing or was built for a different ABI version; every wrapper raises
`HicomNativeError` on a non-zero status, with the library's own message.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# HICOM_NATIVE_LIB: dev override (instrumented builds from tools/); the product loads the in-tree library
LIB_PATH = os.environ.get("HICOM_NATIVE_LIB") or os.path.join(HERE, "libhicom_hip.so")
ABI_VERSION = 15

DT_BF16, DT_F32, DT_F16 = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_GELU_TANH = 0, 1, 2

EXPORTS = (
    "hicom_abi_version", "hicom_last_error", "hicom_local_attn_fwd", "hicom_local_attn_bwd", "hicom_trilinear_pool_fwd",
    "hicom_linear_fwd", "hicom_fold_query_fwd", "hicom_split_bf16_fwd", "hicom_global_stream_fwd",
    "hicom_global_stream_nparts", "hicom_global_merge_fwd", "hicom_global_combine_fwd",
    "hicom_readout_gemm_fwd", "hicom_scatter_rows_fwd", "hicom_fold_query_split_fwd",
    "hicom_global_combine_strided_fwd", "hicom_compressor_workspace_bytes", "hicom_compressor_zero_prefix_bytes", "hicom_compressor_is_fused",
    "hicom_compressor_fwd2", "hicom_compressor_takes_shard4", "hicom_compressor_handoff_failures", "hicom_cast16_fwd", "hicom_global_dx_fwd",
    "hicom_compressor_fwd", "hicom_linear_to_rows_fwd", "hicom_fused_stream_fwd", "hicom_fused_stream_nparts",
    "hicom_planes_gemm_fwd", "hicom_row_ln_fwd", "hicom_small_mha_fwd", "hicom_place_blocks_fwd",
    "hicom_global_stream_bwd", "hicom_readout16_gemm_fwd", "hicom_to_f16_fwd", "hicom_merge_vproj_fwd",
    "hicom_dense16_gemm_fwd", "hicom_ln_stream_fwd", "hicom_to_f16_padded_fwd", "hicom_clip_query_prep_fwd", "hicom_inv_norm_fwd",
    "hicom_global_stream_clip_fwd", "hicom_splice_rows_fwd", "hicom_splice_labels_fwd",
    "hicom_query_prep_fwd", "hicom_query_prep_state_bytes", "hicom_partials_sum_fwd", "hicom_l2norm_stream_fwd", "hicom_local_attn_adapt_fwd",
    "hicom_small_mha_scaled_fwd", "hicom_merge_vproj_fixed_fwd", "hicom_dense16_tn_fwd", "hicom_dense16_tn_splits",
    "hicom_local_attn_adapt_bwd", "hicom_adapt_dy_fwd", "hicom_gelu_split_fwd", "hicom_gelu_bwd_fwd", "hicom_colsum_fwd",
    "hicom_global_stream_marg_fwd", "hicom_global_stream_marg_width", "hicom_global_stream_has_marg", "hicom_global_merge_marg_fwd",
    "hicom_act_rows_fwd", "hicom_act_bwd_rows_fwd", "hicom_readout16_gemm_role_fwd", "hicom_r16_chain_state_bytes", "hicom_dense16_gemm_pair_fwd", "hicom_gemv_chain_fwd",
    "hicom_merge_vproj_sets_fwd", "hicom_readout_tail_fwd", "hicom_readout_tail_state_bytes",
    "hicom_compressor_ctx16_offset",
)

PHASE_STREAM, PHASE_FINISH, PHASE_MERGE_ON_NEXT, PHASE_NEXT_IS_MAIN = 1, 2, 4, 8


class HicomNativeError(RuntimeError):
    pass


class Axis(C.Structure):
    """hicom_axis: one axis of the window tiling (include/hicom_hip.h)."""
    _fields_ = [("n", C.c_int32), ("k", C.c_int32), ("nwin", C.c_int32), ("nfull", C.c_int32)]


class AuxGemv(C.Structure):
    """hicom_aux_gemv (include/hicom_hip.h): a single-row linear layer that rides in a readout GEMM's launch."""
    _fields_ = [("xs", C.c_void_p), ("x_parts", C.c_int32), ("x_stride", C.c_int64), ("xb", C.c_void_p), ("w", C.c_void_p),
                ("b", C.c_void_p), ("res", C.c_void_p), ("N", C.c_int32), ("K", C.c_int32), ("act", C.c_int32), ("y", C.c_void_p),
                ("w_dt", C.c_int32), ("b_dt", C.c_int32), ("rows_dst", C.c_void_p), ("rows_dt", C.c_int32), ("rows_reps", C.c_int32),
                ("rows_ld", C.c_int64), ("rows_row0", C.c_int64), ("x_fixed", C.c_void_p), ("x_fixed_clear", C.c_int32)]


class R16Role(C.Structure):
    """hicom_r16_role (include/hicom_hip.h): what the workgroups behind a readout GEMM's tile grid do."""
    _fields_ = [("kind", C.c_int32), ("gemv", AuxGemv), ("gemv2", AuxGemv), ("chain_state", C.c_void_p),
                ("part_m", C.c_void_p), ("part_l", C.c_void_p), ("part_acc", C.c_void_p), ("part_dt", C.c_int32), ("nparts", C.c_int32),
                ("rows", C.c_int32), ("rows_pad", C.c_int32), ("E", C.c_int32), ("w_v", C.c_void_p), ("o_fix", C.c_void_p),
                ("out_ml", C.c_void_p), ("out_ctx", C.c_void_p), ("ctx_unnorm", C.c_int32),
                ("part_marg", C.c_void_p), ("vpe_f16", C.c_void_p), ("marg_slots", C.c_int32)]


class R16Gemm(C.Structure):
    """hicom_r16_gemm (include/hicom_hip.h)."""
    _fields_ = [("a", C.c_void_p), ("w", C.c_void_p), ("b", C.c_void_p), ("b_dt", C.c_int32), ("M", C.c_int32), ("N", C.c_int32),
                ("K", C.c_int32), ("act", C.c_int32), ("out_f16", C.c_void_p), ("y", C.c_void_p), ("y_dt", C.c_int32),
                ("ldy", C.c_int64), ("row0", C.c_int64), ("nl_group", C.c_int32)]


ROLE_NONE, ROLE_GEMV, ROLE_MERGE_VPROJ, ROLE_GEMV_CHAIN = 0, 1, 2, 3


class Adaptor(C.Structure):
    """hicom_compressor_args.ak / .av: one k / v adaptor of the local stage (include/hicom_hip.h)."""
    _fields_ = [("w0", C.c_void_p), ("b0", C.c_void_p), ("w2_f16", C.c_void_p), ("b2", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("alpha", C.c_void_p), ("y", C.c_void_p)]


class Injector(C.Structure):
    """hicom_compressor_args::hicom_injector (include/hicom_hip.h)."""
    _fields_ = [("mode", C.c_int32), ("guide", C.c_void_p), ("guide_rows", C.c_int32),
                ("c_w0", C.c_void_p), ("c_b0", C.c_void_p), ("c_w2", C.c_void_p), ("c_b2", C.c_void_p), ("c_hidden", C.c_int32),
                ("wq", C.c_void_p), ("bq", C.c_void_p), ("wk", C.c_void_p), ("bk", C.c_void_p), ("wv", C.c_void_p), ("bv", C.c_void_p),
                ("wo", C.c_void_p), ("bo", C.c_void_p), ("nheads", C.c_int32), ("ln_w", C.c_void_p), ("ln_b", C.c_void_p), ("eps", C.c_float),
                ("visual", C.c_void_p)]


class CompressorArgs(C.Structure):
    """hicom_compressor_args (include/hicom_hip.h) -- field order and types must match the header."""
    _fields_ = [
        ("ff", C.c_void_p), ("fe", C.c_void_p),
        ("T", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("E", C.c_int32),
        ("has_local", C.c_int32), ("has_global", C.c_int32), ("phases", C.c_int32), ("hidden", C.c_int32),
        ("at", Axis), ("ay", Axis), ("ax", Axis),
        ("lq", C.c_void_p), ("lq_dt", C.c_int32), ("l2norm", C.c_int32), ("lq_stride", C.c_int64),
        ("l_scale", C.c_float), ("l_bias", C.c_float),
        ("lw0", C.c_void_p), ("lb0", C.c_void_p), ("lw2", C.c_void_p), ("lb2", C.c_void_p),
        ("lw0_f16", C.c_void_p), ("lw2_f16", C.c_void_p),
        ("gq", C.c_void_p), ("nq", C.c_int32), ("nh", C.c_int32), ("n_global_rows", C.c_int32), ("P", C.c_int32),
        ("wq", C.c_void_p), ("bq", C.c_void_p), ("wk", C.c_void_p), ("wv", C.c_void_p), ("bv", C.c_void_p),
        ("wo", C.c_void_p), ("bo", C.c_void_p),
        ("gw0", C.c_void_p), ("gb0", C.c_void_p), ("gw2", C.c_void_p), ("gb2", C.c_void_p),
        ("pe", C.c_void_p), ("kpe", C.c_void_p), ("pe_hi", C.c_void_p), ("pe_lo", C.c_void_p),
        ("t_index0", C.c_int32), ("y_index0", C.c_int32), ("x_index0", C.c_int32), ("nsets", C.c_int32),
        ("out", C.c_void_p), ("out_dt", C.c_int32), ("nl_group", C.c_int32),
        ("ldo", C.c_int64), ("local_row0", C.c_int64), ("global_row0", C.c_int64),
        ("newline", C.c_void_p), ("newline_dt", C.c_int32), ("nl_count", C.c_int32),
        ("nl_first", C.c_int64), ("nl_step", C.c_int64),
        ("local_out", C.c_void_p), ("state_out", C.c_void_p), ("state_sets", C.c_void_p),
        ("state_set_stride", C.c_int64),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
        ("stream_main", C.c_void_p), ("stream_side", C.c_void_p), ("ev_fork", C.c_void_p), ("ev_join", C.c_void_p),
        ("ev_merge", C.c_void_p), ("defer_join", C.c_int32), ("reserved_", C.c_int32),
        ("place_src", C.c_void_p), ("place_block_stride", C.c_int64), ("place_block_rows", C.c_int32), ("place_nblocks", C.c_int32),
        ("ev_done", C.c_void_p), ("stream_next", C.c_void_p),
        ("gc0", C.c_void_p), ("local_logits", C.c_void_p), ("reuse_queries", C.c_int32),
        ("ak", Adaptor), ("av", Adaptor), ("adapt_alpha_dt", C.c_int32), ("adapt_eps", C.c_float),
        ("r0_buf", C.c_void_p),
        ("gq_dt", C.c_int32),
        ("ev_queries", C.c_void_p),
        ("inj_l", Injector), ("inj_g", Injector),
        ("vpe_f16", C.c_void_p), ("marg_slots", C.c_int32),
        ("ag_fn", C.c_void_p), ("ag_comm", C.c_void_p), ("ag_send", C.c_void_p), ("ag_recv", C.c_void_p), ("ag_bytes", C.c_int64),
        ("ag_group_start", C.c_void_p), ("ag_group_end", C.c_void_p), ("ag_send2", C.c_void_p), ("ag_recv2", C.c_void_p), ("ag_bytes2", C.c_int64),
    ]


_LIB: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise HicomNativeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  hicom_amd has no non-HIP execution path.")
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise HicomNativeError(f"{LIB_PATH} does not export {name}; rebuild it")
    L.hicom_abi_version.restype = C.c_int
    L.hicom_last_error.restype = C.c_char_p
    if L.hicom_abi_version() != ABI_VERSION:
        raise HicomNativeError(f"ABI mismatch: library {L.hicom_abi_version()} vs binding {ABI_VERSION}; rebuild")
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.hicom_local_attn_fwd.argtypes = [vp, i32, vp, i32, i32, Axis, Axis, Axis, vp, i32, i64, f32, f32, i32, vp, vp, vp]
    L.hicom_local_attn_bwd.argtypes = [vp, vp, i32, Axis, Axis, Axis, vp, i32, i64, f32, f32, vp, vp, vp, i32, vp, vp, i32, vp]
    L.hicom_global_dx_fwd.argtypes = [vp, vp, i64, vp, vp, vp, i32, i64, i32, vp, i32, vp]
    L.hicom_trilinear_pool_fwd.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    L.hicom_linear_fwd.argtypes = [vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    L.hicom_fold_query_fwd.argtypes = [vp, vp, i32, i32, i32, f32, vp, vp]
    L.hicom_split_bf16_fwd.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    L.hicom_global_stream_fwd.argtypes = [vp, i64, i32, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, i64,
                                          vp, vp, vp, i32, vp]
    L.hicom_global_stream_nparts.argtypes = [i64, i32]
    L.hicom_global_stream_marg_fwd.argtypes = [vp, i64, i32, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, i64,
                                               vp, vp, vp, vp, i32, vp]
    L.hicom_global_stream_marg_width.argtypes = [i32, i32]
    L.hicom_global_stream_has_marg.argtypes = [i64, i32, i32, i32, i32, i32]
    L.hicom_global_merge_marg_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i64, i32, i32, vp, i32, i32, i32, vp, vp, vp, i32, vp]
    L.hicom_global_stream_bwd.argtypes = [vp, i64, i32, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, i64, vp, vp,
                                          vp, vp, vp, i32, vp]
    L.hicom_global_merge_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, i64, i64, i32, i32, vp, i32, i32, i32,
                                         vp, vp, vp, i32, vp]
    L.hicom_linear_to_rows_fwd.argtypes = [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, i32, i64, i64, i32, vp]
    L.hicom_fused_stream_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, i32, f32, f32, vp, i32, vp, vp, i32, i32, i32,
                                         vp, vp, vp, i32, vp, vp, vp, vp, vp, i64, vp, vp, i32, vp]
    L.hicom_readout16_gemm_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, i64, i64, i32, C.POINTER(AuxGemv), vp]
    L.hicom_readout16_gemm_role_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, i64, i64, i32, C.POINTER(R16Role), vp]
    L.hicom_r16_chain_state_bytes.argtypes = [i32]
    L.hicom_gemv_chain_fwd.argtypes = [C.POINTER(R16Role), vp]
    L.hicom_merge_vproj_sets_fwd.argtypes = [vp, i64, i32, i32, i32, vp, vp, vp, vp, vp]
    L.hicom_r16_chain_state_bytes.restype = i64
    L.hicom_readout_tail_fwd.argtypes = [C.POINTER(R16Gemm), C.POINTER(R16Gemm), C.POINTER(R16Role), C.POINTER(R16Role), vp, vp]
    L.hicom_readout_tail_state_bytes.argtypes = []
    L.hicom_readout_tail_state_bytes.restype = i64
    L.hicom_compressor_ctx16_offset.argtypes = [C.POINTER(CompressorArgs)]
    L.hicom_compressor_ctx16_offset.restype = i64
    L.hicom_to_f16_fwd.argtypes = [vp, i32, vp, i64, vp]
    L.hicom_splice_rows_fwd.argtypes = [vp, i64, i32, vp, vp]
    L.hicom_splice_labels_fwd.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, i64, vp, vp, vp]
    L.hicom_to_f16_padded_fwd.argtypes = [vp, i32, i64, i64, vp, i64, vp]
    L.hicom_dense16_gemm_fwd.argtypes = [vp, i64, vp, i64, i32, vp, i32, i32, i32, i32, i32, vp, i64, i32, vp, i64, vp, i32, i64, vp, i64, vp,
                                         vp, i64, i32, i32, i32, i32, i32, vp, i32, vp, vp]
    L.hicom_dense16_gemm_pair_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, i32, i32, i64, i32, i64, vp]
    L.hicom_partials_sum_fwd.argtypes = [vp, i32, i64, vp, vp]
    L.hicom_dense16_tn_splits.argtypes = [i32, i32, i64]
    L.hicom_dense16_tn_fwd.argtypes = [vp, i64, vp, i64, i32, i64, i32, i32, vp, i64, i32, vp]
    L.hicom_l2norm_stream_fwd.argtypes = [vp, vp, i64, i32, vp]
    L.hicom_local_attn_adapt_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, i32, Axis, Axis, Axis, vp, i32, i64, f32, f32, vp, vp]
    L.hicom_local_attn_adapt_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, i32, Axis, Axis, Axis, vp, i32, i64, f32, f32,
                                             vp, vp, vp, vp, vp, vp, vp, vp]
    L.hicom_adapt_dy_fwd.argtypes = [vp, vp, vp, i32, i64, vp, vp, i32, f32, i32, Axis, Axis, Axis, vp, vp, vp, i32, vp]
    L.hicom_gelu_split_fwd.argtypes = [vp, vp, vp, i64, vp]
    L.hicom_gelu_bwd_fwd.argtypes = [vp, vp, i64, i32, vp, i32, vp]
    L.hicom_act_rows_fwd.argtypes = [vp, i64, i64, i32, i32, vp, vp]
    L.hicom_act_bwd_rows_fwd.argtypes = [vp, vp, i64, i64, i32, i32, vp]
    L.hicom_colsum_fwd.argtypes = [vp, i64, i32, vp, i32, vp]
    L.hicom_clip_query_prep_fwd.argtypes = [vp, vp, i32, i32, i32, f32, vp, vp]
    L.hicom_inv_norm_fwd.argtypes = [vp, i32, i64, vp, vp]
    L.hicom_global_stream_clip_fwd.argtypes = [vp, i64, i32, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, i64,
                                               vp, vp, vp, i32, vp]
    L.hicom_ln_stream_fwd.argtypes = [vp, i32, i64, vp, vp, vp, vp, i32, f32, vp, i32, i32, i32, vp]
    L.hicom_merge_vproj_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.hicom_merge_vproj_fixed_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    L.hicom_fused_stream_nparts.argtypes = [i32]
    L.hicom_query_prep_state_bytes.argtypes = [i32]
    L.hicom_query_prep_state_bytes.restype = i64
    L.hicom_query_prep_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp]
    L.hicom_planes_gemm_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, i64, i64, i32, vp]
    L.hicom_row_ln_fwd.argtypes = [vp, i32, i64, vp, i64, vp, i64, vp, vp, i32, vp, i32, i64, vp, i32, f32, vp, i32, i64,
                                   i32, i32, vp]
    L.hicom_small_mha_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp]
    L.hicom_small_mha_scaled_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, f32, vp, vp]
    L.hicom_fold_query_split_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, i32, vp, i32, i32, vp]
    L.hicom_global_combine_strided_fwd.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp]
    ap = C.POINTER(CompressorArgs)
    L.hicom_compressor_workspace_bytes.argtypes = [ap]
    L.hicom_compressor_zero_prefix_bytes.argtypes = [ap]
    L.hicom_compressor_is_fused.argtypes = [ap]
    L.hicom_compressor_fwd2.argtypes = [ap, ap]
    L.hicom_cast16_fwd.argtypes = [vp, i32, vp, i32, i64, vp]
    L.hicom_compressor_takes_shard4.argtypes = [ap]
    L.hicom_compressor_handoff_failures.argtypes = [ap, C.POINTER(C.c_int32), vp]
    L.hicom_compressor_fwd.argtypes = [ap]
    L.hicom_global_combine_fwd.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.hicom_readout_gemm_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, i64, i64, i32, vp]
    L.hicom_scatter_rows_fwd.argtypes = [vp, i32, i32, i32, vp, i32, i64, i64, i64, i32, i32, vp]
    L.hicom_place_blocks_fwd.argtypes = [vp, i32, i32, i64, i32, vp, i64, i64, i32, vp]
    for name in EXPORTS[2:]:
        getattr(L, name).restype = C.c_int
    L.hicom_compressor_workspace_bytes.restype = C.c_int64
    L.hicom_compressor_zero_prefix_bytes.restype = C.c_int64
    _LIB = L
    return L


def _check(status: int, what: str):
    if status != 0:
        msg = lib().hicom_last_error().decode(errors="replace")
        raise HicomNativeError(f"{what} failed ({status}): {msg}")


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise HicomNativeError("hicom_amd kernels need device tensors (HIP); got a CPU tensor")
    if not t.is_contiguous():
        raise HicomNativeError("hicom_amd kernels need dense row-major tensors")
    return C.c_void_p(t.data_ptr())


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype == torch.float32:
        return DT_F32
    if t.dtype == torch.float16:
        return DT_F16
    raise HicomNativeError(f"unsupported dtype {t.dtype} (bf16 or f32 only)")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------------------------------
# thin typed wrappers (one per entry point)
# ------------------------------------------------------------------------------------------
def local_attn(key, value, axes, query, query_stride, scale, bias, l2norm, ctx, ctx_f16=None):
    D = value.shape[-1]
    _check(lib().hicom_local_attn_fwd(_ptr(key), _dt(key), _ptr(value), _dt(value), D, axes[0], axes[1], axes[2], _ptr(query), _dt(query),
                                      query_stride, scale, bias, l2norm, _ptr(ctx), _ptr(ctx_f16), _stream()), "hicom_local_attn_fwd")


def local_attn_adapt(key_x, key_y, k_norm, k_alpha, value_x, value_y, v_norm, v_alpha, axes, query, query_stride, scale, bias, ctx, eps=1e-6):
    """Window attention with the adaptor blends fused into the row loads; key_y / value_y fp16 [N, D] or None (see include/hicom_hip.h)."""
    D = value_x.shape[-1]
    alpha = k_alpha if k_alpha is not None else v_alpha
    _check(lib().hicom_local_attn_adapt_fwd(_ptr(key_x), _ptr(key_y), _ptr(k_norm.weight.detach()) if key_y is not None else None,
                                            _ptr(k_norm.bias.detach()) if key_y is not None else None, _ptr(k_alpha) if key_y is not None else None,
                                            _ptr(value_x), _ptr(value_y), _ptr(v_norm.weight.detach()) if value_y is not None else None,
                                            _ptr(v_norm.bias.detach()) if value_y is not None else None, _ptr(v_alpha) if value_y is not None else None,
                                            _dt(alpha), eps, D, axes[0], axes[1], axes[2], _ptr(query), _dt(query), query_stride,
                                            scale, bias, _ptr(ctx), _stream()), "hicom_local_attn_adapt_fwd")


def local_attn_adapt_bwd(key_x, key_y, k_norm, k_alpha, value_x, value_y, v_norm, v_alpha, axes, query, query_stride, scale, bias, dctx,
                         ds, pw, sxk, syk, sxv, syv, eps=1e-6):
    """Backward of the blend-fused window attention (see include/hicom_hip.h); key_y / value_y fp16 [N, D] or None."""
    D = value_x.shape[-1]
    alpha = k_alpha if k_alpha is not None else v_alpha
    _check(lib().hicom_local_attn_adapt_bwd(_ptr(key_x), _ptr(key_y), _ptr(k_norm.weight.detach()) if key_y is not None else None,
                                            _ptr(k_norm.bias.detach()) if key_y is not None else None, _ptr(k_alpha) if key_y is not None else None,
                                            _ptr(value_x), _ptr(value_y), _ptr(v_norm.weight.detach()) if value_y is not None else None,
                                            _ptr(v_norm.bias.detach()) if value_y is not None else None, _ptr(v_alpha) if value_y is not None else None,
                                            _dt(alpha), eps, D, axes[0], axes[1], axes[2], _ptr(query), _dt(query), query_stride, scale, bias,
                                            _ptr(dctx), _ptr(ds), _ptr(pw), _ptr(sxk), _ptr(syk), _ptr(sxv), _ptr(syv), _stream()),
           "hicom_local_attn_adapt_bwd")


_COL_PARTS = 1024        # workgroups (= partial rows) of the streaming kernels that leave column sums beside their output


def _col_parts(rows, D, dev):
    n = max(1, min(_COL_PARTS, (rows + 3) // 4))
    return torch.empty((n, D), dtype=torch.float32, device=dev)


def adapt_dy(y, gamma, vec, vec_stride, coef, alpha, axes, dy, r1=None, eps=1e-6, colsum=False):
    """colsum=True: also returns the column sums of dy (f32 [D], the bias gradient) from partials the same launch leaves."""
    D = y.shape[-1]
    parts = _col_parts(dy.shape[0], D, dy.device) if colsum else None
    _check(lib().hicom_adapt_dy_fwd(_ptr(y), _ptr(gamma), _ptr(vec), _dt(vec), vec_stride, _ptr(coef), _ptr(alpha), _dt(alpha), eps, D,
                                    axes[0], axes[1], axes[2], _ptr(dy), _ptr(r1), _ptr(parts), parts.shape[0] if colsum else 0, _stream()),
           "hicom_adapt_dy_fwd")
    if not colsum:
        return None
    out = torch.empty((D,), dtype=torch.float32, device=dy.device)
    partials_sum(parts, out)
    return out


def gelu_split(h16, a16, abf):
    _check(lib().hicom_gelu_split_fwd(_ptr(h16), _ptr(a16), _ptr(abf), h16.numel(), _stream()), "hicom_gelu_split_fwd")


def gelu_bwd_(da_bf16, h16, colsum=False):
    """da *= GELU'(h) in place; colsum=True (rows of 1152 / 768): also returns the column sums of the result (f32 [D])."""
    D = da_bf16.shape[-1]
    parts = _col_parts(da_bf16.shape[0], D, da_bf16.device) if colsum else None
    _check(lib().hicom_gelu_bwd_fwd(_ptr(da_bf16), _ptr(h16), da_bf16.numel(), D, _ptr(parts), parts.shape[0] if colsum else 0, _stream()),
           "hicom_gelu_bwd_fwd")
    if not colsum:
        return None
    out = torch.empty((D,), dtype=torch.float32, device=da_bf16.device)
    partials_sum(parts, out)
    return out


def act_rows(h16, cols, act, out=None):
    """bf16 [rows, cols] = act(h16[:, :cols]) for a pitched fp16 matrix (act: ACT_GELU | ACT_GELU_TANH)."""
    rows = h16.shape[0]
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.bfloat16, device=h16.device)
    _check(lib().hicom_act_rows_fwd(_ptr(h16), h16.shape[1], rows, cols, act, _ptr(out), _stream()), "hicom_act_rows_fwd")
    return out


def act_bwd_rows_(da_bf16, h16, act):
    """da_bf16 [rows, cols] *= act'(h16[:, :cols]) in place."""
    rows, cols = da_bf16.shape
    _check(lib().hicom_act_bwd_rows_fwd(_ptr(da_bf16), _ptr(h16), h16.shape[1], rows, cols, act, _stream()), "hicom_act_bwd_rows_fwd")
    return da_bf16


def colsum(x_bf16, nparts=128):
    """Column sums of a bf16 [N, D] matrix -> f32 [D] (partials + hicom_partials_sum_fwd: deterministic)."""
    N, D = x_bf16.shape
    nparts = max(1, min(nparts, (N + 3) // 4))
    parts = torch.empty((nparts, D), dtype=torch.float32, device=x_bf16.device)
    _check(lib().hicom_colsum_fwd(_ptr(x_bf16), N, D, _ptr(parts), nparts, _stream()), "hicom_colsum_fwd")
    out = torch.empty((D,), dtype=torch.float32, device=x_bf16.device)
    partials_sum(parts, out)
    return out


def local_attn_bwd(key, value, axes, query, query_stride, scale, bias, dctx, dq, dkey=None, l2norm_key=False, dls=None, dvalue=None,
                   value_is_key=False):
    """l2norm_key / dls: clip-scale on the local stage -- key rows L2-normalised, dls f32 [Nw] = per-window share of d logit_scale.
    dvalue bf16 [T,H,W,D]: gradient w.r.t. the value stream (value_is_key: the key-side gradient of the same rows is added in)."""
    D = value.shape[-1]
    assert key.dtype == torch.bfloat16 and value.dtype == torch.bfloat16 and dctx.dtype == torch.float32 and dq.dtype == torch.float32
    _check(lib().hicom_local_attn_bwd(_ptr(key), _ptr(value), D, axes[0], axes[1], axes[2], _ptr(query), _dt(query), query_stride,
                                      scale, bias, _ptr(dctx), _ptr(dq), _ptr(dkey), int(bool(l2norm_key)), _ptr(dls), _ptr(dvalue),
                                      int(bool(value_is_key)), _stream()), "hicom_local_attn_bwd")


def global_dx(S, dS, ml, qt, dctx, N, dx, accumulate):
    """d frames_feature of the global stage, direct recipe: dx[n] (+)= sum_r dS[r, n] qt[r] + softmax(S)[r, n] dctx[r] (hicom_global_dx_fwd)."""
    rows, E = qt.shape
    _check(lib().hicom_global_dx_fwd(_ptr(S), _ptr(dS), S.shape[1], _ptr(ml), _ptr(qt), _ptr(dctx), rows, N, E, _ptr(dx), int(bool(accumulate)),
                                     _stream()), "hicom_global_dx_fwd")


def trilinear_pool(x, out):
    T, H, W, D = x.shape
    To, Ho, Wo, _ = out.shape
    _check(lib().hicom_trilinear_pool_fwd(_ptr(x), T, H, W, D, To, Ho, Wo, _ptr(out), _stream()),
           "hicom_trilinear_pool_fwd")


def linear(x, w, b, y, res=None, res_bcast=False, act=ACT_NONE, head_rows=0, head_dim=0, M=None):
    res_flags = (1 if res_bcast else 0) | (2 if (res is not None and res.dtype == torch.bfloat16) else 0)
    N, K = w.shape
    M = y.shape[0] if M is None else M
    _check(lib().hicom_linear_fwd(_ptr(x), _dt(x), _ptr(w), _dt(w), _ptr(b), _dt(b) if b is not None else 0,
                                  _ptr(res), res_flags, M, N, K, head_rows, head_dim, act, _ptr(y), _stream()),
           "hicom_linear_fwd")


def fold_query(qp, w_k, nh, scale, qt):
    nq, E = qp.shape
    _check(lib().hicom_fold_query_fwd(_ptr(qp), _ptr(w_k), nq, nh, E, scale, _ptr(qt), _stream()), "hicom_fold_query_fwd")


def split_bf16(x, rows_pad, hi, lo):
    rows, E = x.shape
    _check(lib().hicom_split_bf16_fwd(_ptr(x), rows, rows_pad, E, _ptr(hi), _ptr(lo), _stream()), "hicom_split_bf16_fwd")


def global_stream_nparts(N, rows_pad) -> int:
    n = lib().hicom_global_stream_nparts(N, rows_pad)
    if n <= 0:
        raise HicomNativeError("hicom_global_stream_nparts: bad arguments")
    return n


def global_stream(x, N, qhi, qlo, pos_a, H, W, t0i, y0i, x0i, scores, part_m, part_l, part_acc, rows=None):
    E = x.shape[-1]
    rows_pad = qhi.shape[0]
    rows = rows_pad if rows is None else rows
    nparts = part_m.shape[0]
    _check(lib().hicom_global_stream_fwd(_ptr(x), N, E, _ptr(qhi), _ptr(qlo), rows, rows_pad, _ptr(pos_a),
                                         pos_a.shape[1] if pos_a is not None else 0, H, W, t0i, y0i, x0i,
                                         _ptr(scores), scores.shape[1], _ptr(part_m), _ptr(part_l), _ptr(part_acc),
                                         nparts, _stream()), "hicom_global_stream_fwd")


def global_stream_has_marg(N, E, rows_pad, H, W, nparts) -> bool:
    """True when the stream kernel can accumulate the positional marginals itself for this shape (global_stream_marg)."""
    return lib().hicom_global_stream_has_marg(N, E, rows_pad, H, W, nparts) == 1


def global_stream_marg_width(H, W) -> int:
    return lib().hicom_global_stream_marg_width(H, W)


def global_stream_marg(x, N, qhi, qlo, pos_a, H, W, t0i, y0i, x0i, scores, part_m, part_l, part_acc, part_marg, rows=None):
    """hicom_global_stream_marg_fwd: positional marginals in the kernel; scores may be None (no logit tensor)."""
    E = x.shape[-1]
    rows_pad = qhi.shape[0]
    rows = rows_pad if rows is None else rows
    _check(lib().hicom_global_stream_marg_fwd(_ptr(x), N, E, _ptr(qhi), _ptr(qlo), rows, rows_pad, _ptr(pos_a), pos_a.shape[1],
                                              H, W, t0i, y0i, x0i, _ptr(scores), scores.shape[1] if scores is not None else 0,
                                              _ptr(part_m), _ptr(part_l), _ptr(part_acc), _ptr(part_marg), part_m.shape[0],
                                              _stream()), "hicom_global_stream_marg_fwd")


def global_merge_marg(part_m, part_l, part_acc, part_marg, rows, N, H, W, pe, t0i, y0i, x0i, scratch, out_ml, out_acc,
                      normalize=False):
    nparts, rows_pad = part_m.shape
    _check(lib().hicom_global_merge_marg_fwd(_ptr(part_m), _ptr(part_l), _ptr(part_acc), _ptr(part_marg), nparts, rows, rows_pad,
                                             part_acc.shape[-1], N, H, W, _ptr(pe), t0i, y0i, x0i, _ptr(scratch), _ptr(out_ml),
                                             _ptr(out_acc), int(normalize), _stream()), "hicom_global_merge_marg_fwd")


def global_stream_bwd(x, N, dhi, dlo, pos_b, H, W, t0i, y0i, x0i, s_in, ml, delta, ds_out, part_acc, rows, part_marg=None):
    """part_marg (shapes with global_stream_has_marg): the positional marginals of dS per token chunk; ds_out may then be None."""
    E = x.shape[-1]
    _check(lib().hicom_global_stream_bwd(_ptr(x), N, E, _ptr(dhi), _ptr(dlo), rows, dhi.shape[0], _ptr(pos_b),
                                         pos_b.shape[1] if pos_b is not None else 0, H, W, t0i, y0i, x0i, _ptr(s_in),
                                         s_in.shape[1], _ptr(ml), _ptr(delta), _ptr(ds_out), _ptr(part_acc), _ptr(part_marg),
                                         part_acc.shape[0], _stream()), "hicom_global_stream_bwd")


_MARG_IDX = {}


def marg_frame_index(N, H, W, nparts, device):
    """Frame of column c of chunk i's frame block in a part_marg tensor (hicom_global_stream_marg_fwd / _bwd): [nparts, 8] int64,
    entries past the clip's last frame clamped to it (their marginals are zero).  Cached per shape: the upload must not happen inside
    a graph capture."""
    key = (N, H, W, nparts, str(device))
    hit = _MARG_IDX.get(key)
    if hit is not None:
        return hit
    if len(_MARG_IDX) >= 16:
        _MARG_IDX.clear()
    ntiles = (N + 15) // 16
    T = N // (H * W)
    first = torch.tensor([((ntiles * i) // nparts * 16) // (H * W) for i in range(nparts)], dtype=torch.int64)
    idx = (first[:, None] + torch.arange(8)[None, :]).clamp_(max=T - 1)
    hit = _MARG_IDX[key] = idx.to(device)
    return hit


def global_merge(part_m, part_l, part_acc, rows, scores, N, H, W, pe, t0i, y0i, x0i, scratch, out_ml, out_acc,
                 normalize=False):
    nparts, rows_pad = part_m.shape
    E = part_acc.shape[-1]
    _check(lib().hicom_global_merge_fwd(_ptr(part_m), _ptr(part_l), _ptr(part_acc), nparts, rows, rows_pad, E,
                                        _ptr(scores), scores.shape[1] if scores is not None else 0, N, H, W, _ptr(pe), t0i, y0i, x0i,
                                        _ptr(scratch), _ptr(out_ml), _ptr(out_acc), int(normalize), _stream()),
           "hicom_global_merge_fwd")


def global_combine(ml, acc, ctx):
    nsets, rows, E = acc.shape
    _check(lib().hicom_global_combine_fwd(_ptr(ml), _ptr(acc), nsets, rows, E, _ptr(ctx), _stream()),
           "hicom_global_combine_fwd")


def readout_gemm(x, w, b, y, act=ACT_NONE, row0=0, nl_group=0, M=None):
    N, K = w.shape
    M = x.shape[0] if M is None else M
    _check(lib().hicom_readout_gemm_fwd(_ptr(x), _ptr(w), _ptr(b), _dt(b) if b is not None else 0, M, N, K, act,
                                        _ptr(y), _dt(y), y.shape[-1], row0, nl_group, _stream()), "hicom_readout_gemm_fwd")


def scatter_rows(src, dst, row0, count, row_step=1, nl_group=0, stream=None):
    src2 = src.reshape(-1, src.shape[-1])
    _check(lib().hicom_scatter_rows_fwd(_ptr(src2), _dt(src2), src2.shape[0], src2.shape[1], _ptr(dst), _dt(dst),
                                        dst.shape[-1], row0, row_step, nl_group, count, _stream() if stream is None else stream),
           "hicom_scatter_rows_fwd")


def place_blocks(src_ptr: int, block_rows, nblocks, block_stride_bytes, row_bytes, dst, row0, nl_group=0, stream=None):
    _check(lib().hicom_place_blocks_fwd(src_ptr, block_rows, nblocks, block_stride_bytes, row_bytes, _ptr(dst),
                                        dst.shape[-1] * dst.element_size(), row0, nl_group, _stream() if stream is None else stream),
           "hicom_place_blocks_fwd")


def fold_query_split(qp, w_k, kpe, nh, scale, qhi, qlo, pos_a, fill_row=None, fill_row0=0, fill_rows=0):
    nq, E = qp.shape
    P = kpe.shape[1] if kpe is not None else 0
    _check(lib().hicom_fold_query_split_fwd(_ptr(qp), _ptr(w_k), _ptr(kpe), nq, nh, E, P, scale, _ptr(qhi), _ptr(qlo),
                                            _ptr(pos_a), pos_a.shape[1] if pos_a is not None else 0, _ptr(fill_row),
                                            fill_row0, fill_rows, _stream()), "hicom_fold_query_split_fwd")


def global_combine_strided(ml, acc, set_stride, nsets, rows, E, ctx):
    _check(lib().hicom_global_combine_strided_fwd(_ptr(ml), _ptr(acc), set_stride, nsets, rows, E, _ptr(ctx), _stream()),
           "hicom_global_combine_strided_fwd")


def compressor_workspace(args: CompressorArgs):
    L = lib()
    total, prefix = L.hicom_compressor_workspace_bytes(C.byref(args)), L.hicom_compressor_zero_prefix_bytes(C.byref(args))
    if total <= 0 or prefix < 0:
        raise HicomNativeError("hicom_compressor_workspace_bytes: bad arguments")
    return total, prefix


def compressor_fwd(args: CompressorArgs):
    _check(lib().hicom_compressor_fwd(C.byref(args)), "hicom_compressor_fwd")


def compressor_fwd2(first: CompressorArgs, second: CompressorArgs):
    """hicom_compressor_fwd(first) then hicom_compressor_fwd(second) in ONE host call (the two phases of a frame-sharded step)."""
    _check(lib().hicom_compressor_fwd2(C.byref(first), C.byref(second)), "hicom_compressor_fwd2")


def compressor_takes_shard4(args: CompressorArgs) -> bool:
    return bool(lib().hicom_compressor_takes_shard4(C.byref(args)))


def compressor_handoff_failures(args: CompressorArgs, stream=None):
    """(query prep, GEMV chain): in-launch hand-offs of this workspace whose bounded spin expired (their rows were poisoned with NaN).
    Synchronises the stream: tests / bench / debug only."""
    out = (C.c_int32 * 2)()
    _check(lib().hicom_compressor_handoff_failures(C.byref(args), out, _stream() if stream is None else stream), "hicom_compressor_handoff_failures")
    return int(out[0]), int(out[1])


def rccl_fns():
    """Addresses of (ncclAllGather, ncclGroupStart, ncclGroupEnd) in the RCCL that torch has loaded (torch/lib/librccl.so): handed to
    hicom_compressor_args.ag_fn / ag_group_* so that the FINISH call of a frame-sharded step enqueues its collective itself.
    libhicom_hip.so does not link RCCL."""
    global _RCCL
    if _RCCL is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        L = C.CDLL(path if os.path.exists(path) else "librccl.so")        # (already mapped by torch.distributed: the same instance)
        _RCCL = tuple(C.cast(getattr(L, n), C.c_void_p).value for n in ("ncclAllGather", "ncclGroupStart", "ncclGroupEnd"))
    return _RCCL


def rccl_allgather_fn() -> int:
    return rccl_fns()[0]


_RCCL = None


def compressor_ctx16_offset(args) -> int:
    """Workspace offset of the fp16 window-context plane a call with these arguments leaves behind, or -1 (hicom_compressor_ctx16_offset)."""
    off = int(lib().hicom_compressor_ctx16_offset(C.byref(args)))
    return off if off >= 0 else -1


def compressor_is_fused(args: CompressorArgs) -> bool:
    """True when these arguments take the release-recipe path: one streaming kernel for both levels."""
    return bool(lib().hicom_compressor_is_fused(C.byref(args)))


def linear_to_rows(x, w, b, dst, row0, n_rows, act=ACT_NONE):
    N, K = w.shape
    _check(lib().hicom_linear_to_rows_fwd(_ptr(x), _dt(x), _ptr(w), _dt(w), _ptr(b), _dt(b) if b is not None else 0,
                                          x.shape[0], N, K, act, _ptr(dst), _dt(dst), dst.shape[-1], row0, n_rows,
                                          _stream()), "hicom_linear_to_rows_fwd")


def fused_stream_nparts(n_windows: int) -> int:
    n = lib().hicom_fused_stream_nparts(n_windows)
    if n <= 0:
        raise HicomNativeError("hicom_fused_stream_nparts: bad arguments")
    return n


def fused_stream(ff, fe, kt, ks, qhi, qlo, rows, l_scale, l_bias, pos_a, pe_hi, pe_lo, t0i, y0i, x0i, part_m, part_l,
                 part_acc, ctx_local, ctx_hi=None, ctx_lo=None, ctx_f16=None, local_logits=None, zero=None, part_ctx_f16=None, part_marg=None):
    """pos_a f32 [16, P] + pe_hi / pe_lo bf16 [P, E] (all three or none): the kernel folds the value-side
    pos-emb into part_acc.  local_logits f32 [T*H*W] (fe . local query per token) replaces the frames_embed stream.
    part_marg fp16 [nparts, rows, S] (with pe_hi = pe_lo = None): the normalised positional marginals leave the kernel instead."""
    T, H, W, E = ff.shape
    _check(lib().hicom_fused_stream_fwd(_ptr(ff), _ptr(fe), _ptr(local_logits), T, H, W, E, kt, ks, _ptr(qhi), _ptr(qlo), rows, l_scale,
                                        l_bias, _ptr(pos_a), pos_a.shape[1] if pos_a is not None else 0,
                                        _ptr(pe_hi), _ptr(pe_lo), t0i, y0i, x0i,
                                        _ptr(part_m), _ptr(part_l), _ptr(part_acc), part_m.shape[0], _ptr(ctx_local),
                                        _ptr(ctx_hi), _ptr(ctx_lo), _ptr(ctx_f16), _ptr(zero),
                                        zero.numel() * zero.element_size() if zero is not None else 0, _ptr(part_ctx_f16),
                                        _ptr(part_marg), part_marg.shape[-1] if part_marg is not None else 0, _stream()),
           "hicom_fused_stream_fwd")


def planes_gemm(a_hi, a_lo, w, b, act=ACT_NONE, out_hi=None, out_lo=None, y=None, row0=0, nl_group=0):
    N, K = w.shape
    M = a_hi.shape[0]
    _check(lib().hicom_planes_gemm_fwd(_ptr(a_hi), _ptr(a_lo), _ptr(w), _ptr(b), _dt(b) if b is not None else 0, M, N, K,
                                       act, _ptr(out_hi), _ptr(out_lo), _ptr(y), _dt(y) if y is not None else 0,
                                       y.shape[-1] if y is not None else 0, row0, nl_group, _stream()),
           "hicom_planes_gemm_fwd")


def cast16(src, dtype):
    """fp16 -> bf16 (round to nearest even) or bf16 -> fp16 (saturating) copy of a contiguous tensor (hicom_cast16_fwd)."""
    src = src.contiguous()
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    if src.numel():
        _check(lib().hicom_cast16_fwd(_ptr(src), _dt(src), _ptr(dst), _dt(dst), src.numel(), _stream()), "hicom_cast16_fwd")
    return dst


def to_f16(src, dst=None):
    """Saturating cast of a bf16 / f32 tensor to fp16 (weights of the fp16 readout path; activations in tests)."""
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.float16, device=src.device)
    _check(lib().hicom_to_f16_fwd(_ptr(src), _dt(src), _ptr(dst), src.numel(), _stream()), "hicom_to_f16_fwd")
    return dst


# ---- weight-derived device caches ---------------------------------------------------------------------------------------
# fp16 copies of nn.Linear weights, kpe = W_k . PE^T, the plans of engine.py ... are keyed by every source weight's
# (storage pointer, in-place version counter).  That sees `p.data = new`, optimizer steps through torch ops, load_state_dict
# and .to(); it does NOT see writes that bypass the version counter: `p.data.copy_(...)` and writes through an alias of the
# storage (DeepSpeed's bf16 optimizer updates its flat buffer exactly like that).  Hence a global EPOCH in every stamp:
#   * every training-mode forward (HIComProjector.forward / siglip_head_embed with autograd on and trainable parameters) bumps
#     it (`note_training_forward`), so a training forward always reads the live weights;
#   * a training forward also leaves a DIRTY mark: the tables it (or its backward) built carry the current epoch but pre-date
#     the optimizer step that follows.  The next INFERENCE entry point (`begin_inference`: HIComProjector.forward under
#     no_grad, forward_deferred, sharded_forward, the stage modules' forwards, siglip_head_embed / _scores) consumes the mark by
#     bumping the epoch once more, so Trainer.evaluate / inference after training rebuilds from the post-step weights; the
#     backward passes never consume it (they run between the forward and the optimizer step);
#   * `hicom_amd.invalidate_weight_caches()` bumps it by hand (after modifying weights behind torch's back at inference).
_WEIGHTS_EPOCH = [0]
_TRAIN_DIRTY = [False]


_DATA_ALIASES = []          # [alias tensor handed out by TrackedParameter.data, version counter last seen]


def _alias_refs_floor() -> int:
    """sys.getrefcount of a tensor that only `_DATA_ALIASES` holds, as `_scan_data_aliases` sees it (list entry + loop variable + argument)."""
    ent = [torch.empty(0), 0]
    al = ent[0]
    return sys.getrefcount(al)


_ALIAS_FLOOR = None


def _scan_data_aliases() -> None:
    """A `.data` alias of a projector parameter has a version counter of its own: an in-place op on it (or on a view of it) is the
    write that `p._version` does not see.  Aliases written since the last scan bump the weights epoch; aliases nobody else holds any
    more are dropped (a written one after it was counted), the others stay watched."""
    global _ALIAS_FLOOR
    if not _DATA_ALIASES:
        return
    if _ALIAS_FLOOR is None:
        _ALIAS_FLOOR = _alias_refs_floor()
    keep, moved = [], False
    for ent in _DATA_ALIASES:
        al = ent[0]
        v = al._version
        if v != ent[1]:
            moved = True
            ent[1] = v
        if sys.getrefcount(al) > _ALIAS_FLOOR:
            keep.append(ent)
    _DATA_ALIASES[:] = keep
    if moved:
        _WEIGHTS_EPOCH[0] += 1


def weights_epoch() -> int:
    _scan_data_aliases()
    return _WEIGHTS_EPOCH[0]


def invalidate_weight_caches() -> None:
    """Every weight-derived device cache (fp16 weight copies, positional products, executor plans) is rebuilt on next use."""
    _WEIGHTS_EPOCH[0] += 1


def note_training_forward() -> None:
    """Called by every forward that builds an autograd graph over trainable weights."""
    _WEIGHTS_EPOCH[0] += 1
    _TRAIN_DIRTY[0] = True


def begin_inference() -> None:
    """Called at every inference entry point: the first one after a training forward rebuilds the weight-derived tables."""
    if _TRAIN_DIRTY[0]:
        _TRAIN_DIRTY[0] = False
        _WEIGHTS_EPOCH[0] += 1


def weight_stamp(*weights):
    return (weights_epoch(),) + tuple(v for w in weights for v in (w.data_ptr(), w._version))


class TrackedParameter(torch.nn.Parameter):
    """nn.Parameter whose `.data` attribute reports WRITES through it (round 5 closed the stale-weights footgun of eval-mode
    `p.data.copy_`; round 6 stopped charging reads for it).

    `p.data` hands out an alias of the storage with a FRESH version counter: `p.data.copy_(w)`, `p.data.mul_(s)` change the weights
    without any trace on `p._version`, and at inference nothing else (no training forward) bumps the weights epoch, so the
    weight-derived tables (fp16 readout copies, kpe, folded products, executor plans) would be served stale.  Round 5 bumped the
    process-wide epoch on every ACCESS of `.data` -- and a read (`p.data.norm()` in a logger, an EMA, DeepSpeed's bookkeeping, deepcopy,
    torch.save) then cost every projector of the process a rebuild of its tables on the next forward (174 against 76 us at the benchmark
    shape).  Now the getter only REGISTERS the alias it hands out; the alias's own version counter tells whether it was written
    (`_scan_data_aliases`, run wherever the epoch is read), and only then the epoch moves.  `p.data = w` (the setter) still bumps at
    once.  Forward / backward, optimizers, state_dict(), load_state_dict() and this package's own code never touch `.data`.
    Not covered: writes through an alias of the STORAGE taken by other means (DeepSpeed's flat bf16 buffer, `p.detach()` is covered by
    torch itself: it shares p's version counter) -- the training-forward epoch and its dirty mark (above) handle those;
    `invalidate_weight_caches()` stays as the explicit form."""

    @property
    def data(self):
        al = torch.Tensor.data.__get__(self)
        _DATA_ALIASES.append([al, al._version])
        if len(_DATA_ALIASES) > 4096:                       # (a loop that hoards aliases without ever running a forward)
            _scan_data_aliases()
        return al

    @data.setter
    def data(self, value):
        _WEIGHTS_EPOCH[0] += 1
        torch.Tensor.data.__set__(self, value)


def track_parameters(module) -> None:
    """Every nn.Parameter of `module` becomes a TrackedParameter IN PLACE (same object: optimizers and state dicts keep their
    references; `copy.deepcopy` keeps the class, pickling falls back to nn.Parameter and is re-tracked on the next plan build)."""
    for p in module.parameters():
        if type(p) is torch.nn.Parameter:
            p.__class__ = TrackedParameter


_F16_MAX = 65504.0
_RANGE_CHECKED = set()


def f16_weight_copy(w, ld=None, out=None):
    """fp16 copy of a bf16 weight for the fp16-operand GEMMs.  bf16 -> fp16 is exact for 2^-14 <= |w| < 65504; smaller
    magnitudes land on fp16 subnormals (absolute error <= 2^-25 = 3e-8) and larger ones would saturate, so the range is
    checked ONCE per weight storage (one device reduction + host read when the copy is first built; rebuilds of the same
    storage -- training -- are not re-checked)."""
    w = w.detach()
    key = (w.data_ptr(), tuple(w.shape))
    if key not in _RANGE_CHECKED:
        amax = float(w.abs().max())
        if not amax < _F16_MAX:
            raise NotImplementedError(f"hicom_amd: a weight of magnitude {amax:.3g} does not fit fp16 (the fp16-operand GEMM path "
                                      "saturates at 65504); no bf16-operand fallback for this layer")
        if len(_RANGE_CHECKED) > 4096:
            _RANGE_CHECKED.clear()
        _RANGE_CHECKED.add(key)
    # out: a previous copy of the same shape is refreshed IN PLACE (its address is baked into executor plans: a training step
    # then re-runs the cast, not the plan build)
    if out is not None and (out.dtype != torch.float16 or tuple(out.shape) != ((w.shape[0], ld) if ld is not None else tuple(w.shape))):
        out = None
    return to_f16(w, out) if ld is None else to_f16_padded(w, ld, out)


def to_f16_padded(src, ld, dst=None):
    """fp16 copy of a [rows, cols] bf16 / f32 matrix with its rows zero-padded to `ld` columns (K padding of a GEMM operand)."""
    rows, cols = src.shape
    if dst is None:
        dst = torch.empty((rows, ld), dtype=torch.float16, device=src.device)
    _check(lib().hicom_to_f16_padded_fwd(_ptr(src), _dt(src), rows, cols, _ptr(dst), ld, _stream()), "hicom_to_f16_padded_fwd")
    return dst


def dense16_gemm(a, w, b, N=None, K=None, act=ACT_NONE, out_f16=None, n_store=None, y=None, res=None, ssq=None, row_tab=None,
                 row_dot=None, pre_f16=None):
    """C = epi(A . W^T + b) on matrix cores; a [M, lda], w [N, ldw] both fp16 or both bf16 (see include/hicom_hip.h).
    row_dot = (vec [N] bf16 | f32, parts f32 [ceil(N/64), M]): per-slice partials of vec . (value + res) per row.
    pre_f16: fp16 [M, >= N] that receives acc + b before the activation."""
    if a.dtype != w.dtype or a.dtype not in (torch.float16, torch.bfloat16):
        raise HicomNativeError("dense16_gemm: operands are both fp16 or both bf16")
    M = a.shape[0]
    N = w.shape[0] if N is None else N
    K = w.shape[1] if K is None else K
    _check(lib().hicom_dense16_gemm_fwd(_ptr(a), a.shape[1], _ptr(w), w.shape[1], _dt(a), _ptr(b), _dt(b) if b is not None else 0,
                                        M, N, K, act, _ptr(out_f16), out_f16.shape[1] if out_f16 is not None else 0,
                                        (out_f16.shape[1] if n_store is None else n_store) if out_f16 is not None else 0,
                                        _ptr(pre_f16), pre_f16.shape[1] if pre_f16 is not None else 0,
                                        _ptr(y), _dt(y) if y is not None else 0, y.shape[1] if y is not None else 0,
                                        _ptr(res), res.shape[1] if res is not None else 0, _ptr(ssq),
                                        _ptr(row_tab[0]) if row_tab else None, row_tab[0].shape[1] if row_tab else 0,
                                        *(row_tab[1:] if row_tab else (0, 0, 0, 0, 0)),
                                        _ptr(row_dot[0]) if row_dot else None, _dt(row_dot[0]) if row_dot else 0,
                                        _ptr(row_dot[1]) if row_dot else None, _stream()),
           "hicom_dense16_gemm_fwd")


def dense16_gemm_pair(a_k, w_k, b_k, out_k, a_v, w_v, b_v, out_v, act=ACT_NONE, pre_k=None, pre_v=None):
    """Two GEMMs of ONE shape in one launch (the same layer of the k and of the v adaptor MLP): out_x = act(a_x . w_x^T + b_x), fp16 outputs."""
    if a_k.dtype != w_k.dtype or a_v.dtype != a_k.dtype or w_v.dtype != a_k.dtype or a_k.shape != a_v.shape or w_k.shape != w_v.shape:
        raise HicomNativeError("dense16_gemm_pair: two problems of one shape and one operand dtype")
    M, (N, K) = a_k.shape[0], w_k.shape
    _check(lib().hicom_dense16_gemm_pair_fwd(_ptr(a_k), _ptr(w_k), _ptr(b_k), _ptr(out_k), _ptr(pre_k), _ptr(a_v), _ptr(w_v), _ptr(b_v), _ptr(out_v),
                                             _ptr(pre_v), a_k.shape[1], w_k.shape[1], _dt(a_k), _dt(b_k) if b_k is not None else 0, M, N, K, act,
                                             out_k.shape[1], out_k.shape[1], pre_k.shape[1] if pre_k is not None else 0, _stream()),
           "hicom_dense16_gemm_pair_fwd")


def dense16_tn(a, b, M=None, N=None, out=None, splits=None):
    """a^T b over the ROWS of a [Kt, lda] and b [Kt, ldb] (both fp16 or both bf16) -> f32 [M, N]: the weight gradients dW = dY^T X of
    the token-stream layers on matrix cores (hicom_dense16_tn_fwd + hicom_partials_sum_fwd; see include/hicom_hip.h)."""
    if a.dtype != b.dtype or a.dtype not in (torch.float16, torch.bfloat16) or a.shape[0] != b.shape[0]:
        raise HicomNativeError("dense16_tn: operands are both fp16 or both bf16 with the same number of rows")
    Kt = a.shape[0]
    M = a.shape[1] if M is None else M
    N = b.shape[1] if N is None else N
    if splits is None:
        splits = lib().hicom_dense16_tn_splits(M, N, Kt)
    parts = torch.empty((splits, M, N), dtype=torch.float32, device=a.device)
    _check(lib().hicom_dense16_tn_fwd(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _dt(a), Kt, M, N, _ptr(parts), N, splits, _stream()),
           "hicom_dense16_tn_fwd")
    if splits == 1:
        return parts[0] if out is None else out.copy_(parts[0])
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    partials_sum(parts.view(splits, M * N), out.view(-1))
    return out


def l2norm_stream(x, out):
    M, E = x.shape
    _check(lib().hicom_l2norm_stream_fwd(_ptr(x), _ptr(out), M, E, _stream()), "hicom_l2norm_stream_fwd")


def partials_sum(parts, out):
    n, M = parts.shape
    _check(lib().hicom_partials_sum_fwd(_ptr(parts), n, M, _ptr(out), _stream()), "hicom_partials_sum_fwd")


def clip_query_prep(qp, b_k, nh, scale, c):
    nq, E = qp.shape
    _check(lib().hicom_clip_query_prep_fwd(_ptr(qp), _ptr(b_k), nq, nh, E, scale, _ptr(c), _stream()), "hicom_clip_query_prep_fwd")


def inv_norm(ssq, inv):
    parts, M = ssq.shape
    _check(lib().hicom_inv_norm_fwd(_ptr(ssq), parts, M, _ptr(inv), _stream()), "hicom_inv_norm_fwd")


def global_stream_clip(x, N, qhi, qlo, pos_a, H, W, t0i, y0i, x0i, inv, row_const, scores, part_m, part_l, part_acc, rows):
    E = x.shape[-1]
    _check(lib().hicom_global_stream_clip_fwd(_ptr(x), N, E, _ptr(qhi), _ptr(qlo), rows, qhi.shape[0], _ptr(pos_a),
                                              pos_a.shape[1] if pos_a is not None else 0, H, W, t0i, y0i, x0i, _ptr(inv), _ptr(row_const),
                                              _ptr(scores), scores.shape[1], _ptr(part_m), _ptr(part_l), _ptr(part_acc),
                                              part_m.shape[0], _stream()), "hicom_global_stream_clip_fwd")


def ln_stream(x, gamma, beta, out, src=None, alpha=None, eps=1e-6):
    M, E = out.shape
    _check(lib().hicom_ln_stream_fwd(_ptr(x), _dt(x), x.shape[1], _ptr(gamma), _ptr(beta), _ptr(src), _ptr(alpha),
                                     _dt(alpha) if alpha is not None else 0, eps, _ptr(out), _dt(out), M, E, _stream()),
           "hicom_ln_stream_fwd")


def _aux_gemv(aux) -> AuxGemv:
    ag = AuxGemv()
    if aux.get("x_fixed") is not None:                    # int64 [K] fixed-point accumulators (merge_vproj_fixed)
        ag.x_fixed = aux["x_fixed"].data_ptr()
    elif aux.get("xs") is not None:
        xs = aux["xs"].reshape(-1, aux["w"].shape[1])
        ag.xs, ag.x_parts, ag.x_stride = xs.data_ptr(), xs.shape[0], xs.shape[1]
    ag.xb = None if aux.get("xb") is None else aux["xb"].data_ptr()
    ag.w, ag.N, ag.K = aux["w"].data_ptr(), aux["w"].shape[0], aux["w"].shape[1]
    ag.b = None if aux.get("b") is None else aux["b"].data_ptr()
    ag.res = None if aux.get("res") is None else aux["res"].data_ptr()
    ag.act, ag.y = aux.get("act", ACT_NONE), (None if aux.get("y") is None else aux["y"].data_ptr())
    ag.w_dt = _dt(aux["w"])
    ag.b_dt = 0 if aux.get("b") is None else _dt(aux["b"])
    if aux.get("rows") is not None:
        dst, row0_, reps = aux["rows"]
        ag.rows_dst, ag.rows_dt, ag.rows_reps, ag.rows_ld, ag.rows_row0 = dst.data_ptr(), _dt(dst), reps, dst.shape[-1], row0_
    return ag


def _merge_role(merge):
    role = R16Role()
    role.kind = ROLE_MERGE_VPROJ
    pc = merge["part_ctx16"]
    role.part_m, role.part_l, role.part_acc, role.part_dt = merge["part_m"].data_ptr(), merge["part_l"].data_ptr(), pc.data_ptr(), DT_F16
    role.nparts, role.rows_pad, role.E, role.rows = pc.shape[0], pc.shape[1], pc.shape[2], merge["rows"]
    role.w_v, role.o_fix = _ptr(merge.get("w_v")), _ptr(merge.get("o_fix"))
    role.out_ml, role.out_ctx = _ptr(merge.get("out_ml")), _ptr(merge.get("out_ctx"))
    role.ctx_unnorm = int(bool(merge.get("unnorm", False)))
    if merge.get("part_marg") is not None:               # value-side pos-emb in the merge: fp16 [nparts, rows, S] marginals + fp16 [E, S] = W_v . pe^T
        role.part_marg, role.vpe_f16, role.marg_slots = merge["part_marg"].data_ptr(), merge["vpe_f16"].data_ptr(), merge["part_marg"].shape[-1]
    return role


def _chain_role(chain):
    a1, a2, state = chain
    role = R16Role()
    role.kind = ROLE_GEMV_CHAIN
    role.gemv, role.gemv2, role.chain_state = _aux_gemv(a1), _aux_gemv(a2), state.data_ptr()
    return role


def readout_tail_state(device):
    """Zeroed counter block of hicom_readout_tail_fwd, owned by ONE stream of launches."""
    return torch.zeros(int(lib().hicom_readout_tail_state_bytes()), dtype=torch.uint8, device=device)


def readout_tail(a16, w1_16, b1, hid16, w2_16, b2, y, merge, chain, state, row0=0, nl_group=0):
    """GELU(a . w1^T + b1) -> hid16 -> hid16 . w2^T + b2 -> y rows, with the merge role and the chain role, ONE launch
    (hicom_readout_tail_fwd).  Raises HicomNativeError(HICOM_EUNSUP) for shapes outside the fused form."""
    M, K1 = a16.shape
    N1, N2 = w1_16.shape[0], w2_16.shape[0]
    g1 = R16Gemm(_ptr(a16), _ptr(w1_16), _ptr(b1), _dt(b1) if b1 is not None else 0, M, N1, K1, ACT_GELU, _ptr(hid16), None, 0, 0, 0, 0)
    g2 = R16Gemm(_ptr(hid16), _ptr(w2_16), _ptr(b2), _dt(b2) if b2 is not None else 0, M, N2, N1, ACT_NONE, None, _ptr(y), _dt(y), y.shape[-1],
                 row0, nl_group)
    r1, r2 = _merge_role(merge), _chain_role(chain)
    _check(lib().hicom_readout_tail_fwd(C.byref(g1), C.byref(g2), C.byref(r1), C.byref(r2), _ptr(state), _stream()), "hicom_readout_tail_fwd")


def readout16_gemm(a16, w16, b, act=ACT_NONE, out_f16=None, y=None, row0=0, nl_group=0, aux=None, merge=None, chain=None):
    """aux: dict(xs f32 [parts, K] | x_fixed int64 [K], xb bf16 [K] | None, w bf16 | f32 [N, K], b bf16 | f32 [N] | None, res bf16 [N] | None,
    act, y f32 [N] | None, rows=(dst [*, ld], row0, reps) | None): one GEMV in the launch (HICOM_ROLE_GEMV).
    merge: dict(part_m, part_l f32 [nparts, rows_pad], part_ctx16 fp16 [nparts, rows_pad, E], rows, w_v bf16 [E, E], o_fix int64 [E] (zero),
    out_ml | None, out_ctx | None): the merge + v_proj items in the launch (HICOM_ROLE_MERGE_VPROJ).
    chain: (aux1, aux2, state): two dependent GEMVs in the launch (HICOM_ROLE_GEMV_CHAIN), aux1 from x_fixed; state = r16_chain_state()."""
    N, K = w16.shape
    M = a16.shape[0]
    args = (_ptr(a16), _ptr(w16), _ptr(b), _dt(b) if b is not None else 0, M, N, K, act,
            _ptr(out_f16), _ptr(y), _dt(y) if y is not None else 0, y.shape[-1] if y is not None else 0, row0, nl_group)
    if merge is None and chain is None:
        ag = _aux_gemv(aux) if aux is not None else None
        _check(lib().hicom_readout16_gemm_fwd(*args, C.byref(ag) if ag is not None else None, _stream()), "hicom_readout16_gemm_fwd")
        return
    role = _merge_role(merge) if merge is not None else _chain_role(chain)
    _check(lib().hicom_readout16_gemm_role_fwd(*args, C.byref(role), _stream()), "hicom_readout16_gemm_role_fwd")


def gemv_chain(a1, a2, state):
    """The two-layer GEMV chain as a launch of its own (hicom_gemv_chain_fwd): see readout16_gemm(chain=)."""
    role = R16Role()
    role.kind = ROLE_GEMV_CHAIN
    role.gemv, role.gemv2, role.chain_state = _aux_gemv(a1), _aux_gemv(a2), state.data_ptr()
    _check(lib().hicom_gemv_chain_fwd(C.byref(role), _stream()), "hicom_gemv_chain_fwd")


def merge_vproj_sets(sets, rows, E, w_v, o_fix, out_ml=None, out_ctx=None):
    """merge + v_proj over gathered shard states `sets` f32 [nsets, stride >= 2 rows + rows * E] (hicom_merge_vproj_sets_fwd)."""
    _check(lib().hicom_merge_vproj_sets_fwd(_ptr(sets), sets.shape[1], sets.shape[0], rows, E, _ptr(w_v), _ptr(o_fix), _ptr(out_ml), _ptr(out_ctx),
                                            _stream()), "hicom_merge_vproj_sets_fwd")


def r16_chain_state(n_mid, device):
    """Zeroed state block of the GEMV chain role (arrival counter + granules), owned by ONE sequence of launches of one shape."""
    return torch.zeros(int(lib().hicom_r16_chain_state_bytes(n_mid)), dtype=torch.uint8, device=device)


def query_prep_state(E, device):
    """Zeroed state block of hicom_query_prep_fwd (epoch word + granules), private to one stream of calls."""
    return torch.zeros(int(lib().hicom_query_prep_state_bytes(E)), dtype=torch.uint8, device=device)


def query_prep(guide, local_q, w_q, b_q, w_k, kpe, nh, scale, qhi, qlo, pos_a, state, g_w0=None, g_b0=None, b_o=None, r0=None):
    """One-launch q_proj + fold (+ positional table, local query rows, r0) of the direct recipe; see include/hicom_hip.h."""
    E = w_q.shape[0]
    _check(lib().hicom_query_prep_fwd(_ptr(guide), _ptr(local_q), _ptr(w_q), _ptr(b_q), _ptr(w_k), _ptr(kpe), nh, E,
                                      kpe.shape[1] if kpe is not None else 0, scale, _ptr(qhi), _ptr(qlo), _ptr(pos_a),
                                      pos_a.shape[1] if pos_a is not None else 0, nh, _ptr(g_w0), _ptr(g_b0), _ptr(b_o),
                                      g_w0.shape[0] if g_w0 is not None else 0, _ptr(r0), _ptr(state), _stream()), "hicom_query_prep_fwd")


def merge_vproj(part_m, part_l, part_acc, rows, w_v, po, out_ml=None, out_ctx=None):
    nparts, rows_pad = part_m.shape
    E = part_acc.shape[-1]
    _check(lib().hicom_merge_vproj_fwd(_ptr(part_m), _ptr(part_l), _ptr(part_acc), nparts, rows, rows_pad, E, _ptr(w_v), _ptr(po),
                                       _ptr(out_ml), _ptr(out_ctx), _stream()), "hicom_merge_vproj_fwd")


def merge_vproj_fixed(part_m, part_l, part_acc, rows, w_v, o_fix, out_ml=None, out_ctx=None):
    """o_fix int64 [E], zero on entry: fixed-point sums of the partial v_proj outputs (see include/hicom_hip.h)."""
    nparts, rows_pad = part_m.shape
    E = part_acc.shape[-1]
    _check(lib().hicom_merge_vproj_fixed_fwd(_ptr(part_m), _ptr(part_l), _ptr(part_acc), _dt(part_acc), nparts, rows, rows_pad, E, _ptr(w_v), _ptr(o_fix),
                                             _ptr(out_ml), _ptr(out_ctx), _stream()), "hicom_merge_vproj_fixed_fwd")


def splice_rows(row_src, dst):
    """dst [nrows, hidden] <- rows at the device addresses in row_src (int64 device tensor; 0 = zero row)."""
    nrows = row_src.numel()
    _check(lib().hicom_splice_rows_fwd(_ptr(row_src), nrows, dst.shape[-1] * dst.element_size(), _ptr(dst), _stream()),
           "hicom_splice_rows_fwd")


def splice_labels(labels, mask, idx_map, new_len, S, ignore_index, new_labels, new_mask):
    B, Lmax = idx_map.shape
    _check(lib().hicom_splice_labels_fwd(_ptr(labels), _ptr(mask), mask.element_size() if mask is not None else 0, _ptr(idx_map),
                                         _ptr(new_len), B, S, Lmax, ignore_index, _ptr(new_labels), _ptr(new_mask), _stream()),
           "hicom_splice_labels_fwd")


def row_ln(x, norm, out, mul=None, add=None, src=None, alpha=None, eps=1e-6):
    """out = (1-alpha)*src + alpha*LN(x*(1+mul)+add); mul/add rows broadcast when they have one row."""
    x2, o2 = x.reshape(-1, x.shape[-1]), out.reshape(-1, out.shape[-1])
    M, E = x2.shape
    bstride = lambda t: 0 if (t is None or t.reshape(-1, E).shape[0] == 1) else E
    s2 = src.reshape(-1, E) if src is not None else None
    if s2 is not None and s2.shape[0] == 1 and M > 1:
        raise HicomNativeError("row_ln: blend source must have one row per output row")
    _check(lib().hicom_row_ln_fwd(_ptr(x2), _dt(x2), E, _ptr(mul), bstride(mul), _ptr(add), bstride(add),
                                  _ptr(norm.weight.detach()), _ptr(norm.bias.detach()), _dt(norm.weight),
                                  _ptr(s2), _dt(s2) if s2 is not None else 0, E,
                                  _ptr(alpha), _dt(alpha) if alpha is not None else 0, eps,
                                  _ptr(o2), _dt(o2), E, M, E, _stream()), "hicom_row_ln_fwd")


def small_mha(q, k, v, nh, out, scale=None):
    """scale None: head_dim^-1/2 (ref :145); a float: the logit scale of the clip form (q, k already L2-normalised, ref :184-191)."""
    M, E = q.shape
    if scale is None:
        _check(lib().hicom_small_mha_fwd(_ptr(q), _ptr(k), _ptr(v), M, k.shape[0], nh, E // nh, _ptr(out), _stream()),
               "hicom_small_mha_fwd")
    else:
        _check(lib().hicom_small_mha_scaled_fwd(_ptr(q), _ptr(k), _ptr(v), M, k.shape[0], nh, E // nh, float(scale), _ptr(out), _stream()),
               "hicom_small_mha_scaled_fwd")
