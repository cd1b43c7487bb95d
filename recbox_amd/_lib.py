"""ctypes binding of librecbox_hip.so (the C ABI declared in include/recbox_hip.h).

The product path has NO fallback: if the shared library is missing or does not
export a symbol, importing this module raises.  (``python -m recbox_amd.build``
or ``__graft_entry__.build()`` produces the library with hipcc for gfx950.)
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# RECBOX_HIP_LIB points at another build of the same C ABI (kernel A/B runs, system-wide installs)
LIB_PATH = os.environ.get("RECBOX_HIP_LIB") or os.path.join(HERE, "lib", "librecbox_hip.so")

RBX_MAX_FIELDS = 64
RBX_NO_ID = -(1 << 63)

RBX_OK, RBX_ERR_INVALID, RBX_ERR_LAUNCH, RBX_ERR_WORKSPACE, RBX_ERR_UNSUPPORTED = 0, -1, -2, -3, -4
RBX_I32, RBX_I64, RBX_F32, RBX_F64 = 0, 1, 2, 3
FIELD_CATEGORICAL, FIELD_NUMERIC, FIELD_DENSE = 0, 1, 2
POOL_NONE, POOL_SUM, POOL_MEAN_VALUE, POOL_MEAN_ID, POOL_SUM_ID, POOL_CONCAT = 0, 1, 2, 3, 4, 5
INTERACTION_MODES = {"product_sum": 0, "bi_interaction": 1, "inner_product": 2, "elementwise_product": 3}


class rbx_field_t(ctypes.Structure):
    _fields_ = [("ids", ctypes.c_void_p),
                ("table", ctypes.c_void_p),
                ("grad", ctypes.c_void_p),
                ("ids_stride_b", ctypes.c_int64),
                ("ids_stride_l", ctypes.c_int64),
                ("vocab", ctypes.c_int64),
                ("padding_idx", ctypes.c_int64),
                ("mask_id", ctypes.c_int64),
                ("out_off", ctypes.c_int64),
                ("dim", ctypes.c_int32),
                ("seq_len", ctypes.c_int32),
                ("ids_dtype", ctypes.c_int32),
                ("kind", ctypes.c_int32),
                ("pool", ctypes.c_int32),
                ("eps", ctypes.c_float),
                ("table_stride", ctypes.c_int64)]


class rbx_shard_geom_t(ctypes.Structure):
    _fields_ = [("world", ctypes.c_int32), ("dim", ctypes.c_int32), ("n_rows", ctypes.c_int32),
                ("n_pool", ctypes.c_int32), ("batch", ctypes.c_int64), ("cap_rows", ctypes.c_int64),
                ("cap_pool", ctypes.c_int64)]


class rbx_opt_t(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float),
                ("eps", ctypes.c_float), ("weight_decay", ctypes.c_float), ("d_step_size", ctypes.c_void_p)]


OPT_SGD, OPT_ADAGRAD, OPT_ADAM = 0, 1, 2


class rbx_rowcopy_t(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("row_bytes", ctypes.c_int64)]


_P = ctypes.c_void_p
_FP = ctypes.POINTER(rbx_field_t)
_RP = ctypes.POINTER(rbx_rowcopy_t)
_GP = ctypes.POINTER(rbx_shard_geom_t)
_OP = ctypes.POINTER(rbx_opt_t)
_PP = ctypes.POINTER(ctypes.c_void_p)          # host array of device pointers
_u64 = ctypes.c_uint64
_i32, _i64, _sz, _f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float

# name -> (restype, argtypes); must list every symbol of include/recbox_hip.h
SIGNATURES = {
    "rbx_last_error": (ctypes.c_char_p, []),
    "rbx_version": (ctypes.c_int, []),
    "rbx_embed_fwd": (ctypes.c_int, [_FP, _i32, _i64, _P, _i64, _P, _P, _P]),
    "rbx_embed_bwd_workspace_size": (_sz, [_FP, _i32, _i64]),
    "rbx_embed_sort": (ctypes.c_int, [_FP, _i32, _i64, _P, _sz, _P, _P]),
    "rbx_embed_bwd": (ctypes.c_int, [_FP, _i32, _i64, _P, _i64, _P, _i32, _P, _sz, _P]),
    "rbx_embed_bwd_indexed": (ctypes.c_int, [_FP, _i32, _i64, _P, _i64, _P, _P, _i32, _P, _sz, _P]),
    "rbx_shard_int_chunk": (_sz, [_GP]),
    "rbx_shard_float_rows": (_sz, [_GP]),
    "rbx_shard_route_workspace_size": (_sz, [_GP, _i32]),
    "rbx_shard_route": (ctypes.c_int, [_GP, _FP, _FP, _P, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "rbx_shard_serve": (ctypes.c_int, [_GP, _P, _P, _i64, _P, _P, _P, _P, _P]),
    "rbx_shard_combine_fwd": (ctypes.c_int, [_GP, _P, _P, _P, _P, _i64, ctypes.POINTER(ctypes.c_int64), _P]),
    "rbx_shard_combine_bwd": (ctypes.c_int, [_GP, _P, _i64, ctypes.POINTER(ctypes.c_int64), _P, _P, _P, _P]),
    "rbx_interaction_fwd": (ctypes.c_int, [_P, _i64, _i64, _i32, _i32, _i32, _P, _P]),
    "rbx_interaction_bwd": (ctypes.c_int, [_P, _i64, _P, _i64, _i32, _i32, _i32, _P, _i64, _P]),
    "rbx_fm_fwd": (ctypes.c_int, [_FP, _FP, _i32, _i64, _P, _P, _i32, _i32, _i32, _P, _i64, _P, _P, _P, _P, _P]),
    "rbx_fm_extra_bwd": (ctypes.c_int, [_P, _P, _P, _i64, _i32, _i32, _i32, _i32, _P, _i64, _P, _P]),
    "rbx_route_workspace_size": (_sz, [_i64, _i32]),
    "rbx_route": (ctypes.c_int, [_FP, _i32, _i64, _i32, _i64, _P, _P, _P, _P, _P, _sz, _P]),
    "rbx_route32": (ctypes.c_int, [_FP, _i32, _i64, _i32, _i64, _P, _P, _P, _P, _P, _sz, _P]),
    "rbx_fm_bwd_workspace_size": (_sz, [_FP, _FP, _i32, _i64]),
    "rbx_fm_sort": (ctypes.c_int, [_FP, _FP, _i32, _i64, _P, _sz, _P, _P]),
    "rbx_fm_sort_phases": (ctypes.c_int, [_FP, _FP, _i32, _i64, _P, _sz, _P, _i32, _P]),
    "rbx_embed_sparse_update": (ctypes.c_int, [_FP, _i32, _i64, _P, _sz, _OP, _PP, _PP, _P]),
    "rbx_opt_advance": (ctypes.c_int, [_i32, _f32, _f32, _f32, _f32, _P, _P, _P]),
    "rbx_fm_sparse_update": (ctypes.c_int, [_FP, _FP, _i32, _i64, _P, _sz, _OP, _PP, _PP, _PP, _PP, _P]),
    "rbx_comm_bind": (ctypes.c_int, [_P, _P, _P, _P, _P]),
    "rbx_all_to_all": (ctypes.c_int, [_P, _P, _P, _sz, _i32, _P]),
    "rbx_comm_bind_collectives": (ctypes.c_int, [_P, _P, _P]),
    "rbx_all_reduce": (ctypes.c_int, [_P, _P, _P, _sz, _i32, _i32, _P]),
    "rbx_all_gather": (ctypes.c_int, [_P, _P, _P, _sz, _P]),
    "rbx_embed_rezero": (ctypes.c_int, [_FP, _i32, _i64, _P, _sz, _P]),
    "rbx_sort_share": (ctypes.c_int, [_FP, _FP, _i32, _i32, _P, _FP, _FP, _i32, _i32, _P, _sz, _i64, _P]),
    "rbx_rowscale": (ctypes.c_int, [_P, _P, _P, _i64, _i32, _f32, _P, _P]),
    "rbx_rowscale_seq": (ctypes.c_int, [_P, _P, _i64, _P, _i64, _i32, _f32, _P, _P]),
    "rbx_seq_colsum_workspace_size": (_sz, [_i64, _i32, _i32]),
    "rbx_seq_colsum": (ctypes.c_int, [_P, _P, _i64, _i32, _i32, _P, _P, _sz, _P]),
    "rbx_sum_prefix": (ctypes.c_int, [_P, _i64, _P, _i64, _P, _i64, _i64, _i32, _i32, _P, _i64, _P]),
    "rbx_fm_quad": (ctypes.c_int, [_i32]),
    "rbx_sort_chained": (ctypes.c_int, [_i32]),
    "rbx_fm_rezero": (ctypes.c_int, [_FP, _FP, _i32, _i64, _P, _sz, _P]),
    "rbx_fm_bwd": (ctypes.c_int, [_FP, _FP, _i32, _i64, _P, _P, _P, _i32, _i32, _P, _sz, _P]),
    "rbx_gatherdot_fwd": (ctypes.c_int, [_FP, _i32, _i64, _P, _i64, _f32, _P, _P, _P]),
    "rbx_gatherdot_bwd_workspace_size": (_sz, [_FP, _i32, _i64]),
    "rbx_gatherdot_sort": (ctypes.c_int, [_FP, _i32, _i64, _P, _sz, _P, _P]),
    "rbx_gatherdot_bwd": (ctypes.c_int, [_FP, _i32, _i64, _P, _i64, _P, _f32, _P, _i64, _i32, _P, _sz, _P]),
    "rbx_negsample": (ctypes.c_int, [_i64, _i64, _i32, _u64, _u64, _P, _P, _P, _P, _P, _P]),
    "rbx_negsample_checked": (ctypes.c_int, [_i64, _i64, _i32, _u64, _u64, _P, _P, _P, _P, _i64, _P, _P, _P]),
    "rbx_gather_rows": (ctypes.c_int, [_RP, _i32, _P, _i64, _i64, _P, _P]),
    "rbx_topk_workspace_size": (_sz, [_i64, _i64, _i32]),
    "rbx_topk": (ctypes.c_int, [_P, _P, _i64, _i64, _i64, _i32, _P, _P, _P, _sz, _P]),
    "rbx_penalize_members": (ctypes.c_int, [_P, _i64, _i32, _P, _P, _P, ctypes.c_double, _P, _P]),
    "rbx_membership": (ctypes.c_int, [_P, _i64, _i32, _P, _P, _P, _P, _P]),
    "rbx_batchnorm_workspace_size": (_sz, [_i64, _i32]),
    "rbx_batchnorm_fwd": (ctypes.c_int, [_P, _i64, _i32, _P, _P, _f32, _i32, _f32, _P, _P, _i32, _P, _P, _P, _P, _sz, _P]),
    "rbx_batchnorm_bwd": (ctypes.c_int, [_P, _P, _P, _i64, _i32, _P, _P, _P, _i32, _P, _P, _P, _P, _sz, _P]),
    "rbx_batchnorm_prelu_fwd": (ctypes.c_int, [_P, _i64, _i32, _P, _P, _P, _i32, _f32, _i32, _f32, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "rbx_batchnorm_prelu_bwd": (ctypes.c_int, [_P, _P, _i64, _i32, _P, _P, _P, _i32, _P, _P, _i32, _P, _P, _P, _P, _P, _sz, _P]),
    "rbx_batchnorm_stats": (ctypes.c_int, [_P, _i64, _i32, _P, _P, _sz, _P]),
    "rbx_batchnorm_apply": (ctypes.c_int, [_P, _i64, _i32, _P, _P, _P, _P, _i32, _P, _P]),
    "rbx_batchnorm_bwd_reduce": (ctypes.c_int, [_P, _P, _P, _i64, _i32, _P, _P, _P, _P, _P, _sz, _P]),
    "rbx_batchnorm_bwd_dx": (ctypes.c_int, [_P, _P, _P, _i64, _i32, _P, _P, _P, _P, _P, _i64, _P, _P]),
    "rbx_cin_outer_fwd": (ctypes.c_int, [_P, _P, _i64, _i32, _i32, _i32, _P, _P]),
    "rbx_cin_outer_bwd": (ctypes.c_int, [_P, _P, _P, _i64, _i32, _i32, _i32, _P, _P, _P]),
    "rbx_act_workspace_size": (_sz, [_i64, _i32]),
    "rbx_prelu_fwd": (ctypes.c_int, [_P, _i64, _i32, _P, _i32, _P, _P]),
    "rbx_prelu_bwd": (ctypes.c_int, [_P, _P, _i64, _i32, _P, _i32, _P, _P, _P, _sz, _P]),
    "rbx_dropout": (ctypes.c_int, [_P, _i64, _f32, _u64, _P, _P, _P]),
    "rbx_dice_fwd": (ctypes.c_int, [_P, _i64, _i32, _P, _f32, _i32, _f32, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "rbx_dice_bwd": (ctypes.c_int, [_P, _P, _i64, _i32, _P, _P, _P, _i32, _P, _P, _P, _sz, _P]),
    "rbx_layernorm_fwd": (ctypes.c_int, [_P, _i64, _i32, _P, _P, _f32, _P, _P, _P, _P]),
    "rbx_layernorm_bwd_workspace_size": (_sz, [_i64, _i32]),
    "rbx_layernorm_bwd": (ctypes.c_int, [_P, _P, _i64, _i32, _P, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "rbx_cross_fwd": (ctypes.c_int, [_P, _P, _P, _P, _i64, _i32, _i32, _P, _P]),
    "rbx_cross_bwd": (ctypes.c_int, [_P, _P, _P, _i64, _i32, _i32, _P, _P, _P]),
    "rbx_bce_workspace_size": (_sz, [_i64]),
    "rbx_bce_mean_fwd": (ctypes.c_int, [_P, _P, _i64, _P, _P, _sz, _P]),
    "rbx_bce_mean_bwd": (ctypes.c_int, [_P, _P, _P, _i64, _P, _P]),
    "rbx_sigmoid_bce_mean": (ctypes.c_int, [_P, _P, _i64, _f32, _P, _P, _P, _P, _sz, _P]),
    "rbx_scale_by_scalar": (ctypes.c_int, [_P, _P, _i64, _P, _P]),
    "rbx_sigmoid_bce_mean_onepass": (ctypes.c_int, [_P, _P, _i64, _f32, _P, _P, _P, _P, _sz, _P, _P]),
    "rbx_pairmul_fwd": (ctypes.c_int, [_P, _P, _i64, _i32, _i32, _i32, _P, _P]),
    "rbx_pairmul_bwd": (ctypes.c_int, [_P, _P, _P, _i64, _i32, _i32, _i32, _P, _P, _P]),
    "rbx_l2norm_fwd": (ctypes.c_int, [_P, _i64, _i32, _f32, _P, _P, _P]),
    "rbx_l2norm_fwd_strided": (ctypes.c_int, [_P, _i64, _i64, _i64, _i32, _f32, _P, _P, _P]),
    "rbx_cosdot_fwd": (ctypes.c_int, [_P, _P, _i64, _i64, _i32, _i32, _f32, _f32, _P, _P, _P]),
    "rbx_cosdot_bwd": (ctypes.c_int, [_P, _P, _i64, _P, _P, _i64, _i32, _i32, _f32, _P, _P, _i64, _P]),
    "rbx_l2norm_bwd": (ctypes.c_int, [_P, _P, _P, _i64, _i32, _P, _P]),
    "rbx_pairdot_fwd": (ctypes.c_int, [_P, _P, _i64, _i32, _i32, _f32, _P, _P]),
    "rbx_pairdot_bwd": (ctypes.c_int, [_P, _P, _P, _i64, _i32, _i32, _f32, _P, _P, _P]),
    "rbx_linear_fwd": (ctypes.c_int, [_P, _i64, _P, _P, _i64, _i32, _i32, _i32, _P, _P]),
    "rbx_linear_bwd_workspace_size": (_sz, [_i64, _i32, _i32, _i32]),
    "rbx_linear_fwd_fused": (ctypes.c_int, [_P, _i64, _P, _P, _i64, _i32, _i32, _i32, _P, _i64, _P, _P, _i64, _P]),
    "rbx_linear_dx_fused": (ctypes.c_int, [_P, _i64, _P, _i64, _i32, _i32, _P, _i64, _P, _i64, _P, _i64, _P]),
    "rbx_linear_dx_scaled": (ctypes.c_int, [_P, _i64, _P, _i64, _i32, _i32, _P, _i64, _P, _i64, _P, _P, _i64, _P]),
    "rbx_seqblock_qkv_fwd": (ctypes.c_int, [_P, _i64, _P, _P, ctypes.c_float, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rbx_seqblock_ffn_fwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _i64, _P, _P, ctypes.c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                            _P, _P, _P, _P, _P, _P]),
    "rbx_seqblock_ffn_bwd_workspace_size": (_sz, [_i64]),
    "rbx_seqblock_ffn_bwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _i64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _sz, _P]),
    "rbx_seqblock_attn_in_bwd_workspace_size": (_sz, [_i64]),
    "rbx_seqblock_attn_in_bwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _i64, _P, _P, _P, _f32, _P, _P, _P, _P, _sz, _P]),
    "rbx_seqblock_attn_out_bwd_workspace_size": (_sz, [_i64]),
    "rbx_seqblock_attn_out_bwd": (ctypes.c_int, [_P, _P, _i64, _P, _P, _P, _P, _P, _sz, _P]),
    "rbx_seqblock_inproj_dw_workspace_size": (_sz, [_i64]),
    "rbx_seqblock_inproj_dw": (ctypes.c_int, [_P, _P, _P, _P, _P, _i64, _P, _P, _P, _P, _P, _sz, _P]),
    "rbx_linear_dwdb_scaled": (ctypes.c_int, [_P, _i64, _P, _i64, _P, _i64, _i32, _i32, _P, _P, _P, _sz, _P]),
    "rbx_fm_sum_fwd": (ctypes.c_int, [_P, _i64, _i64, _i32, _i32, _P, _P, _P]),
    "rbx_fm_sum_lr_fwd": (ctypes.c_int, [_P, _i64, _i64, _i32, _i32, _P, _P, _P, _P, _P, _P]),
    "rbx_linear_dx_deepfm": (ctypes.c_int, [_P, _i64, _P, _i64, _i32, _i32, _P, _i64, _P, _i32, _i32, _P, _P, _P, _P, _i64,
                                            _P]),
    "rbx_linear_bwd": (ctypes.c_int, [_P, _i64, _P, _P, _P, _i64, _i32, _i32, _i32, _P, _i64, _P, _P, _P, _sz, _P]),
    "rbx_split_bf16_size": (_sz, [_i32, _i32, _i32]),
    "rbx_split_bf16": (ctypes.c_int, [_P, _i64, _i32, _i32, _i32, _P, _P]),
    "rbx_split_register": (ctypes.c_int, [_P, _P, _i32, _i32, _i32]),
    "rbx_split_unregister": (ctypes.c_int, [_P]),
    "rbx_gemm_bx6_count": (ctypes.c_uint64, []),
    "rbx_attn_fwd": (ctypes.c_int, [_P, _P, _P, _P, _i64, _i32, _i32, _i32, _f32, _i32, _f32, _P, _P, _P, _P]),
    "rbx_attn_bwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _i64, _i32, _i32, _i32, _f32, _i32, _f32,
                                    _P, _P, _P, _P, _P]),
    "rbx_attn_dropout_fwd": (ctypes.c_int, [_P, _P, _P, _P, _i64, _i32, _i32, _i32, _f32, _i32, _f32, _f32, _u64, _P, _P, _P,
                                            _P, _P]),
    "rbx_attn_dropout_bwd": (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _i64, _i32, _i32, _i32, _f32, _i32, _f32, _f32, _u64,
                                            _P, _P, _P, _P, _P, _P]),
    "rbx_attn_dropout_mask": (ctypes.c_int, [_i64, _i32, _i32, _f32, _u64, _P, _P, _P]),
    "rbx_attn_packed_fwd": (ctypes.c_int, [_P, _i64, _P, _i64, _P, _i64, _i64, _i32, _i32, _i32, _f32, _i32, _f32, _u64, _P,
                                           _P, _i64, _P, _P]),
    "rbx_attn_packed_bwd": (ctypes.c_int, [_P, _i64, _P, _i64, _P, _i64, _P, _i64, _P, _i64, _P, _i64, _i32, _i32, _i32,
                                           _f32, _i32, _f32, _u64, _P, _P, _i64, _P, _i64, _P, _i64, _P, _P]),
    "rbx_loss_workspace_size": (_sz, [_i64]),
    "rbx_softmax_ce_fwd": (ctypes.c_int, [_P, _i64, _i64, _i32, _P, _P, _P, _P, _P, _sz, _P]),
    "rbx_softmax_ce_bwd": (ctypes.c_int, [_P, _i64, _i64, _i32, _P, _P, _P, _P, _P]),
    "rbx_pair_logsigmoid_fwd": (ctypes.c_int, [_P, _P, _P, _i64, _f32, _P, _P, _sz, _P]),
    "rbx_pair_logsigmoid_bwd": (ctypes.c_int, [_P, _P, _P, _P, _i64, _f32, _P, _P, _P]),
    "rbx_pool_fwd": (ctypes.c_int, [_P, _P, _i64, _i32, _i32, _i32, _i32, _f32, _P, _P, _P]),
    "rbx_pool_bwd": (ctypes.c_int, [_P, _P, _P, _i64, _i32, _i32, _i32, _P, _P]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "recbox_amd: %s is missing. Build it with `python -m recbox_amd.build` (hipcc, gfx950). "
            "There is no CPU or PyTorch fallback for the hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise ImportError("recbox_amd: %s does not export %s; rebuild the extension" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error():
    msg = lib.rbx_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Map a C status code to the exception type the reference raises."""
    if rc == RBX_OK:
        return
    msg = last_error()
    if rc == RBX_ERR_INVALID:
        raise ValueError(msg)
    if rc == RBX_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError("librecbox_hip: %s (code %d)" % (msg, rc))
