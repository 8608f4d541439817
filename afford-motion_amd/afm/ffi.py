"""ctypes binding of libafm_hip.so (include/afm_hip.h) - the thin FFI layer between the
Python host code and the hand-written gfx950 kernels.

There is NO fallback: if the library is missing or a call fails, an exception is raised.
Tensors are passed as raw device pointers (``tensor.data_ptr()``) plus explicit sizes; work is
enqueued on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libafm_hip.so")
_lib = None

ACT_NONE, ACT_GELU, ACT_RELU, ACT_SILU = 0, 1, 2, 3
ARITH_DEFAULT, ARITH_F32, ARITH_BF16X1, ARITH_BF16X6, ARITH_BF16X9 = 0, 1, 3, 6, 9       # afm_linear_args.arith (include/afm_hip.h)
TUNE_NO_DMA, TUNE_TILE_SHIFT = 0x1, 4
CMDM_NO_L0_CACHE, CMDM_FUSED_LN, CMDM_NO_LN_FOLD, CMDM_ALL_QUERIES, CMDM_CLIP_X0, CMDM_NO_RIDERS, CMDM_PAIR_LAUNCH = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40
CDM_NO_GEN, CDM_CHAIN_SIDE, CDM_DEC_CHUNKS_SHIFT, CDM_CLIP_X0, CDM_PIPELINE = 0x2, 0x4, 12, 0x8, 0x10
ABI_VERSION = 7
MAX_LAYERS = 16

c_f32p = C.c_void_p
i32, i64, u64 = C.c_int32, C.c_int64, C.c_uint64


class AfmError(RuntimeError):
    pass


class LinearArgs(C.Structure):
    _fields_ = [
        ("A", c_f32p), ("lda", i64), ("W", c_f32p), ("ldw", i64), ("C", c_f32p), ("ldc", i64),
        ("M", i32), ("N", i32), ("K", i32),
        ("bias", c_f32p), ("scale", c_f32p), ("residual", c_f32p), ("ldr", i64),
        ("rowtab", c_f32p), ("rowtab_period", i32), ("act", i32), ("act_post", i32),
        ("a_grp", i32), ("a_stride", i32), ("a_off", i32),
        ("c_grp", i32), ("c_stride", i32), ("c_off", i32),
        ("ddpm_xt", c_f32p), ("ddpm_noise", c_f32p), ("ddpm_out", c_f32p), ("ldx", i64),
        ("ddpm_c1", c_f32p), ("ddpm_c2", c_f32p), ("ddpm_sigma", c_f32p), ("rows_per_sample", i32),
        # training hooks (ABI v2)
        ("preact", c_f32p), ("ldp", i64), ("dact_z", c_f32p), ("ldz", i64), ("dact", i32),
        ("drop_p", C.c_float), ("drop_seed", u64), ("drop_id", C.c_uint32), ("drop_after", i32),
        # arithmetic of the product + bit-neutral tuning (ABI v3: fields instead of a process-wide switch)
        ("arith", i32), ("arith_min_n", i32), ("tune", i32),
        # fused row-dot epilogue (ABI v4)
        ("rowdot_w", c_f32p), ("rowdot_out", c_f32p), ("rowdot_n", i32),
        # fused LayerNorm of the output rows (ABI v5)
        ("ln_gamma", c_f32p), ("ln_beta", c_f32p), ("ln_out", c_f32p), ("ldo", i64), ("ln_eps", C.c_float), ("ln_counters", C.c_void_p),
        # LayerNorm folded across kernel boundaries (ABI v5)
        ("stat_out", c_f32p), ("a_stat", c_f32p), ("a_stat_groups", i32), ("a_fold_g", c_f32p),
        ("res_stat", c_f32p), ("res_gamma", c_f32p), ("res_beta", c_f32p), ("ln_eps2", C.c_float),
        # a hole inside every group of the row remaps (ABI v6)
        ("a_skip_after", i32), ("a_skip", i32), ("c_skip_after", i32), ("c_skip", i32),
        ("ddpm_clip", i32),
        ("ddpm_out2", c_f32p), ("ldx2", i64),
        ("aux_src", c_f32p), ("aux_idx", C.c_void_p), ("aux_add", c_f32p), ("aux_dst", c_f32p), ("aux_dst_ld", i64),
        ("aux_rows", i32), ("aux_cols", i32), ("aux_idx_max", i32),
    ]


class WgradArgs(C.Structure):
    _fields_ = [
        ("dY", c_f32p), ("lddy", i64), ("X", c_f32p), ("ldx", i64), ("dW", c_f32p), ("lddw", i64), ("db", c_f32p),
        ("M", i32), ("N", i32), ("K", i32),
        ("dy_grp", i32), ("dy_stride", i32), ("dy_off", i32), ("x_grp", i32), ("x_stride", i32), ("x_off", i32),
        ("accumulate", i32), ("ws", C.c_void_p), ("ws_bytes", i64),
    ]


class PtAttentionArgs(C.Structure):
    _fields_ = [("p", c_f32p), ("qkv", c_f32p), ("knn_idx", C.c_void_p), ("out", c_f32p),
                ("n", i32), ("channels", i32), ("nsample", i32), ("share_planes", i32)] + \
               [(n, c_f32p) for n in ("lp0_w", "lp0_b", "lp_bn_scale", "lp_bn_shift", "lp3_w", "lp3_b", "w0_bn_scale",
                                      "w0_bn_shift", "w2_w", "w2_b", "w3_bn_scale", "w3_bn_shift", "w5_w", "w5_b",
                                      "out_scale", "out_shift")] + [("relu", i32)]


class Lin(C.Structure):
    _fields_ = [("w", c_f32p), ("b", c_f32p)]


class Ln(C.Structure):
    _fields_ = [("g", c_f32p), ("b", c_f32p)]


class MhaW(C.Structure):
    _fields_ = [("q", Lin), ("k", Lin), ("v", Lin), ("o", Lin)]


class MlpW(C.Structure):
    _fields_ = [("norm", Ln), ("fc1", Lin), ("fc2", Lin)]


class CdmWeights(C.Structure):
    _fields_ = [
        ("contact_dim", i32), ("feat_dim", i32), ("dq", i32), ("dkv", i32), ("enc_heads", i32), ("dec_heads", i32),
        ("n_self", i32), ("text_dim", i32), ("time_dim", i32), ("n_timesteps", i32),
        ("time_q0", c_f32p), ("time_u", c_f32p), ("time_cu", c_f32p),
        ("language_adapter", Lin), ("time_embedding_adapter", Lin), ("encoder_adapter", Lin), ("decoder_adapter", Lin),
        ("enc_q_norm", Ln), ("enc_kv_norm", Ln), ("enc_attn", MhaW), ("enc_mlp", MlpW),
        ("self_norm", Ln * 4), ("self_attn", MhaW * 4), ("self_mlp", MlpW * 4),
        ("dec_q_norm", Ln), ("dec_kv_norm", Ln), ("dec_attn", MhaW), ("dec_mlp", MlpW),
        ("contact_layer", Lin),
        ("gemm_arith", i32), ("gemm_arith_min_n", i32),
        # weight products of the folded sampling form (ABI v4; all five or none)
        ("fold_xu", c_f32p), ("fold_xv", c_f32p), ("fold_w2", c_f32p), ("flags", i32), ("fold_q", c_f32p), ("fold_c0", c_f32p),
        # generator tables of the two adapters (ABI v5; all three or none)
        ("gen_qe", c_f32p),
        ("dec_c", c_f32p), ("dec_twx", c_f32p), ("dec_qxx", c_f32p), ("dec_qdd", c_f32p),
        ("enc_ec", c_f32p), ("enc_qee", c_f32p), ("enc_wove", c_f32p), ("enc_c1", c_f32p),
        ("dec_dwq", c_f32p), ("dec_wqb", c_f32p), ("dec_wco", c_f32p), ("dec_wow", c_f32p), ("dec_wog", c_f32p), ("dec_xwo", c_f32p),
        ("lat_fold", C.c_void_p),
    ]


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", i64), ("total_ms", C.c_double), ("total_work", C.c_double)]


class EncoderLayerWeights(C.Structure):
    _fields_ = [(n, c_f32p) for n in (
        "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b",
        "norm1_w", "norm1_b", "norm2_w", "norm2_b", "lin1_wg", "lin1_g", "lin1_c", "in_proj_wg", "in_proj_g", "in_proj_c")]


class CmdmWeights(C.Structure):
    _fields_ = [
        ("d", i32), ("heads", i32), ("ff", i32), ("n_layers", i32), ("motion_dim", i32), ("n_cond", i32),
        ("motion_adapter_w", c_f32p), ("motion_adapter_b", c_f32p),
        ("motion_layer_w", c_f32p), ("motion_layer_b", c_f32p),
        ("time_table", c_f32p), ("n_timesteps", i32), ("pos_table", c_f32p),
        ("layer", EncoderLayerWeights * MAX_LAYERS),
        ("gemm_arith", i32), ("gemm_arith_min_n", i32), ("attn_group_waves", i32), ("flags", i32),
        ("motion_adapter_kpad", i32),
        ("motion_layer_wg", c_f32p), ("motion_layer_g", c_f32p), ("motion_layer_c", c_f32p),
    ]


class DdpmArgs(C.Structure):
    _fields_ = [("noise", c_f32p), ("x_next", c_f32p), ("c1", c_f32p), ("c2", c_f32p), ("sigma", c_f32p),
                ("seed", u64), ("sample_index0", i64), ("step", i32)]


EXPORTS = {
    # name: (restype, argtypes)
    "afm_version": (C.c_int, []),
    "afm_linear": (C.c_int, [C.POINTER(LinearArgs), C.c_void_p]),
    "afm_linear_pair": (C.c_int, [C.POINTER(LinearArgs), C.POINTER(LinearArgs), C.c_void_p]),
    "afm_mha_fwd": (C.c_int, [c_f32p, C.c_void_p, c_f32p, i32, i32, i32, i32, C.c_void_p]),
    "afm_mha_fwd_grouped": (C.c_int, [c_f32p, C.c_void_p, c_f32p, i32, i32, i32, i32, i32, C.c_void_p]),
    "afm_clamp": (C.c_int, [c_f32p, i64, C.c_float, C.c_float, C.c_void_p]),
    "afm_mha_fwd_rows": (C.c_int, [c_f32p, C.c_void_p, c_f32p, i32, i32, i32, i32, i32, i32, C.c_void_p]),
    "afm_mha_fwd_arith": (C.c_int, [c_f32p, C.c_void_p, c_f32p, i32, i32, i32, i32, i32, i32, i32, C.c_void_p]),
    "afm_mha_cross_fwd": (C.c_int, [c_f32p, c_f32p, C.c_void_p, c_f32p, i32, i32, i32, i32, i32, C.c_void_p]),
    "afm_layernorm": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, i64, i32, C.c_float, C.c_void_p]),
    "afm_layernorm_rows": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, i64, i32, C.c_float, i32, i32, i32, C.c_void_p]),
    "afm_ddpm_step": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i32, i64, u64, i64, i32, C.c_void_p]),
    "afm_randn": (C.c_int, [c_f32p, i32, i64, u64, i64, i32, C.c_void_p]),
    "afm_bn_fold": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_float, c_f32p, c_f32p, c_f32p, i32, C.c_void_p]),
    "afm_contact_glue": (C.c_int, [c_f32p, c_f32p, i64, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "afm_masked_mse": (C.c_int, [c_f32p, c_f32p, C.c_void_p, c_f32p, i32, i32, i32, C.c_void_p]),
    "afm_transpose": (C.c_int, [c_f32p, c_f32p, i32, i32, C.c_void_p]),
    "afm_linear_wgrad_workspace_bytes": (i64, [i32, i32, i32]),
    "afm_linear_wgrad": (C.c_int, [C.POINTER(WgradArgs), C.c_void_p]),
    "afm_layernorm_bwd_workspace_bytes": (i64, [i64, i32]),
    "afm_layernorm_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i64, i32, C.c_float, C.c_float, u64,
                                    C.c_uint32, C.c_void_p, i64, C.c_void_p]),
    "afm_mha_fwd_train": (C.c_int, [c_f32p, C.c_void_p, c_f32p, c_f32p, i32, i32, i32, i32, C.c_float, u64, C.c_uint32, C.c_void_p]),
    "afm_mha_bwd": (C.c_int, [c_f32p, C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, i32, i32, C.c_float, u64, C.c_uint32,
                              C.c_void_p, i64, C.c_void_p]),
    "afm_mha_cross_fwd_train": (C.c_int, [c_f32p, c_f32p, C.c_void_p, c_f32p, c_f32p, i32, i32, i32, i32, i32, C.c_float, u64, C.c_uint32, C.c_void_p]),
    "afm_mha_cross_bwd": (C.c_int, [c_f32p, c_f32p, C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, i32, i32, i32, C.c_float, u64,
                                   C.c_uint32, C.c_void_p, i64, C.c_void_p]),
    "afm_masked_mse_bwd": (C.c_int, [c_f32p, c_f32p, C.c_void_p, c_f32p, c_f32p, i32, i32, i32, C.c_void_p]),
    "afm_rowop": (C.c_int, [c_f32p, c_f32p, i32, c_f32p, i32, c_f32p, i64, i32, C.c_float, u64, C.c_uint32, C.c_void_p]),
    "afm_colstats_workspace_bytes": (i64, [i64, i32]),
    "afm_colstats": (C.c_int, [c_f32p, i64, i32, c_f32p, C.c_void_p, i64, C.c_void_p]),
    "afm_bn_finalize": (C.c_int, [c_f32p, i32, i64, c_f32p, c_f32p, C.c_float, C.c_float, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i32,
                                  C.c_void_p]),
    "afm_colaffine": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, i32, c_f32p, i64, i32, C.c_void_p]),
    "afm_bn_bwd_stats": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i64, i32, c_f32p, C.c_void_p, i64, C.c_void_p]),
    "afm_bn_bwd_apply": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i64, c_f32p, c_f32p, i64, i32, C.c_void_p]),
    "afm_group_points": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_void_p, c_f32p, i64, i32, i32, C.c_void_p]),
    "afm_scatter_add_rows": (C.c_int, [c_f32p, i64, i32, C.c_void_p, c_f32p, i64, i32, C.c_void_p]),
    "afm_group_max": (C.c_int, [c_f32p, c_f32p, C.c_void_p, i64, i32, i32, C.c_void_p]),
    "afm_group_max_bwd": (C.c_int, [c_f32p, C.c_void_p, c_f32p, i64, i32, i32, C.c_void_p]),
    "afm_scatter_plan_words": (C.c_int64, [i64, i64]),
    "afm_scatter_plan": (C.c_int, [C.c_void_p, i64, i64, C.c_void_p, C.c_void_p]),
    "afm_segment_sum_rows": (C.c_int, [c_f32p, i64, i32, i32, C.c_void_p, C.c_void_p, c_f32p, i64, i32, C.c_void_p]),
    "afm_group_sum": (C.c_int, [c_f32p, c_f32p, i64, i32, i32, C.c_float, C.c_void_p]),
    "afm_pt_w0": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, i64, i32, i32, C.c_void_p]),
    "afm_pt_aggregate": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i64, i32, i32, i32, C.c_void_p]),
    "afm_pt_aggregate_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i64, i32, i32, i32, C.c_void_p]),
    "afm_xq_workspace_bytes": (i64, [i32, i32, i32]),
    "afm_xq_attention_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, i32, i32, C.c_float, u64, C.c_uint32, C.c_void_p, i64,
                                       C.c_void_p]),
    "afm_xq_attention_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, i32, i32, C.c_float, u64,
                                       C.c_uint32, C.c_void_p, i64, C.c_void_p]),
    "afm_xk_workspace_bytes": (i64, [i32, i32, i32]),
    "afm_xk_attention_fwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, i32, i32, C.c_float, u64, C.c_uint32, C.c_void_p]),
    "afm_xk_attention_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, i32, i32, C.c_float, u64, C.c_uint32,
                                       C.c_void_p, i64, C.c_void_p]),
    "afm_adamw_multi": (C.c_int, [C.c_void_p, i32, i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i32, C.c_void_p]),
    "afm_adamw": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i32, C.c_void_p]),
    "afm_fps": (C.c_int, [c_f32p, i32, i32, i32, C.c_void_p, C.c_void_p]),
    "afm_knn": (C.c_int, [i32, c_f32p, c_f32p, i32, i32, i32, C.c_void_p, c_f32p, C.c_void_p]),
    "afm_knn_workspace_bytes": (C.c_int64, [i32, i32, i32, i32]),
    "afm_knn_ws": (C.c_int, [i32, c_f32p, c_f32p, i32, i32, i32, C.c_void_p, c_f32p, C.c_void_p, i64, C.c_void_p]),
    "afm_gather_rows": (C.c_int, [c_f32p, C.c_void_p, c_f32p, i64, i32, C.c_void_p]),
    "afm_interpolate": (C.c_int, [c_f32p, C.c_void_p, c_f32p, c_f32p, c_f32p, i64, i32, i32, C.c_void_p]),
    "afm_segment_mean": (C.c_int, [c_f32p, c_f32p, i32, i32, i32, C.c_void_p]),
    "afm_interpolate_bwd": (C.c_int, [c_f32p, C.c_void_p, c_f32p, c_f32p, i64, i64, i32, i32, C.c_void_p]),
    "afm_transition_down": (C.c_int, [c_f32p, c_f32p, i32, c_f32p, C.c_void_p, i32, c_f32p, i32, c_f32p, c_f32p, c_f32p,
                                      i32, C.c_void_p]),
    "afm_pt_attention": (C.c_int, [C.POINTER(PtAttentionArgs), C.c_void_p]),
    "afm_cdm_workspace_bytes": (i64, [C.POINTER(CdmWeights), i32, i32]),
    "afm_cdm_forward": (C.c_int, [C.POINTER(CdmWeights), c_f32p, c_f32p, C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p,
                                  C.POINTER(DdpmArgs), i32, i32, C.c_void_p, i64, C.c_void_p]),
    "afm_cdm_forward_overlap": (C.c_int, [C.POINTER(CdmWeights), c_f32p, c_f32p, C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p,
                                          C.POINTER(DdpmArgs), i32, i32, C.c_void_p, i64, C.c_void_p, C.c_void_p]),
    "afm_cdm_loop_workspace_bytes": (i64, [C.POINTER(CdmWeights), i32, i32, i32]),
    "afm_cdm_sample_loop": (C.c_int, [C.POINTER(CdmWeights), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, c_f32p, c_f32p, c_f32p,
                                      i32, u64, i64, i32, i32, C.c_void_p, C.c_void_p, i64, i32, C.POINTER(C.c_void_p), C.c_void_p]),
    "afm_cdm_sample_loop_range": (C.c_int, [C.POINTER(CdmWeights), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, c_f32p, c_f32p,
                                            c_f32p, i32, i32, u64, i64, i32, i32, C.c_void_p, C.c_void_p, i64, i32, C.POINTER(C.c_void_p),
                                            C.c_void_p]),
    "afm_cdm_latent_tokens": (C.c_int, [C.POINTER(CdmWeights), i32, c_f32p, i32, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "afm_profile_enable": (C.c_int, [i32]),
    "afm_profile_read": (C.c_int, [C.POINTER(ProfileEntry), i32]),
    "afm_cmdm_workspace_bytes": (i64, [C.POINTER(CmdmWeights), i32, i32]),
    "afm_cmdm_forward": (C.c_int, [C.POINTER(CmdmWeights), c_f32p, C.c_void_p, c_f32p, C.c_void_p, c_f32p,
                                   C.POINTER(DdpmArgs), i32, i32, C.c_void_p, i64, C.c_void_p]),
    "afm_cmdm_sched_scratch_bytes": (i64, [i32, i32]),
    "afm_cmdm_loop_workspace_bytes": (i64, [C.POINTER(CmdmWeights), i32, i32, i32]),
    "afm_cmdm_sample_loop": (C.c_int, [C.POINTER(CmdmWeights), c_f32p, c_f32p, C.c_void_p, c_f32p, C.c_void_p, c_f32p,
                                       c_f32p, c_f32p, i32, u64, i64, i32, i32, C.c_void_p, C.c_void_p, i64, i32,
                                       C.POINTER(C.c_void_p), C.c_void_p]),
    "afm_cmdm_sample_loop_range": (C.c_int, [C.POINTER(CmdmWeights), c_f32p, c_f32p, C.c_void_p, c_f32p, C.c_void_p, c_f32p,
                                             c_f32p, c_f32p, i32, i32, u64, i64, i32, i32, C.c_void_p, C.c_void_p, i64, i32,
                                             C.POINTER(C.c_void_p), C.c_void_p]),
}


def lib_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load (once) and type every export.  Raises AfmError when the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise AfmError(f"{_LIB_PATH} not found: build it with `python afford-motion_amd/build_hip.py` "
                       "(the HIP path has no CPU or eager fallback)")
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = {-1: "bad argument", -2: "workspace too small", -3: "unsupported shape"}.get(rc, f"hipError {rc}")
        raise AfmError(f"{what} failed: {kind}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_of(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def sched_scratch(owner, n_steps: int, batch: int, device) -> torch.Tensor:
    """The expanded-schedule scratch of a native sampling loop (afm_cmdm_sched_scratch_bytes), kept on `owner` (the model) across calls and
    grown geometrically from 1 MiB (1000 steps x 32 samples need 0.64 MB): a loop call allocates NOTHING after the first one, whatever its
    step count - a fresh device allocation inside a short loop call costs tens of milliseconds (tools/probe_k20_first.py: the first 20-step
    call 100 ms instead of 46).  Calls are stream-ordered, so the next call's expand kernel cannot overtake the previous call's readers."""
    need = int(load().afm_cmdm_sched_scratch_bytes(n_steps, batch))
    if need < 0:
        check(need, "afm_cmdm_sched_scratch_bytes")
    cache = owner.__dict__.setdefault("_afm_sched", {})
    buf = cache.get(str(device))
    if buf is None or buf.numel() < need:
        size = 1 << 20
        while size < need:
            size *= 2
        buf = cache[str(device)] = torch.empty(size, dtype=torch.uint8, device=device)
    return buf


_HIP_RT = None


_STREAM_POOL = {}


def stream_pool(device, n: int):
    """The first `n` side streams of the PROCESS-WIDE pool of `device` (created on demand, never destroyed).  The HIP runtime multiplexes its
    streams onto a handful of hardware queues (four by default): kernels of two streams that share a queue run strictly one after the other,
    whatever the events between them say (round 6: the CDM pipeline with private streams per model - seven alive in the process - ran its
    "concurrent" latent chains serialised behind the point kernels, 2.7 k against 5.8 k steps/s).  Every loop of this package therefore
    takes its side streams from ONE pool, so that the streams alive in a sampling process stay at the null stream + at most three."""
    key = str(device)
    pool = _STREAM_POOL.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def cu_masked_stream(device, mask_words):
    """A HIP stream restricted to the compute units whose bits are set in `mask_words` (uint32 words, bit i = CU i as the runtime numbers
    them), wrapped so that torch sees it (`.cuda_stream`, events).  hipExtStreamCreateWithCUMask lives in the HIP runtime the process has
    already loaded (libamdhip64): the library of this package neither creates nor owns streams."""
    global _HIP_RT
    if _HIP_RT is None:
        _HIP_RT = C.CDLL("libamdhip64.so")
        _HIP_RT.hipExtStreamCreateWithCUMask.restype = C.c_int
        _HIP_RT.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    words = [int(w) & 0xFFFFFFFF for w in mask_words]
    arr = (C.c_uint32 * len(words))(*words)
    handle = C.c_void_p()
    with torch.cuda.device(device):
        rc = _HIP_RT.hipExtStreamCreateWithCUMask(C.byref(handle), len(words), arr)
    if rc != 0 or not handle.value:
        raise AfmError(f"hipExtStreamCreateWithCUMask failed (hipError {rc})")
    return torch.cuda.ExternalStream(handle.value, device=device)


def require_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise AfmError("afford-motion_amd runs on the MI355X HIP path only; got a CPU tensor "
                           "(the CPU oracle lives in oracle/ and is test infrastructure)")


def f32c(t: torch.Tensor) -> torch.Tensor:
    """float32 + contiguous view/copy of ``t``."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def profile_enable(on: bool) -> None:
    check(load().afm_profile_enable(1 if on else 0), "afm_profile_enable")


def profile_read():
    """-> {kernel name: dict(launches, total_ms, total_work)} since the last read (synchronises)."""
    buf = (ProfileEntry * 32)()
    n = load().afm_profile_read(buf, 32)
    if n < 0:
        check(n, "afm_profile_read")
    return {buf[i].name.decode(): dict(launches=buf[i].launches, total_ms=buf[i].total_ms, total_work=buf[i].total_work)
            for i in range(n)}


def progress_slices(n_steps: int, progress: bool, slices: int = 50):
    """Executed-step ranges [(j0, j1), ...] of a native sampling loop: one range without a progress bar, ~`slices` ranges with one
    (the bar advances between native enqueues; gaussian_diffusion.py:520-523's tqdm over the step indices)."""
    if not progress:
        return [(0, n_steps)]
    per = max(1, -(-n_steps // slices))
    return [(j, min(j + per, n_steps)) for j in range(0, n_steps, per)]


def cut_slices(ranges, cuts):
    """Split the (j0, j1) ranges at every executed-step count in `cuts` (chained slices stay bit-identical to the whole loop)."""
    out = []
    for j0, j1 in ranges:
        for c in cuts:
            if j0 < c < j1:
                out.append((j0, c))
                j0 = c
        out.append((j0, j1))
    return out


def run_slices(ranges, enqueue, progress: bool, device):
    """Enqueue every slice; with a progress bar, advance it as the device finishes each slice (one event per slice, the next slice is
    already queued while the host waits, so the GPU never idles on the bar)."""
    import torch
    if not progress:
        for j0, j1 in ranges:
            enqueue(j0, j1)
        return
    from tqdm.auto import tqdm
    bar = tqdm(total=ranges[-1][1])
    pending = None
    for j0, j1 in ranges:
        enqueue(j0, j1)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        if pending is not None:
            pending[0].synchronize()
            bar.update(pending[1])
        pending = (ev, j1 - j0)
    pending[0].synchronize()
    bar.update(pending[1])
    bar.close()
