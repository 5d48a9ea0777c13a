"""ctypes binding of libmsam_hip.so (C ABI declared in include/msam_hip.h).

Torch is used only as the owner of device memory and streams: every call passes raw ``tensor.data_ptr()``
values and the current HIP stream handle.  There is NO fallback: if the shared library is missing the import
of any compute entry point raises, and calling a kernel without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import build as _build

MSAM_MAX_BLOCKS = 32
F32, BF16, FP8, F16, U8, U16 = 1, 2, 3, 4, 5, 6
ACT_NONE, ACT_GELU, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
PROFILE_FAMILIES = 8          # include/msam_hip.h MSAM_PROFILE_FAMILIES


class GemmParams(C.Structure):
    _fields_ = [
        ("A", _vp), ("lda", _i64), ("W", _vp), ("ldw", _i64), ("M", _i32), ("N", _i32), ("K", _i32),
        ("bias", _vp), ("table", _vp), ("table_rows", _i32), ("table_cols", _i32), ("table_ld", _i64),
        ("resid", _vp), ("resid_dtype", _i32), ("resid_rows", _i32), ("ldr", _i64), ("act", _i32),
        ("out", _vp), ("out_dtype", _i32), ("ldc", _i64), ("out_mode", _i32),
        ("q", _vp), ("k", _vp), ("v", _vp), ("heads", _i32), ("head_dim", _i32), ("tokens", _i32),
        ("use_glds", _i32), ("ln_mode", _i32), ("ln_w", _vp), ("ln_b", _vp), ("ln_eps", _f32),
        ("a_dtype", _i32), ("row_scale", _vp), ("col_scale", _vp), ("split_k", _i32),
        ("ln_add", _vp), ("ln_out_a", _vp), ("ln_out_b", _vp),
    ]


class WsGemmParams(C.Structure):
    _fields_ = [
        ("A", _vp), ("W", _vp), ("M", _i32), ("N", _i32), ("K", _i32), ("bias", _vp),
        ("table", _vp), ("table_rows", _i32), ("table_cols", _i32), ("table_ld", _i64),
        ("resid", _vp), ("resid_rows", _i32), ("ldr", _i64),
        ("ln_mode", _i32), ("ln_w", _vp), ("ln_b", _vp), ("ln_eps", _f32),
        ("out", _vp), ("ldc", _i64), ("kv_split", _i32), ("k_out", _vp), ("vT_out", _vp), ("tokens", _i32),
        ("head_major", _i32),
    ]


class ImageLayerParams(C.Structure):
    _fields_ = [
        ("xin", _vp), ("q_shared", _vp), ("wq", _vp), ("bq", _vp), ("peq", _vp),
        ("wo", _vp), ("bo", _vp), ("ln_w", _vp), ("ln_b", _vp), ("ln_eps", _f32),
        ("ktok", _vp), ("vtok", _vp), ("Nt", _i32), ("out", _vp), ("rows", _i32),
    ]


_blk = _vp * MSAM_MAX_BLOCKS


class EncoderParams(C.Structure):
    _fields_ = [
        ("embed_dim", _i32), ("depth", _i32), ("heads", _i32), ("is_global", _i32 * MSAM_MAX_BLOCKS),
        ("patch_w", _vp), ("patch_b", _vp), ("pos_embed", _vp),
        ("ln1_w", _blk), ("ln1_b", _blk), ("qkv_w", _blk), ("qkv_b", _blk), ("rel_h", _blk), ("rel_w", _blk),
        ("proj_w", _blk), ("proj_b", _blk), ("ln2_w", _blk), ("ln2_b", _blk),
        ("lin1_w", _blk), ("lin1_b", _blk), ("lin2_w", _blk), ("lin2_b", _blk),
        ("neck0_w", _vp), ("neck1_w", _vp), ("neck1_b", _vp), ("neck2_w", _vp), ("neck3_w", _vp), ("neck3_b", _vp),
        ("use_glds", _i32), ("head_dim_stored", _i32), ("fp8", _i32),
        ("qkv_w8", _blk), ("qkv_cs", _blk), ("proj_w8", _blk), ("proj_cs", _blk),
        ("lin1_w8", _blk), ("lin1_cs", _blk), ("lin2_w8", _blk), ("lin2_cs", _blk),
        ("dtype16", _i32), ("split_io", _i32),
    ]


class AttnW(C.Structure):
    _fields_ = [("q_w", _vp), ("q_b", _vp), ("k_w", _vp), ("k_b", _vp), ("v_w", _vp), ("v_b", _vp),
                ("o_w", _vp), ("o_b", _vp)]


class TwoWayLayer(C.Structure):
    _fields_ = [("self_attn", AttnW), ("t2i", AttnW), ("i2t", AttnW),
                ("n1_w", _vp), ("n1_b", _vp), ("n2_w", _vp), ("n2_b", _vp), ("n3_w", _vp), ("n3_b", _vp),
                ("n4_w", _vp), ("n4_b", _vp),
                ("mlp1_w", _vp), ("mlp1_b", _vp), ("mlp2_w", _vp), ("mlp2_b", _vp), ("mlp1_ws", _vp), ("mlp2_ws", _vp)]


class DecoderParams(C.Structure):
    _fields_ = [
        ("pe_gauss", _vp), ("point_embed", _vp), ("not_a_point", _vp), ("no_mask", _vp), ("out_tokens", _vp),
        ("layer", TwoWayLayer * 2), ("final_attn", AttnW), ("nf_w", _vp), ("nf_b", _vp),
        ("up1_w", _vp), ("up1_b", _vp), ("up_ln_w", _vp), ("up_ln_b", _vp), ("up2_w", _vp), ("up2_b", _vp),
        ("hyp_w", (_vp * 3) * 4), ("hyp_b", (_vp * 3) * 4), ("iou_w", _vp * 3), ("iou_b", _vp * 3),
        ("use_glds", _i32), ("low_res_dtype", _i32), ("up1_centred", _i32),
    ]


class SGemmParams(C.Structure):
    """include/msam_hip.h msam_sgemm_t (the strict mode's fp32 product)."""
    _fields_ = [("A", _vp), ("lda", _i64), ("A2", _vp), ("lda2", _i64), ("a2_rows", _i64), ("W", _vp), ("ldw", _i64),
                ("M", _i64), ("N", _i32), ("K", _i32), ("bias", _vp), ("act", _i32), ("res", _vp), ("ldr", _i64), ("res_rows", _i64),
                ("out", _vp), ("ldc", _i64), ("col_scale", _vp), ("col_shift", _vp), ("conv_h", _i32), ("conv_w", _i32), ("conv_c", _i32),
                ("shuffle_h", _i32), ("shuffle_w", _i32), ("shuffle_c", _i32), ("a2_cols", _i32), ("split16", _i32), ("a_scale", _f32), ("w_scale", _f32), ("w_pairs", _vp)]


class SI2TParams(C.Structure):
    """include/msam_hip.h msam_si2t_t (the strict mode's fused image -> token block)."""
    _fields_ = [("keys", _vp), ("key_batch_stride", _i64), ("pos", _vp), ("wq", _vp), ("bq", _vp), ("tok_k", _vp), ("tok_v", _vp),
                ("ld_tok", _i64), ("tok_batch_stride", _i64), ("wo", _vp), ("bo", _vp), ("ln_weight", _vp), ("ln_bias", _vp),
                ("ln_eps", _f32), ("denom", _f32), ("out", _vp), ("B", _i32), ("Tk", _i32), ("split16", _i32), ("wq_scale", _f32), ("wo_scale", _f32), ("wq_pairs", _vp), ("wo_pairs", _vp)]


class ST2IParams(C.Structure):
    """include/msam_hip.h msam_st2i_t (split16: token -> image attention with the k / v projections folded into the token side)."""
    _fields_ = [("keys", _vp), ("key_batch_stride", _i64), ("pos", _vp), ("q", _vp), ("ldq", _i64), ("wk", _vp), ("wv", _vp), ("bv", _vp),
                ("denom", _f32), ("out", _vp), ("ldo", _i64), ("B", _i32), ("Tk", _i32), ("workspace", _vp), ("workspace_bytes", _i64)]


class SUp2Params(C.Structure):
    """include/msam_hip.h msam_sup2_t (split16: LayerNorm2d + GELU + ConvT2 + GELU + hyper product in one launch)."""
    _fields_ = [("u1", _vp), ("ln_weight", _vp), ("ln_bias", _vp), ("ln_eps", _f32), ("w2", _vp), ("b2", _vp), ("w_scale", _f32),
                ("hyper", _vp), ("hyper_ld", _i32), ("mask0", _i32), ("nmask", _i32), ("low_res", _vp), ("P", _i64)]


class MaskPromptParams(C.Structure):
    """include/msam_hip.h msam_mask_prompt_t: fp32 weights of prompt_encoder.mask_downscaling."""
    _fields_ = [(n, _vp) for n in ("c1_w", "c1_b", "ln1_w", "ln1_b", "c2_w", "c2_b", "ln2_w", "ln2_b", "c3_w", "c3_b")] + [("exact_gelu", _i32)]


_PROTOS = {
    "msam_last_error": (C.c_char_p, []),
    "msam_abi_version": (_i32, []),
    "msam_gemm_bf16": (_i32, [C.POINTER(GemmParams), _vp]),
    "msam_wsgemm_bf16": (_i32, [C.POINTER(WsGemmParams), _vp]),
    "msam_decoder_image_layer": (_i32, [C.POINTER(ImageLayerParams), _vp]),
    "msam_t2i_fold_workspace_bytes": (_i64, [_i32]),
    "msam_t2i_fold_attention": (_i32, [_vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "msam_upscale_fused": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "msam_uncrop_bits": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "msam_decoder_forward_masks": (_i32, [C.POINTER(DecoderParams), C.POINTER(MaskPromptParams), _vp, _vp, _vp, _vp, _i32, _vp, _vp,
                                          _i32, _i32, _vp, _vp, _vp, _i64, _vp]),
    "msam_i2t_fold_workspace_bytes": (_i64, [_i32]),
    "msam_i2t_fold_layer": (_i32, [_vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _i64, _vp]),
    "msam_profile_enable": (_i32, [_i32]),
    "msam_profile_collect": (_i32, [C.POINTER(_i32), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "msam_profile_collect_family": (_i32, [_i32 * PROFILE_FAMILIES, C.c_double * PROFILE_FAMILIES, C.c_double * PROFILE_FAMILIES,
                                          C.c_double * PROFILE_FAMILIES]),
    "msam_layernorm": (_i32, [_vp, _vp, _vp, _f32, _i64, _i32, _vp, _i32, _i32, _i32, _vp]),
    "msam_layernorm_backward": (_i32, [_vp, _vp, _vp, _f32, _i64, _i32, _vp, _vp, _vp, _vp]),
    "msam_attention_forward": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    "msam_attention_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "msam_relpos_attention_forward": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    "msam_relpos_attention_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp,
                                              _vp, _vp, _vp, _vp]),
    "msam_amg_generate_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "msam_amg_generate_labels": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, C.POINTER(_i32), _f32, _f32, _f32, _i32, _i32,
                                        _vp, _vp, _vp, _i64, _vp]),
    "msam_to_image": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "msam_patchify": (_i32, [_vp, _i32, _vp, _vp]),
    "msam_patchify_u8": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "msam_im2col3x3": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "msam_cast_f32_to_bf16": (_i32, [_vp, _vp, _i64, _vp]),
    "msam_cast_transpose": (_i32, [_vp, _i32, _i64, _i32, _i64, _vp, _vp, _vp, _vp]),
    "msam_gemm256_set_staging": (_i32, [_i32]),
    "msam_gemm_set_trace": (_i32, [_vp]),
    "msam_gemm_group_bf16": (_i32, [_vp, _i32, _vp]),
    "msam_fold_attn_set_dma": (_i32, [_i32]),
    "msam_tune_set": (_i32, [C.c_char_p, _i32]),
    "msam_i2t_fold_operand_bytes": (_i64, [_i32]),
    "msam_i2t_fold_operands": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp]),
    "msam_i2t0_t2i_workspace_bytes": (_i64, [_i32]),
    "msam_i2t0_t2i_fused": (_i32, [_vp, _vp, _vp, _vp, _f32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "msam_i2t01_fused": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _i32, _i32, _vp, _vp]),
    "msam_chain_tables2_bytes": (_i64, []),
    "msam_chain_prepare_tables2": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "msam_chain_const2_bytes": (_i64, []),
    "msam_chain_prepare_const2": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "msam_chain_prepare_tables2_c": (_i32, [_vp, _vp, _vp, _vp]),
    "msam_t2i_fold_values_bytes": (_i64, [_i32]),
    "msam_t2i_fold_values": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "msam_i2t_fold_operands_values": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "msam_i2t0_t2i_v2_workspace_bytes": (_i64, [_i32]),
    "msam_i2t0_t2i_fused_v2": (_i32, [_vp, _vp, _vp, _vp, _vp, _f32, _vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp]),
    "msam_chain_tables_bytes": (_i64, []),
    "msam_chain_prepare_tables": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "msam_upscale_fused_layout": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "msam_upscale_fused_out": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp]),
    "msam_upscale_set_prio": (_i32, [_i32]),
    "msam_layernorm_fp8": (_i32, [_vp, _vp, _vp, _f32, _i64, _i32, _vp, _vp, _vp]),
    "msam_quant_rows_fp8": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "msam_debug_i2t_timing": (_i32, [_i32, _vp]),
    "msam_window_attention": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, C.c_float, _vp, _vp]),
    "msam_global_attention": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, C.c_float, _vp, _vp]),
    "msam_window_attention16": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, C.c_float, _i32, _vp, _vp]),
    "msam_global_attention16": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, C.c_float, _i32, _vp, _vp]),
    "msam_patchify16": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "msam_patchify_u8_16": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "msam_cast_f32_to_16": (_i32, [_vp, _i32, _vp, _i64, _vp]),
    "msam_resample_u8": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp]),
    "msam_patchify_split16": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "msam_patchify_u8_split16": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "msam_im2col3x3_split16": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "msam_cast_f32_split16": (_i32, [_vp, _i32, _vp, _i64, _i32, _vp]),
    "msam_encoder_workspace_bytes": (_i64, [C.POINTER(EncoderParams), _i32]),
    "msam_encoder_forward": (_i32, [C.POINTER(EncoderParams), _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _i32, _vp]),
    "msam_decoder_dtype": (_i32, []),
    "msam_decoder_const_bytes": (_i64, []),
    "msam_decoder_prepare_const": (_i32, [C.POINTER(DecoderParams), _vp, _vp]),
    "msam_decoder_image_bytes": (_i64, []),
    "msam_decoder_prepare_image": (_i32, [C.POINTER(DecoderParams), _vp, _vp, _vp, _vp, _i64, _vp]),
    "msam_decoder_workspace_bytes": (_i64, [_i32]),
    "msam_prompt_encode": (_i32, [C.POINTER(DecoderParams), C.POINTER(MaskPromptParams), _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "msam_decoder_forward_embeddings": (_i32, [C.POINTER(DecoderParams), _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp,
                                               _i64, _vp]),
    "msam_decoder_forward": (_i32, [C.POINTER(DecoderParams), _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp,
                                    _i64, _vp]),
    "msam_postprocess_masks": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "msam_postprocess_masks16": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "msam_rle_run_counts": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "msam_rle_encode": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "msam_mask_nms": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp]),
    "msam_box_nms": (_i32, [_vp, _i32, _f32, _vp, _vp, _vp]),
    "msam_box_nms_valid": (_i32, [_vp, _vp, _i32, _f32, _vp, _vp, _vp]),
    "msam_paint_label_image_dev": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "msam_label_components_async": (_i32, [_vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    "msam_component_sizes": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "msam_labels_from_masks_workspace_bytes": (_i64, [_i32, _i32]),
    "msam_labels_from_masks": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _vp]),
    "msam_host_seeded_watershed": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "msam_slice_overlaps": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp]),
    "msam_paint_label_image": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "msam_label_components": (_i32, [_vp, _i32, _i32, _vp, _vp, _i32, C.POINTER(_i32), _vp]),
    "msam_strict_gemm": (_i32, [C.POINTER(SGemmParams), _vp]),
    "msam_strict_i2t_block": (_i32, [C.POINTER(SI2TParams), _vp]),
    "msam_strict_layernorm": (_i32, [_vp, _vp, _vp, _f32, _i64, _i32, _vp, _i32, _i32, _vp]),
    "msam_strict_relpos_attention": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "msam_split16_relpos_attention": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "msam_strict_attention": (_i32, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _i64,
                                     _i64, _vp]),
    "msam_strict_patchify": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "msam_strict_im2col3x3": (_i32, [_vp, _i32, _i32, _vp, _vp]),
    "msam_strict_source": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "msam_strict_hyper_masks": (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    "msam_strict_upscale2": (_i32, [C.POINTER(SUp2Params), _vp]),
    "msam_split16_t2i_attention": (_i32, [C.POINTER(ST2IParams), _vp]),
    "msam_split16_prepare_pairs": (_i32, [_vp, _i64, _i32, _f32, _i32, _vp, _vp]),
    "msam_split16_i2t_block": (_i32, [C.POINTER(SI2TParams), _vp, _i64, _vp]),
    "msam_strict_instance_norm": (_i32, [_vp, _i64, _i32, _i64, _i32, _f32, _vp, _vp, _i64, _vp]),
    "msam_strict_resize_bilinear": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _i64, _i32, _i32, _i32, _f32, _f32, _i32, _vp, _vp]),
}
OPTIONAL = set()

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    """libmsam_hip.so, or the ablation variant named by MSAM_LIB_VARIANT (build.variant_path; "decbf16" = bf16 decoder)."""
    return _build.variant_path(os.environ.get("MSAM_LIB_VARIANT", ""))


def load() -> C.CDLL:
    """Load (once) libmsam_hip.so; raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"micro_sam_amd: {path} is missing. Build it with `python -m micro_sam_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU / PyTorch fallback."
        )
    lib = C.CDLL(path)
    for name, (res, args) in _PROTOS.items():
        if not hasattr(lib, name):
            if name in OPTIONAL:
                continue
            raise RuntimeError(f"micro_sam_amd: {path} does not export {name}")
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    # A/B knobs from the command line: MSAM_TUNE="dec_chain=0,i2t_variant=0" (include/msam_hip.h msam_tune_set)
    for item in filter(None, os.environ.get("MSAM_TUNE", "").split(",")):
        key, _, val = item.partition("=")
        if lib.msam_tune_set(key.strip().encode(), int(val)) != 0:
            raise RuntimeError(f"micro_sam_amd: MSAM_TUNE: unknown knob {key!r}")
    return lib


def decoder_dtype() -> torch.dtype:
    """16-bit type of the mask decoder's weights and tensors in this build of the library (``msam_decoder_dtype``):
    torch.float16 (default) or torch.bfloat16 (``python -m micro_sam_amd.build --dec-bf16``)."""
    return torch.float16 if load().msam_decoder_dtype() == F16 else torch.bfloat16


def exported_symbols():
    return [n for n in _PROTOS if n not in OPTIONAL]


def check(status: int, what: str = "") -> None:
    if status == 0:
        return
    msg = load().msam_last_error().decode("utf-8", "replace")
    if status == 1:
        raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg}")


def require_gpu(device=None) -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("micro_sam_amd needs an AMD GPU (gfx950); torch.cuda.is_available() is False. "
                           "There is no CPU fallback.")
    return torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_cuda, "device tensor expected"
    return t.data_ptr()
