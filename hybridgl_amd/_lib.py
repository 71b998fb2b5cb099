"""ctypes binding of libhybridgl.so (the C ABI declared in include/hybridgl.h).

The library is the ONLY compute path of this package: there is no Python/CPU fallback.
Importing this module without the built library raises; calling any compute entry point
without a HIP device raises HybridGLError (HGL_ENODEVICE).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhybridgl.so")

c_float_p = C.POINTER(C.c_float)
c_u8_p = C.POINTER(C.c_uint8)
c_i32_p = C.POINTER(C.c_int32)
c_i64_p = C.POINTER(C.c_int64)


class HybridGLError(RuntimeError):
    pass


class HglSentence(C.Structure):
    """include/hybridgl.h HglSentence: one referring expression of a ref for hgl_score_ref"""
    _fields_ = [("sentence_row", C.c_int), ("noun_phrase_row", C.c_int), ("other_row0", C.c_int), ("n_other", C.c_int),
                ("dirflag", C.c_int), ("relaword", C.c_int), ("has_other_nouns", C.c_int), ("black", C.c_float),
                ("imgattn", C.c_void_p), ("target", C.c_void_p)]


class HglGroupRef(C.Structure):
    """include/hybridgl.h HglGroupRef: one ref of a group for hgl_score_group"""
    _fields_ = [("hybrid", C.c_void_p), ("text", C.c_void_p), ("T", C.c_int), ("boxes", C.c_void_p), ("masks", C.c_void_p),
                ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("sentences", C.POINTER(HglSentence)), ("S", C.c_int),
                ("k1", C.c_int), ("k2", C.c_int), ("idx", C.c_void_p), ("iu", C.c_void_p), ("score_clip", C.c_void_p),
                ("score_neg", C.c_void_p), ("gem_score", C.c_void_p)]


class HglResBlockW(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ln1_w", "ln1_b", "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b",
        "ln2_w", "ln2_b", "fc_w", "fc_b", "proj_w", "proj_b")]


class HglClipVisionW(C.Structure):
    _fields_ = [("width", C.c_int), ("layers", C.c_int), ("heads", C.c_int), ("patch", C.c_int),
                ("grid", C.c_int), ("embed", C.c_int),
                ("conv1_w", C.c_void_p), ("class_embedding", C.c_void_p),
                ("positional_embedding", C.c_void_p), ("ln_pre_w", C.c_void_p), ("ln_pre_b", C.c_void_p),
                ("blocks", C.POINTER(HglResBlockW)),
                ("ln_post_w", C.c_void_p), ("ln_post_b", C.c_void_p), ("proj_t", C.c_void_p)]


class HglClipTextW(C.Structure):
    _fields_ = [("width", C.c_int), ("layers", C.c_int), ("heads", C.c_int), ("context", C.c_int),
                ("vocab", C.c_int), ("embed", C.c_int),
                ("token_embedding", C.c_void_p), ("positional_embedding", C.c_void_p),
                ("blocks", C.POINTER(HglResBlockW)),
                ("ln_final_w", C.c_void_p), ("ln_final_b", C.c_void_p), ("text_projection_t", C.c_void_p)]


class HglSamBlockW(C.Structure):
    _fields_ = [("window", C.c_int), ("rel_len", C.c_int)] + [(n, C.c_void_p) for n in (
        "norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "rel_pos_h", "rel_pos_w",
        "norm2_w", "norm2_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b")]


class HglSamEncoderW(C.Structure):
    _fields_ = [("embed_dim", C.c_int), ("depth", C.c_int), ("heads", C.c_int), ("img_size", C.c_int),
                ("patch", C.c_int), ("out_chans", C.c_int),
                ("patch_w", C.c_void_p), ("patch_b", C.c_void_p), ("pos_embed", C.c_void_p),
                ("blocks", C.POINTER(HglSamBlockW)),
                ("neck0_w", C.c_void_p), ("neck1_w", C.c_void_p), ("neck1_b", C.c_void_p),
                ("neck2_w", C.c_void_p), ("neck3_w", C.c_void_p), ("neck3_b", C.c_void_p)]


class HglLinearW(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p)]


class HglNormW(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p)]


class HglSamAttnW(C.Structure):
    _fields_ = [("q", HglLinearW), ("k", HglLinearW), ("v", HglLinearW), ("out", HglLinearW), ("internal", C.c_int)]


class _Layer(C.Structure):
    _fields_ = [("self_attn", HglSamAttnW), ("t2i", HglSamAttnW), ("i2t", HglSamAttnW),
                ("n1", HglNormW), ("n2", HglNormW), ("n3", HglNormW), ("n4", HglNormW),
                ("lin1", HglLinearW), ("lin2", HglLinearW)]


class HglSamDecoderW(C.Structure):
    _fields_ = [("C", C.c_int), ("grid", C.c_int), ("heads", C.c_int), ("mlp_dim", C.c_int),
                ("pe_gauss", C.c_void_p), ("point_embed_pos", C.c_void_p), ("not_a_point", C.c_void_p),
                ("no_mask", C.c_void_p), ("dense_pe", C.c_void_p), ("iou_token", C.c_void_p),
                ("mask_tokens", C.c_void_p),
                ("layer", _Layer * 2), ("final_t2i", HglSamAttnW), ("norm_final", HglNormW),
                ("up0_w", C.c_void_p), ("up0_b", C.c_void_p), ("up1", HglNormW),
                ("up3_w", C.c_void_p), ("up3_b", C.c_void_p),
                ("hyper", (HglLinearW * 3) * 4), ("iou_head", HglLinearW * 3),
                ("kvq1_w", C.c_void_p), ("kvq1_b", C.c_void_p), ("kvq1_pe", C.c_void_p),
                ("kvf_w", C.c_void_p), ("kvf_b", C.c_void_p), ("kvf_pe", C.c_void_p),
                ("point_embed_neg", C.c_void_p), ("point_embed_box0", C.c_void_p), ("point_embed_box1", C.c_void_p),
                ("md_c1_w", C.c_void_p), ("md_c1_b", C.c_void_p), ("md_n1_w", C.c_void_p), ("md_n1_b", C.c_void_p),
                ("md_c2_w", C.c_void_p), ("md_c2_b", C.c_void_p), ("md_n2_w", C.c_void_p), ("md_n2_b", C.c_void_p),
                ("md_c3_w", C.c_void_p), ("md_c3_b", C.c_void_p)]



# name -> (restype, argtypes).  Must list EVERY symbol declared in include/hybridgl.h
# (tests/test_abi.py cross-checks this table against the header).
_VP, _I, _LL, _F, _SZ = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t
PROTOTYPES = {
    "hgl_abi_version": (_I, []),
    "hgl_last_error": (C.c_char_p, []),
    "hgl_device_count": (_I, []),
    "hgl_prof_enable": (_I, [_I]),
    "hgl_prof_read": (_I, [_I, C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double),
                           C.POINTER(C.c_double)]),
    "hgl_set_precision": (_I, [_I]),
    "hgl_get_precision": (_I, []),
    "hgl_gemm_f16x3_select": (_I, [_I]),
    "hgl_split_overflow_count": (_I, [_I, C.POINTER(C.c_ulonglong)]),
    "hgl_split_overflow_peek_async": (_I, [_VP, _VP]),
    "hgl_register_split_weight": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP]),
    "hgl_unregister_split_weight": (_I, [_VP]),
    "hgl_split_weight_is_fp16_valued": (_I, [_VP]),
    "hgl_gemm_f16x3": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _SZ, _VP]),
    "hgl_gemm_f32": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _LL, _LL, _LL, _LL, _I, _VP]),
    "hgl_layernorm_f32": (_I, [_VP, _VP, _VP, _VP, _I, _I, _F, _VP]),
    "hgl_attention_f32": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _LL, _LL, _LL, _LL,
                               _F, _I, _VP, _I, _I, _VP, _VP, _I, _I, _VP]),
    "hgl_clip_hybrid_workspace_bytes": (_SZ, [C.POINTER(HglClipVisionW), _I, _I, _I, _I]),
    "hgl_clip_hybrid_forward": (_I, [C.POINTER(HglClipVisionW), _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP,
                                     _VP, _SZ, _VP]),
    "hgl_clip_hybrid_forward_segments": (_I, [C.POINTER(HglClipVisionW), _VP, _VP, C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                              C.POINTER(C.c_int), C.POINTER(C.c_int), _I, _I, _I, _I, _I, _VP, _VP, _SZ, _VP]),
    "hgl_clip_text_workspace_bytes": (_SZ, [C.POINTER(HglClipTextW), _I]),
    "hgl_clip_encode_text": (_I, [C.POINTER(HglClipTextW), _VP, _I, _VP, _VP, _SZ, _VP]),
    "hgl_clip_encode_text_prefix": (_I, [C.POINTER(HglClipTextW), _VP, _I, _I, _VP, _VP, _SZ, _VP]),
    "hgl_clip_encode_text_ex": (_I, [C.POINTER(HglClipTextW), _VP, _I, _I, _VP, _VP, _I, _I, _VP, _VP, _SZ, _VP]),
    "hgl_gem_workspace_bytes": (_SZ, [C.POINTER(HglClipVisionW)]),
    "hgl_gem_image_features": (_I, [C.POINTER(HglClipVisionW), _VP, _I, _I, _F, _VP, _VP, _VP, _SZ, _VP]),
    "hgl_gem_batch_workspace_bytes": (_SZ, [C.POINTER(HglClipVisionW), _I]),
    "hgl_gem_image_features_batch": (_I, [C.POINTER(HglClipVisionW), _VP, _I, _I, _I, _F, _VP, _VP, _VP, _SZ, _VP]),
    "hgl_gem_heatmap_workspace_bytes": (_SZ, [_I, _I, _I]),
    "hgl_gem_heatmap": (_I, [_VP, _I, _I, _VP, _I, _I, _I, _VP, _VP, _SZ, _VP]),
    "hgl_resize_bilinear_aa": (_I, [_VP, _I, _I, _I, _VP, _I, _I, _VP]),
    "hgl_resize_bilinear": (_I, [_VP, _I, _I, _I, _VP, _I, _I, _VP]),
    "hgl_mask_resize": (_I, [_VP, _I, _I, _I, _I, _VP, _VP]),
    "hgl_calculate_score": (_I, [_VP, _VP, _I, _I, _I, _F, _VP, _VP]),
    "hgl_coherence_workspace_bytes": (_SZ, [_I, _I, _I]),
    "hgl_coherence_scores": (_I, [_VP, _VP, _I, _I, _I, _I, _F, _VP, _VP, _SZ, _VP]),
    "hgl_iou": (_I, [_VP, _VP, _LL, _VP, _VP]),
    "hgl_iou_select": (_I, [_VP, _VP, _I, _VP, _LL, _VP, _VP]),
    "hgl_score_sentence_workspace_bytes": (_SZ, [_I, _I]),
    "hgl_score_sentence": (_I, [_VP, _VP, _VP, _VP, _I, _F, _VP, _VP, _I, _I, _F, _I, _I, _F, _I, _I,
                                _VP, _VP, _VP, _VP, _SZ, _VP]),
    "hgl_synthesize_views": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP]),
    "hgl_resize_pil_bilinear_workspace_bytes": (_SZ, [_I, _I, _I]),
    "hgl_resize_pil_bilinear": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _VP, _I, _VP, _VP, _I, _VP, _VP, _SZ, _VP]),
    "hgl_sam_encode_workspace_bytes": (_SZ, [C.POINTER(HglSamEncoderW)]),
    "hgl_sam_encode": (_I, [C.POINTER(HglSamEncoderW), _VP, _I, _I, _VP, _VP, _SZ, _VP]),
    "hgl_sam_encode_batch_workspace_bytes": (_SZ, [C.POINTER(HglSamEncoderW), _I]),
    "hgl_sam_encode_batch": (_I, [C.POINTER(HglSamEncoderW), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), _I,
                                  _VP, _VP, _SZ, _VP]),
    "hgl_sam_dense_pe": (_I, [C.POINTER(HglSamDecoderW), _VP, _VP, _VP]),
    "hgl_sam_decode_workspace_bytes": (_SZ, [C.POINTER(HglSamDecoderW), _I]),
    "hgl_sam_decode_points": (_I, [C.POINTER(HglSamDecoderW), _VP, _VP, _I, _VP, _VP, _VP, _SZ, _VP]),
    "hgl_sam_decode_points_gated": (_I, [C.POINTER(HglSamDecoderW), _VP, _VP, _I, C.c_float, _VP, _VP, _VP, _SZ, _VP]),
    "hgl_sam_decode_prompts": (_I, [C.POINTER(HglSamDecoderW), _VP, _VP, _VP, _I, _VP, _I, _I, _VP, _VP, _VP, _SZ, _VP]),
    "hgl_sam_embed_masks": (_I, [C.POINTER(HglSamDecoderW), _VP, _I, _VP, _VP]),
    "hgl_sam_decoder_fusion": (_I, [_I]),
    "hgl_attention_presplit": (_I, [_I]),
    "hgl_attention_presplit_f32": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _I, C.c_float, _I, _VP, _I, _I, _VP, C.c_size_t, _VP]),
    "hgl_sam_postprocess_workspace_bytes": (_SZ, [_I]),
    "hgl_sam_postprocess": (_I, [_VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _F, _F, _VP, _VP, _VP, _VP, _VP,
                                 _VP, _SZ, _VP]),
    "hgl_nms": (_I, [_VP, _VP, _VP, _I, _F, _VP, _VP, _VP]),
    "hgl_nms_large_workspace_bytes": (_SZ, [_I]),
    "hgl_nms_large": (_I, [_VP, _VP, _VP, _I, _F, _VP, _VP, _VP, _SZ, _VP]),
    "hgl_box_near_crop_edge": (_I, [_VP, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _F, _VP, _VP]),
    "hgl_remove_small_regions_workspace_bytes": (_SZ, [_I, _I, _I]),
    "hgl_remove_small_regions": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _VP, _VP, _SZ, _VP]),
    "hgl_remove_small_regions_boxes": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "hgl_mask_boxes": (_I, [_VP, _I, _I, _I, _VP, _VP]),
    "hgl_gather_masks": (_I, [_VP, _VP, _VP, _I, _LL, _VP, _VP]),
    "hgl_gen_dir_mask": (_I, [_I, _I, _I, _VP, _VP]),
    "hgl_relation_boxes": (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP, _VP]),
    "hgl_gaussian_blur_u8_workspace_bytes": (_SZ, [_I, _I, _I]),
    "hgl_gaussian_blur_u8": (_I, [_VP, _I, _I, _I, C.POINTER(C.c_double), _I, _VP, _VP, _SZ, _VP]),
    "hgl_cv_gaussian_kernel_q8": (_I, [_I, C.c_double, C.POINTER(C.c_uint16)]),
    "hgl_gaussian_blur_u8_q8": (_I, [_VP, _I, _I, _I, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), _I, _VP, _VP, _SZ, _VP]),
    "hgl_score_ref_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "hgl_score_ref": (_I, [_VP, _VP, _I, _VP, _VP, _I, _I, _I, _I, C.POINTER(HglSentence), _I, C.c_float, C.c_float, _I, _I, C.c_float,
                           _VP, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "hgl_score_group_workspace_bytes": (_SZ, [C.POINTER(HglGroupRef), _I, _I]),
    "hgl_score_group": (_I, [C.POINTER(HglGroupRef), _I, _I, C.c_float, C.c_float, C.c_float, _VP, _VP, _SZ, _VP]),
    "hgl_u8_to_chw_lut": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP]),
    "hgl_gt_mask_from_polygons": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP]),
    "hgl_gt_mask_from_rle_counts": (_I, [_VP, _I, _I, _I, _VP, _VP]),
    "hgl_gt_mask_from_rle_string": (_I, [C.c_char_p, _I, _I, _VP, _VP]),
    "hgl_rle_encode_mask": (_I, [_VP, _I, _I, _VP, _LL, C.POINTER(C.c_longlong)]),
    "hgl_rle_to_string": (_I, [_VP, _LL, C.c_char_p, _SZ, C.POINTER(C.c_size_t)]),
}

_lib = None
ABI_VERSION = 7   # include/hybridgl.h HGL_ABI_VERSION


def load():

    """Load libhybridgl.so (once). Raises HybridGLError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HybridGLError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C hybridgl_amd/csrc). There is no CPU fallback.")
    # torch first: its wheel bundles the HIP runtime it was built with, and the process must hold ONE libamdhip64 -- loading
    # this library before torch pulls in the system copy as well, after which no device is visible to it
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.hgl_abi_version.restype = C.c_int
    if lib.hgl_abi_version() != ABI_VERSION:    # before any symbol lookup: a stale library fails with this message
        raise HybridGLError(f"{LIB_PATH} has ABI version {lib.hgl_abi_version()}, this package binds version {ABI_VERSION}: "
                            "rebuild it (make -C hybridgl_amd/csrc)")
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().hgl_last_error().decode("utf-8", "replace")
        raise HybridGLError(f"{what} failed (code {rc}): {msg}")
