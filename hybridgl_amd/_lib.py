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
    "hgl_gemm_f32": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _LL, _LL, _LL, _LL, _I, _VP]),
    "hgl_layernorm_f32": (_I, [_VP, _VP, _VP, _VP, _I, _I, _F, _VP]),
    "hgl_attention_f32": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _I, _I, _I, _LL, _LL, _LL, _LL,
                               _F, _I, _VP, _I, _I, _VP, _VP, _I, _I, _VP]),
    "hgl_clip_hybrid_workspace_bytes": (_SZ, [C.POINTER(HglClipVisionW), _I, _I, _I, _I]),
    "hgl_clip_hybrid_forward": (_I, [C.POINTER(HglClipVisionW), _VP, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP,
                                     _VP, _SZ, _VP]),
    "hgl_clip_text_workspace_bytes": (_SZ, [C.POINTER(HglClipTextW), _I]),
    "hgl_clip_encode_text": (_I, [C.POINTER(HglClipTextW), _VP, _I, _VP, _VP, _SZ, _VP]),
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
}

_lib = None


def load():
    """Load libhybridgl.so (once). Raises HybridGLError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HybridGLError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C hybridgl_amd/csrc). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.hgl_abi_version() != 1:
        raise HybridGLError("libhybridgl ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().hgl_last_error().decode("utf-8", "replace")
        raise HybridGLError(f"{what} failed (code {rc}): {msg}")
