"""hybridgl_amd -- MI355X-native hot path of the HybridGL referring-segmentation pipeline.

Python here is host glue that mirrors the reference's call surfaces (model/backbone.py
CLIPViTFM, the scoring helpers of utils.py / Hybridgl_main.py); all arithmetic runs in
libhybridgl.so (hand-written HIP for gfx950) through the C ABI in include/hybridgl.h.
"""
__version__ = "0.1.0"
