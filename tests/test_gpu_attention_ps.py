"""The attention kernel on pre-split operands (csrc/attention_ps.hip: q | k | v as the fp16 hi / lo planes the in-projection
GEMM writes, staged by LDS-DMA) against the kernels it replaces, which split the fp32 tensor themselves: the same products on
the same hi / lo values in the same order, so the results must be IDENTICAL -- through the callers (SAM encoder blocks,
CLIP residual blocks), which is also where the planes' layout (pad rows of the window partition, row maps) is exercised."""
import numpy as np
import pytest
import torch

from hybridgl_amd import _lib, ops, weights
from hybridgl_amd import sam as hsam
from hybridgl_amd.synth import synth_image

pytestmark = pytest.mark.gpu


def _presplit(on):
    return _lib.load().hgl_attention_presplit(on)


def _ab(fn):
    old = _presplit(-1)
    try:
        _presplit(0)
        a = fn()
        _presplit(1)
        b = fn()
    finally:
        _presplit(old)
    return a, b


@pytest.mark.parametrize("nb", [1, 3])
def test_sam_encoder_blocks_presplit_equal_fp32_input_kernels(cuda, nb):
    """ViT-H width, one windowed (14 x 14, head dim 80, rel-pos tables in the kernel, pad rows) and one global block
    (image_encoder.py:166-182, 224-240, 325-361)."""
    if ops.default_precision() != "f16x3":
        pytest.skip("pre-split attention belongs to the split-fp16 mode")
    cfg = weights.SAM_CONFIGS["vit_h_d2"]
    m = hsam.Sam(weights.sam_state_dict("vit_h_d2", 0), cfg, cuda)
    imgs = [torch.from_numpy(synth_image(1024, 1024 - 64 * i, 20 + i)).to(cuda) for i in range(nb)]
    e0, e1 = _ab(lambda: m.encode_batch(imgs).clone())
    assert torch.isfinite(e0).all() and float(e0.abs().max()) > 0
    assert torch.equal(e0, e1), float((e0 - e1).abs().max())


@pytest.mark.parametrize("B,H,S,hd,mask", [(40, 12, 197, 64, "none"), (40, 12, 197, 64, "cls_keep"), (3, 12, 197, 64, "cls_keep"),
                                           (5, 12, 785, 64, "none"), (8, 16, 196, 80, "none"), (2, 16, 1024, 80, "none"),
                                           (16, 12, 257, 64, "cls_keep"), (9, 12, 130, 64, "none")])
def test_presplit_kernels_equal_the_fp32_input_kernels(cuda, B, H, S, hd, mask):
    """hgl_attention_presplit_f32 (split planes, LDS-DMA staging, pipelined persistent kernel) against hgl_attention_f32 (the
    kernels that split q / k / v themselves) on CLIP's 197-token sequences with and without the CLS keep row of
    model/backbone.py:108-115 (ragged: 197 = 6 tiles + 5 keys), GEM's 785 tokens, a 14 x 14 window without bias and a
    1024-token sequence at head dim 80.  Same products on the same hi / lo values; the 197-token fp32-input kernel scales q
    before the split, these after (in the exponent): equal to fp32 rounding."""
    if ops.default_precision() != "f16x3":
        pytest.skip("pre-split attention belongs to the split-fp16 mode")
    ops.set_precision("f16x3")
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + S)
    qkv = (torch.randn(B, S, 3 * H * hd, generator=g) * 1.5).to(cuda)
    keep = None
    kw = {}
    if mask == "cls_keep":
        keep = (torch.rand(max(B - 1, 1), S - 1, generator=g) < 0.3).to(torch.uint8).to(cuda)
        keep[0, :] = 0                       # a sequence whose CLS row keeps nothing but itself
        kw = dict(keep=keep, keep_b0=1 if B > 1 else 0, keep_n=keep.shape[0])
    D = H * hd
    q, k, v = (qkv[..., i * D:(i + 1) * D].contiguous() for i in range(3))
    ref = ops.attention(q, k, v, H, mask=mask, **kw)
    got = ops.attention_presplit(qkv, H, mask=mask, **kw)
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max())
    assert err <= 4e-6 * max(1.0, float(ref.abs().max())), err


def test_clip_hybrid_forward_presplit_equals_fp32_input_path(cuda):
    """CLIPViTFM.forward (G2L, ViT-B/16, 6 masks) with the residual blocks' attention on split planes against the path that
    hands the attention an fp32 qkv tensor: equal to fp32 rounding through all twelve blocks."""
    if ops.default_precision() != "f16x3":
        pytest.skip("pre-split attention belongs to the split-fp16 mode")
    from hybridgl_amd.backbone import CLIPViTFM
    from oracle.cases import views_for_case
    model = CLIPViTFM("ViT-B/16", state_dict=weights.clip_state_dict("ViT-B/16", 0), device=cuda)
    loc, glo, masks = views_for_case(6, 224, 160, 200)
    args = (torch.from_numpy(loc).to(cuda), torch.from_numpy(glo).to(cuda), torch.from_numpy(masks).to(cuda))
    y0, y1 = _ab(lambda: model(*args, masking_block=9, fusion_mode="G2L").clone())
    assert torch.isfinite(y0).all()
    assert float((y0 - y1).abs().max()) <= 2e-5 * float(y0.abs().max()), float((y0 - y1).abs().max())
