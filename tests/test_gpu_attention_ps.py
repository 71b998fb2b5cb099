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
    hands the attention an fp32 qkv tensor: equal to fp32 rounding through all twelve blocks.  (The 197-token blocks take the
    pre-split kernel only with HGL_ATTN_PS_CLIPBLOCKS=2 -- read once per process: this test runs the forward in a child.)"""
    import subprocess, sys, os
    if ops.default_precision() != "f16x3":
        pytest.skip("pre-split attention belongs to the split-fp16 mode")
    code = (
        "import sys, torch, numpy as np; sys.path.insert(0, %r)\n"
        "from hybridgl_amd import _lib, weights\n"
        "from hybridgl_amd.backbone import CLIPViTFM\n"
        "from oracle.cases import views_for_case\n"
        "dev = torch.device('cuda:0'); lib = _lib.load()\n"
        "model = CLIPViTFM('ViT-B/16', state_dict=weights.clip_state_dict('ViT-B/16', 0), device=dev)\n"
        "loc, glo, masks = views_for_case(6, 224, 160, 200)\n"
        "args = (torch.from_numpy(loc).to(dev), torch.from_numpy(glo).to(dev), torch.from_numpy(masks).to(dev))\n"
        "ys = []\n"
        "for on in (0, 1):\n"
        "    lib.hgl_attention_presplit(on); ys.append(model(*args, masking_block=9, fusion_mode='G2L').clone())\n"
        "d = float((ys[0] - ys[1]).abs().max()); m = float(ys[0].abs().max())\n"
        "assert torch.isfinite(ys[0]).all() and 0 < d <= 2e-5 * m, (d, m)\n"
        "print('PS_CLIP_OK', d, m)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HGL_ATTN_PS_CLIPBLOCKS="2"))
    assert r.returncode == 0 and "PS_CLIP_OK" in r.stdout, r.stderr[-2000:]


def test_gem_tower_presplit_equals_fp32_input_path(cuda):
    """The GEM image tower at 448 x 448 (785 tokens: its plain residual blocks take the pre-split kernel by default) against
    the path that hands the attention an fp32 qkv tensor: features equal to fp32 rounding."""
    if ops.default_precision() != "f16x3":
        pytest.skip("pre-split attention belongs to the split-fp16 mode")
    from hybridgl_amd import gem as G
    from hybridgl_amd.backbone import CLIPViTFM
    model = CLIPViTFM("ViT-B/16", state_dict=weights.clip_state_dict("ViT-B/16", 0), device=cuda)
    gm = G.create_gem_model("ViT-B/16", clip=model)
    x = torch.from_numpy(np.random.default_rng(3).standard_normal((3, 448, 448)).astype(np.float32)).to(cuda)
    f0, f1 = _ab(lambda: gm.image_features(x).clone())
    assert torch.isfinite(f0).all() and float(f0.abs().max()) > 0
    assert float((f0 - f1).abs().max()) <= 2e-5 * float(f0.abs().max()), float((f0 - f1).abs().max())
