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
