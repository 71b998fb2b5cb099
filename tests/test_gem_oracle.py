"""CPU: the GEM heat-map oracle (oracle/gem_oracle.py) and the host-side pieces of hybridgl_amd.gem.

gem_torch itself is absent (parity unpinned, see the oracle header).  What CAN be pinned is pinned here:
* the three torch resampling operators the stage relies on, against torch itself (the arithmetic the reference calls);
* the original-stream output of the GEM ViT against the CLIP oracle that is pinned to the reference's goldens;
* structural properties of the published algorithm (no GEM blocks -> plain ViT; heat maps in [0, 1]).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hybridgl_amd import gem as G
from hybridgl_amd import weights
from oracle import clip_oracle as O
from oracle import gem_oracle as GO


@pytest.mark.parametrize("h,w,H,W", [(448, 448, 480, 640), (448, 448, 300, 400), (448, 448, 640, 427), (64, 64, 37, 91),
                                     (448, 448, 448, 448), (32, 32, 5, 200)])
def test_resize_antialias_oracle_matches_torch(h, w, H, W):
    """T.Resize((H, W), antialias=True) on a float tensor == F.interpolate(bilinear, antialias=True)"""
    x = np.random.default_rng(h + W).standard_normal((2, h, w)).astype(np.float32)
    ref = F.interpolate(torch.from_numpy(x)[None], size=(H, W), mode="bilinear", antialias=True, align_corners=False)[0].numpy()
    np.testing.assert_allclose(GO.resize_bilinear_aa(x, H, W), ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize("g,R", [(28, 448), (8, 128), (14, 224), (7, 100)])
def test_bilinear_upsample_oracle_matches_torch(g, R):
    x = np.random.default_rng(g).standard_normal((3, g, g)).astype(np.float32)
    ref = F.interpolate(torch.from_numpy(x)[None], size=(R, R), mode="bilinear")[0].numpy()
    np.testing.assert_allclose(O.bilinear_resize(x, R, R), ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize("n,g", [(14, 28), (4, 8), (14, 20), (7, 14), (14, 14)])
def test_pos_embedding_interpolation_matches_torch(n, g):
    """the DINO-style bicubic interpolation (scale factors (g + 0.1) / n) in the oracle AND in the shipped host code"""
    D = 24
    pos = np.random.default_rng(n * g).standard_normal((n * n + 1, D)).astype(np.float32)
    patch = torch.from_numpy(pos[1:]).reshape(1, n, n, D).permute(0, 3, 1, 2)
    if n == g:
        ref = pos
    else:
        up = F.interpolate(patch, scale_factor=((g + 0.1) / n, (g + 0.1) / n), mode="bicubic")
        assert up.shape[-2:] == (g, g)
        ref = np.concatenate([pos[:1], up.permute(0, 2, 3, 1).reshape(-1, D).numpy()], axis=0)
    np.testing.assert_allclose(GO.interpolate_pos_encoding(pos, g, g), ref, rtol=0, atol=1e-5)
    np.testing.assert_allclose(G.interpolate_pos_encoding(pos, g, g), ref, rtol=0, atol=1e-5)


def test_img_transform_layout():
    """gem.get_gem_img_transform: [3, 448, 448] fp32, OpenAI mean / std; a constant image stays constant"""
    img = np.full((37, 53, 3), 128, dtype=np.uint8)
    t = G.get_gem_img_transform()(img)
    assert t.shape == (3, 448, 448) and t.dtype == torch.float32
    want = (128 / 255.0 - np.asarray(G.OPENAI_MEAN)) / np.asarray(G.OPENAI_STD)
    np.testing.assert_allclose(t[:, 5, 7].numpy(), want, rtol=0, atol=1e-6)
    assert G.get_gem_img_transform(224)(img).shape == (3, 224, 224)
    assert G.GEMWrapper.prompts(["cat", "left dog"]) == ["a photo of a cat.", "a photo of a left dog."] == GO.gem_prompts(["cat", "left dog"])


def test_gem_vit_original_stream_is_the_pinned_clip_vit():
    """The original stream of the GEM ViT is the plain CLIP ViT: at the checkpoint's own resolution (no
    position-embedding interpolation) its CLS feature equals the 'crop' mode of the CLIP oracle, which is pinned
    to the reference's goldens.  With no GEM block both streams coincide."""
    sd = weights.clip_state_dict("tiny", 0)
    img = np.random.default_rng(3).standard_normal((2, 3, 64, 64)).astype(np.float32)
    gem, ori = GO.gem_vit_forward(sd, img)
    plain = O.clip_hybrid_forward(sd, img, None, None, fusion_mode="crop")
    np.testing.assert_allclose(ori[:, 0], plain, rtol=0, atol=2e-6)
    assert np.abs(gem - ori).max() > 1e-3            # the GEM stream is a different function
    g1, o1 = GO.gem_vit_forward(sd, img, gem_depth=1)
    assert np.array_equal(g1, o1)
    np.testing.assert_allclose(o1, ori, rtol=0, atol=2e-6)


def test_gem_heatmap_properties():
    sd = weights.clip_state_dict("tiny", 0)
    img = np.random.default_rng(4).standard_normal((1, 3, 128, 128)).astype(np.float32)
    gem, _ = GO.gem_vit_forward(sd, img)
    assert gem.shape == (1, 65, 32)
    txt = np.random.default_rng(5).standard_normal((3, 32)).astype(np.float32)
    heat = GO.gem_heatmap(gem[0], txt, 128)
    assert heat.shape == (3, 128, 128)
    assert np.allclose(heat.reshape(3, -1).min(1), 0) and np.allclose(heat.reshape(3, -1).max(1), 1)
    # min-max removes any positive rescaling of the text embedding and the 100x factor
    np.testing.assert_allclose(GO.gem_heatmap(gem[0], txt * 7, 128), heat, rtol=0, atol=1e-5)
    raw = GO.gem_heatmap(gem[0], txt, 128, normalize=False)
    assert np.abs(raw).max() <= 100.0 + 1e-3
    # iterating the self-self attention or fixing its temperature changes the result but keeps it finite
    for kw in (dict(ss_attn_iter=2), dict(ss_attn_temp=3.0), dict(ss_attn_iter=0)):
        g2, _ = GO.gem_vit_forward(sd, img, **kw)
        assert np.isfinite(g2).all() and np.abs(g2 - gem).max() > 1e-4


# ----------------------------------------------------------------------------- cv2.GaussianBlur fixed-point taps (host code)
@pytest.mark.parametrize("sigma", [0.0, 0.8, 2.6, 5.0])
def test_cv_gaussian_taps_host_vs_oracle(sigma):
    """hgl_cv_gaussian_kernel_q8 (host C++ in libhybridgl.so) == oracle/cv_oracle.py for every odd size: 8.8 fixed
    point, symmetric, sum exactly 256; OpenCV's tabulated small kernels for sigma <= 0."""
    from hybridgl_amd import ops
    from oracle import cv_oracle as CV
    for k in range(1, 32, 2):
        taps = list(ops.cv_gaussian_kernel_q8(k, sigma))
        assert taps == CV.gaussian_kernel_q8(k, sigma), (k, sigma)
        assert sum(taps) == 256 and taps == taps[::-1]
    if sigma == 0.0:
        assert list(ops.cv_gaussian_kernel_q8(3)) == [64, 128, 64]
        assert list(ops.cv_gaussian_kernel_q8(5)) == [16, 64, 96, 64, 16]
        assert list(ops.cv_gaussian_kernel_q8(7)) == [8, 28, 56, 72, 56, 28, 8]
        assert list(ops.cv_gaussian_kernel_q8(15)) == [1, 3, 6, 12, 20, 30, 36, 40, 36, 30, 20, 12, 6, 3, 1]


def test_cv_blur_oracle_properties():
    from oracle import cv_oracle as CV
    flat = np.full((20, 30, 3), 200, dtype=np.uint8)
    assert np.array_equal(CV.gaussian_blur_u8(flat), flat)
    img = np.zeros((31, 31, 1), dtype=np.uint8)
    img[15, 15] = 255
    b = CV.gaussian_blur_u8(img).astype(np.int64)
    t = np.asarray(CV.gaussian_kernel_q8(15), dtype=np.int64)
    assert np.array_equal(b[8:23, 8:23, 0], (np.outer(t, t) * 255 + 32768) >> 16)      # impulse response = tap outer product
