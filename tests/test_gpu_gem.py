"""GPU parity of the GEM heat-map stage (hgl_gem_image_features, hgl_gem_heatmap, hgl_resize_bilinear_aa, the
`gem` package mirror of hybridgl_amd/gem.py) against the numpy oracle on the same seeded inputs and -- for the
resampling -- against torch itself.  gem_torch is absent: the oracle restates the published algorithm (parity
with the package unpinned, oracle/gem_oracle.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hybridgl_amd import gem as G
from hybridgl_amd import weights
from hybridgl_amd.backbone import CLIPViTFM
from oracle import clip_oracle as O
from oracle import gem_oracle as GO

pytestmark = pytest.mark.gpu


from conftest import PRECISIONS  # noqa: E402


@pytest.fixture(scope="module", params=PRECISIONS)
def tiny(cuda, request):
    """the tiny-geometry tower in BOTH arithmetic modes"""
    sd = weights.clip_state_dict("tiny", 0)
    clip = CLIPViTFM("tiny", state_dict=sd, device=cuda, precision=request.param)
    return sd, clip


@pytest.fixture(scope="module")
def b16(cuda):
    sd = weights.clip_state_dict("ViT-B/16", 0)
    return sd, CLIPViTFM("ViT-B/16", state_dict=sd, device=cuda)


@pytest.mark.parametrize("kw", [dict(), dict(gem_depth=1), dict(gem_depth=3), dict(gem_depth=13), dict(ss_attn_iter=0),
                                dict(ss_attn_iter=2), dict(ss_attn_temp=3.0)])
@pytest.mark.parametrize("R", [128, 64])
def test_tiny_image_features_vs_oracle(cuda, tiny, R, kw):
    """both residual streams, every token, at the checkpoint's own grid (R=64) and an interpolated one (R=128)"""
    sd, clip = tiny
    img = np.random.default_rng(R).standard_normal((3, R, R)).astype(np.float32)
    gm = G.create_gem_model("tiny", clip=clip, **kw)
    t = torch.from_numpy(img).to(cuda)
    gem = gm.image_features(t).cpu().numpy()
    ori = gm.image_features(t, return_ori=True).cpu().numpy()
    rg, ro = GO.gem_vit_forward(sd, img[None], **kw)
    np.testing.assert_allclose(ori, ro[0], rtol=0, atol=5e-5)
    np.testing.assert_allclose(gem, rg[0], rtol=0, atol=5e-5)


def test_b16_448_heatmap_vs_oracle(cuda, b16):
    """the reference's configuration: ViT-B/16 at 448 x 448 (785 tokens, the last 6 blocks GEM), 3 phrases"""
    sd, clip = b16
    img = np.random.default_rng(11).standard_normal((3, 448, 448)).astype(np.float32)
    txt = np.random.default_rng(12).standard_normal((3, 512)).astype(np.float32)
    gm = G.create_gem_model("ViT-B/16", clip=clip)
    feat = gm.image_features(torch.from_numpy(img).to(cuda))
    rg, _ = GO.gem_vit_forward(sd, img[None])
    np.testing.assert_allclose(feat.cpu().numpy(), rg[0], rtol=0, atol=2e-4)
    heat = gm.heatmap(feat, torch.from_numpy(txt).to(cuda), 448).cpu().numpy()
    ref = GO.gem_heatmap(rg[0], txt, 448)
    assert heat.shape == (3, 448, 448)
    np.testing.assert_allclose(heat, ref, rtol=0, atol=1e-3)       # north_star bar for similarity-derived scores
    assert heat.min() == 0.0 and heat.max() == 1.0
    # the heat-map kernels alone (same features on both sides): rounding only
    mine = gm.heatmap(torch.from_numpy(rg[0]).to(cuda), torch.from_numpy(txt).to(cuda), 448).cpu().numpy()
    np.testing.assert_allclose(mine, ref, rtol=0, atol=2e-5)
    raw = gm.heatmap(torch.from_numpy(rg[0]).to(cuda), torch.from_numpy(txt).to(cuda), 448, normalize=False).cpu().numpy()
    np.testing.assert_allclose(raw, GO.gem_heatmap(rg[0], txt, 448, normalize=False), rtol=0, atol=1e-4)
    # determinism
    assert torch.equal(gm.image_features(torch.from_numpy(img).to(cuda)), feat)


def test_b32_geometry_vs_oracle(cuda):
    """another checkpoint geometry: ViT-B/32 at 448 x 448 (7x7 position grid interpolated to 14x14, 197 tokens)"""
    sd = weights.clip_state_dict("ViT-B/32", 0)
    gm = G.create_gem_model("ViT-B/32", state_dict=sd, device=cuda)
    img = np.random.default_rng(31).standard_normal((3, 448, 448)).astype(np.float32)
    feat = gm.image_features(torch.from_numpy(img).to(cuda)).cpu().numpy()
    rg, _ = GO.gem_vit_forward(sd, img[None])
    assert feat.shape == (197, 512)
    np.testing.assert_allclose(feat, rg[0], rtol=0, atol=2e-4)
    del gm
    torch.cuda.empty_cache()


def test_l14_geometry_vs_oracle(cuda):
    """the extension geometry BASELINE.json names: ViT-L/14 (width 1024, 24 layers, 16 heads, patch 14 -> the
    patch-embedding GEMM stays fp32: K = 588), at 280 x 280 (16x16 position grid interpolated to 20x20, 401 tokens)"""
    sd = weights.clip_state_dict("ViT-L/14", 0)
    gm = G.create_gem_model("ViT-L/14", state_dict=sd, device=cuda)
    img = np.random.default_rng(41).standard_normal((3, 280, 280)).astype(np.float32)
    feat = gm.image_features(torch.from_numpy(img).to(cuda)).cpu().numpy()
    rg, _ = GO.gem_vit_forward(sd, img[None])
    assert feat.shape == (401, 768)
    np.testing.assert_allclose(feat, rg[0], rtol=0, atol=3e-4)
    del gm
    torch.cuda.empty_cache()


@pytest.mark.parametrize("h,w,H,W", [(448, 448, 480, 640), (448, 448, 300, 400), (448, 448, 640, 427), (64, 64, 37, 91),
                                     (448, 448, 448, 448), (32, 32, 5, 200)])
def test_resize_antialias_vs_torch(cuda, h, w, H, W):
    """hgl_resize_bilinear_aa == T.Resize((H, W), antialias=True) (Hybridgl_main.py:201): torch's own CPU kernel
    is the arithmetic the reference calls"""
    x = np.random.default_rng(h + W).standard_normal((2, h, w)).astype(np.float32)
    got = G.resize_antialias(torch.from_numpy(x).to(cuda), (H, W)).cpu().numpy()
    ref = F.interpolate(torch.from_numpy(x)[None], size=(H, W), mode="bilinear", antialias=True, align_corners=False)[0].numpy()
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)
    np.testing.assert_allclose(got, GO.resize_bilinear_aa(x, H, W), rtol=0, atol=2e-6)


def test_wrapper_call_surface(cuda, golden_dir):
    """gem_model(tensor_img[1,3,R,R], [phrases]) -> [1, T, R, R] as Hybridgl_main.py:200 calls it, with the BPE
    tokenizer and the text tower in the loop ("a photo of a {phrase}.")"""
    from hybridgl_amd.tokenizer import SimpleTokenizer, tokenize
    tk = SimpleTokenizer(os.path.join(golden_dir, "tiny_bpe_vocab.txt.gz"))
    sd = weights.clip_state_dict("tiny", 0)
    sd["token_embedding.weight"] = weights._draw(0, "token_embedding.weight", (len(tk.encoder), 64), 0.02)
    gm = G.create_gem_model("tiny", state_dict=sd, device=cuda, tokenizer=tk)
    img = np.random.default_rng(21).standard_normal((2, 3, 128, 128)).astype(np.float32)
    phrases = ["cat on left", "the dog"]
    out = gm(torch.from_numpy(img).to(cuda), phrases)
    assert out.shape == (2, 2, 128, 128)
    tok = tokenize(GO.gem_prompts(phrases), 16, tokenizer=tk)
    txt = O.encode_text(sd, tok, heads=1)
    rg, ro = GO.gem_vit_forward(sd, img)
    for b in range(2):
        np.testing.assert_allclose(out[b].cpu().numpy(), GO.gem_heatmap(rg[b], txt, 128), rtol=0, atol=1e-3)
    ori = gm(torch.from_numpy(img[:1]).to(cuda), phrases, normalize=False, return_ori=True)[0].cpu().numpy()
    np.testing.assert_allclose(ori, GO.gem_heatmap(ro[0], txt, 128, normalize=False), rtol=0, atol=2e-3)
    lst = gm.batched_forward(torch.from_numpy(img).to(cuda), [["cat on left"], ["the dog", "cat on left"]])
    assert [tuple(t.shape) for t in lst] == [(1, 128, 128), (2, 128, 128)]
    assert torch.equal(lst[0][0], out[0, 0])
    with pytest.raises(Exception):
        gm(torch.from_numpy(img), phrases)          # host tensor: no CPU path


def test_pipeline_computes_heatmaps_like_the_reference_sequence(cuda, b16):
    """HybridGLPipeline with a gem model == the reference's per-sentence sequence (Hybridgl_main.py:200-202)
    gem_model(tensor_img, [phrase])[0] -> T.Resize((h, w), antialias=True) -> coherence scoring, and the
    per-image cache of the GEM image features does not change anything."""
    import dataclasses
    from hybridgl_amd.pipeline import HybridGLPipeline, synthetic_ref
    sd, clip = b16
    gm = G.create_gem_model("ViT-B/16", clip=clip)
    ref, host = synthetic_ref(0, cuda, N=8, H=480, W=640, gem=True)
    assert ref.tensor_img.shape == (3, 448, 448) and ref.tokens.shape[0] == 12
    p1 = HybridGLPipeline(clip, "G2L", 9, gem_model=gm)
    out1 = p1.step(ref)
    # the same ref with the heat-maps produced up front through the package-style call surface
    text = clip.model.encode_text(ref.tokens, seq_len=ref.token_len)     # the pipeline computes the EOT prefix only
    sents = []
    for s in ref.sentences:
        m = gm.heatmap(gm.image_features(ref.tensor_img), text[s.gem_row:s.gem_row + 1], 448)      # gem_model(...)[0]
        a = G.resize_antialias(m, (480, 640))[0]
        sents.append(dataclasses.replace(s, imgattn=a, gem_row=None))
    p2 = HybridGLPipeline(clip, "G2L", 9)
    out2 = p2.step(dataclasses.replace(ref, sentences=sents))
    assert p1.metrics()["cum"] == p2.metrics()["cum"]
    assert torch.equal(out1[2][0], out2[2][0]) and torch.equal(out1[2][3], out2[2][3])
    # oracle heat-map of the last sentence -> the same coherence scores within the similarity tolerance
    rg, _ = GO.gem_vit_forward(sd, host["tensor_img"][None])
    rt = O.encode_text(sd, host["tokens"][11:12], heads=8)
    heat = GO.resize_bilinear_aa(GO.gem_heatmap(rg[0], rt, 448), 480, 640)[0]
    np.testing.assert_allclose(sents[-1].imgattn.cpu().numpy(), heat, rtol=0, atol=1e-3)
    # cached image features (same image id): identical results
    p3 = HybridGLPipeline(clip, "G2L", 9, gem_model=gm)
    p3.step(dataclasses.replace(ref, image_id=5)); p3.step(dataclasses.replace(ref, image_id=5))
    assert p3.metrics()["cum"] == [2 * v for v in p1.metrics()["cum"]]
    with pytest.raises(ValueError):
        HybridGLPipeline(clip, "G2L", 9).step(ref)        # no gem model and no heat-map


@pytest.mark.parametrize("kw", [dict(), dict(ss_attn_temp=3.0), dict(ss_attn_iter=2)])
def test_image_features_batch_equals_single_images(cuda, tiny, b16, kw):
    """hgl_gem_image_features_batch: token rows of several images stacked (one temperature per image, the final
    assignment to v per set over the images) == the images one by one; both residual streams."""
    for (sd, clip), name, R, tol in [(tiny, "tiny", 128, 1e-5), (b16, "ViT-B/16", 224, 1e-4)]:
        gm = G.create_gem_model(name, clip=clip, **kw)
        imgs = torch.from_numpy(np.random.default_rng(R).standard_normal((3, 3, R, R)).astype(np.float32)).to(cuda)
        imgs[1] *= 3.0                                    # different token norms -> different temperatures
        fb = gm.image_features_batch(imgs)
        ob = gm.image_features_batch(imgs, return_ori=True)
        for b in range(3):
            np.testing.assert_allclose(fb[b].cpu().numpy(), gm.image_features(imgs[b]).cpu().numpy(), rtol=0, atol=tol)
            np.testing.assert_allclose(ob[b].cpu().numpy(), gm.image_features(imgs[b], return_ori=True).cpu().numpy(), rtol=0, atol=tol)


def test_grouped_run_equals_per_ref_steps(cuda, b16):
    """HybridGLPipeline.run on given proposals: one text-encoder batch and one hybrid forward over the masks of several refs ==
    the refs stepped one by one (every mask row and every string is independent)."""
    from hybridgl_amd.pipeline import HybridGLPipeline, synthetic_ref
    _, clip = b16
    gm = G.create_gem_model("ViT-B/16", clip=clip)
    # ragged group: the refs bring different numbers of proposals
    refs = [synthetic_ref(i, cuda, N=n, H=320, W=480, gem=True, device_blur=True)[0] for i, n in enumerate((8, 5, 11))]
    p1 = HybridGLPipeline(clip, "G2L", 9, gem_model=gm)
    outs1 = [p1.step(r) for r in refs]
    p2 = HybridGLPipeline(clip, "G2L", 9, gem_model=gm)
    assert p2.run(iter(refs), group=len(refs), collect=True) == len(refs)
    outs2 = p2.collected
    assert p1.metrics()["cum"] == p2.metrics()["cum"] and p1.metrics()["n_sentences"] == p2.metrics()["n_sentences"] == 9
    for a, b in zip(outs1, outs2):
        np.testing.assert_allclose(a[0].cpu().numpy(), b[0].cpu().numpy(), rtol=0, atol=2e-5)      # hybrid features
        np.testing.assert_allclose(a[1].cpu().numpy(), b[1].cpu().numpy(), rtol=0, atol=2e-5)      # text features
        assert torch.equal(a[2][0], b[2][0])                                                      # winning indices


def test_errors(cuda, tiny):
    import ctypes as C
    from hybridgl_amd import _lib
    lib = _lib.load()
    _, clip = tiny
    v = clip.model.visual_w
    assert lib.hgl_gem_workspace_bytes(C.byref(v)) > 0
    x = torch.zeros((3, 64, 64), device=cuda)
    out = torch.zeros((17, 32), device=cuda)
    assert lib.hgl_gem_image_features(C.byref(v), x.data_ptr(), 99, 1, 0.0, out.data_ptr(), None, None, 0, None) != 0
    assert b"gem_blocks" in lib.hgl_last_error()
    assert lib.hgl_gem_image_features(C.byref(v), x.data_ptr(), 6, 1, 0.0, out.data_ptr(), None, None, 0, None) != 0
    assert b"workspace" in lib.hgl_last_error()
    assert lib.hgl_gem_heatmap(out.data_ptr(), 4, 32, out.data_ptr(), 1, 64, 1, x.data_ptr(), None, 0, None) != 0
