"""CPU: the numpy SAM oracle against stage tensors captured from the reference's own
segment_anything package at the tiny geometry (oracle/gen_golden.py:gen_sam_tiny)."""
import os

import numpy as np
import pytest

from hybridgl_amd import weights
from oracle import sam_oracle as S
from oracle.cases import sam_tiny_case


@pytest.fixture(scope="module")
def g(golden_dir):
    path = os.path.join(golden_dir, "sam_tiny.npz")
    if not os.path.exists(path):
        pytest.skip("sam_tiny.npz not generated")
    return np.load(path)


@pytest.fixture(scope="module")
def sd():
    return weights.sam_state_dict("tiny", 0)


@pytest.fixture(scope="module")
def emb(sd):
    c = sam_tiny_case()
    x = S.preprocess(c["resized"], 256)
    return S.image_encoder(sd, x, weights.SAM_CONFIGS["tiny"])


def test_image_encoder(g, emb):
    np.testing.assert_allclose(emb[::2, ::2], g["emb_nhwc"], rtol=0, atol=5e-5)


def test_encoder_blocks(g, sd):
    cfg = weights.SAM_CONFIGS["tiny"]
    c = sam_tiny_case()
    x = S.preprocess(c["resized"], 256)
    w = sd["image_encoder.patch_embed.proj.weight"]
    cols = x.reshape(3, 16, 16, 16, 16).transpose(1, 3, 0, 2, 4).reshape(256, 768)
    t = (cols @ w.reshape(160, -1).T + sd["image_encoder.patch_embed.proj.bias"]).reshape(1, 16, 16, 160)
    t = t + sd["image_encoder.pos_embed"]
    t0 = S.encoder_block(t.astype(np.float32), sd, "image_encoder.blocks.0", cfg["num_heads"], 14)
    t1 = S.encoder_block(t0, sd, "image_encoder.blocks.1", cfg["num_heads"], 0)
    np.testing.assert_allclose(t0[0][::3, ::3], g["blk0"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(t1[0][::3, ::3], g["blk1"], rtol=0, atol=3e-5)


def test_prompt_encoder(g, sd):
    c = sam_tiny_case()
    sp = S.embed_points(sd, c["points_in"], 256)
    np.testing.assert_allclose(sp, g["sparse"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(S.dense_pe(sd, 16, 16)[::5], g["dense_pe"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(S.point_grid(8), g["point_grid8"], rtol=0, atol=0)
    assert S.preprocess_shape(160, 200, 256) == c["input_size"]
    assert S.preprocess_shape(640, 640) == (1024, 1024) and S.preprocess_shape(427, 640) == (683, 1024)


def test_mask_decoder_and_postprocess(g, sd, emb):
    c = sam_tiny_case()
    sp = S.embed_points(sd, c["points_in"], 256)
    low, iou = S.mask_decoder(sd, emb, sp)
    np.testing.assert_allclose(low, g["low_res"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(iou, g["iou"], rtol=0, atol=5e-5)
    full = S.postprocess_masks(g["low_res"], c["input_size"], c["orig_size"], 256)
    np.testing.assert_allclose(full[:, :, ::4, ::4], g["full_logits"], rtol=0, atol=1e-5)
    flat = full.reshape(-1, *full.shape[-2:])
    stab, _, _ = S.stability_score(flat)
    np.testing.assert_allclose(stab, g["stability"], rtol=0, atol=2e-3)   # a logit at +-1 may flip one pixel
    boxes = S.mask_to_box(flat > 0)
    assert np.abs(boxes - g["boxes"]).max() <= 1


def test_amg_generate(g, sd):
    """whole generator: 4x4 grid, thresholds relaxed, CC clean-up min_area=20, NMS disabled."""
    if int(g["amg_n"][0]) == 0:
        pytest.skip("reference produced no masks")
    c = sam_tiny_case()
    cfg = weights.SAM_CONFIGS["tiny"]
    emb = S.image_encoder(sd, S.preprocess(c["resized"], 256), cfg)
    pts = S.point_grid(4) * np.array([[200, 160]])
    pin = pts.copy()
    pin[:, 0] *= c["input_size"][1] / 200
    pin[:, 1] *= c["input_size"][0] / 160
    low, iou = S.mask_decoder(sd, emb, S.embed_points(sd, pin, 256))
    full = S.postprocess_masks(low, c["input_size"], c["orig_size"], 256)
    K = full.shape[0] * 3
    idx, masks, boxes, stab = S.amg_filter(full.reshape(K, 160, 200), iou.reshape(K), -1e9, 0.0, 1.5, 20)
    ref_masks = np.unpackbits(g["amg_masks"], axis=-1)[..., :200].astype(bool)
    assert len(idx) == int(g["amg_n"][0])
    np.testing.assert_allclose(iou.reshape(K)[idx], g["amg_iou"], rtol=0, atol=5e-5)   # same candidates, same order
    mism = (masks != ref_masks).reshape(len(idx), -1).mean(1)
    assert mism.max() < 2e-3, mism.max()          # logits within 1e-4 of zero may flip isolated pixels
    xywh = boxes.copy(); xywh[:, 2] -= xywh[:, 0]; xywh[:, 3] -= xywh[:, 1]
    assert np.abs(xywh - g["amg_bbox"]).max() <= 2
    np.testing.assert_allclose(pts[idx // 3], g["amg_points"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("shape", [(160, 200, 205, 256), (427, 640, 683, 1024), (300, 200, 256, 171), (1500, 2000, 768, 1024)])
def test_pil_bilinear_resize_bit_exact(shape):
    """the restated Pillow resampler against Pillow itself (what torchvision's resize(to_pil_image(.)) calls)"""
    from PIL import Image
    H, W, oh, ow = shape
    img = np.random.default_rng(H).integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    ref = np.array(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(S.pil_bilinear_resize(img, oh, ow), ref)


def test_prompt_kinds_vs_reference_predictor(sd, emb):
    """labelled points, boxes and multimask_output False (predictor.py:169-243, prompt_encoder.py:73-101,
    mask_decoder.py:99-105): the oracle against tests/golden/sam_prompts.npz (reference SamPredictor outputs)"""
    from oracle.cases import sam_prompts_case
    gp = np.load(os.path.join(os.path.dirname(__file__), "golden", "sam_prompts.npz"))
    c, q = sam_tiny_case(), sam_prompts_case()
    sx, sy = c["input_size"][1] / 200, c["input_size"][0] / 160
    pts = q["points"] * np.array([sx, sy])
    co = np.stack([pts, np.zeros_like(pts)], 1)
    lab = np.stack([q["labels"], np.full_like(q["labels"], -1)], 1)
    bx = (q["boxes"].reshape(-1, 2, 2) * np.array([sx, sy]))
    np.testing.assert_allclose(bx.reshape(-1, 4), gp["boxes_in"], rtol=0, atol=1e-12)
    sp_box = S.embed_prompts(sd, bx, np.tile([2, 3], (4, 1)), 256)
    np.testing.assert_allclose(sp_box, gp["box_sparse"], rtol=0, atol=2e-5)
    for tag, sp in (("pts", S.embed_prompts(sd, co, lab, 256)), ("box", sp_box)):
        for mm in (True, False):
            low, iou = S.mask_decoder(sd, emb, sp, multimask=mm)
            k = f"{tag}_{'multi' if mm else 'single'}"
            np.testing.assert_allclose(low[:, :, ::2, ::2], gp[k + "_low"], rtol=0, atol=2e-4)
            np.testing.assert_allclose(iou, gp[k + "_iou"], rtol=0, atol=5e-5)
    # predict(): float32 coordinates (predictor.py:141-150)
    one = (q["one_point"] * np.array([sx, sy])).astype(np.float32)
    sp = S.embed_prompts(sd, np.stack([one, np.zeros_like(one)], 1), np.array([[0, -1]]), 256)
    low, iou = S.mask_decoder(sd, emb, sp)
    np.testing.assert_allclose(low[0], gp["predict_pt_low"], rtol=0, atol=2e-4)
    ob = (q["one_box"].reshape(1, 2, 2) * np.array([sx, sy])).astype(np.float32)
    low, iou = S.mask_decoder(sd, emb, S.embed_prompts(sd, ob, np.array([[2, 3]]), 256), multimask=False)
    np.testing.assert_allclose(low[0], gp["predict_box_low"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(iou[0], gp["predict_box_iou"], rtol=0, atol=5e-5)


def test_three_token_prompts_and_mask_inputs_vs_reference_predictor(sd, emb):
    """two points, point + box, mask inputs (prompt_encoder.py:73-127, predictor.py:169-243): the oracle against the
    reference SamPredictor's outputs (tests/golden/sam_prompts.npz)"""
    from oracle.cases import sam_prompts_case
    gp = np.load(os.path.join(os.path.dirname(__file__), "golden", "sam_prompts.npz"))
    c, q = sam_tiny_case(), sam_prompts_case()
    sc = np.array([c["input_size"][1] / 200, c["input_size"][0] / 160])
    prs = q["pairs"] * sc
    co = np.concatenate([prs, np.zeros((4, 1, 2))], 1)
    lab = np.concatenate([q["pair_labels"], np.full((4, 1), -1)], 1)
    low, iou = S.mask_decoder(sd, emb, S.embed_prompts(sd, co, lab, 256))
    np.testing.assert_allclose(low[:, :, ::2, ::2], gp["pair_low"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(iou, gp["pair_iou"], rtol=0, atol=5e-5)
    bx = q["boxes"].reshape(-1, 2, 2) * sc
    co = np.concatenate([(q["points"] * sc)[:, None, :], bx], 1)
    lab = np.concatenate([q["labels"][:, None], np.tile([2, 3], (4, 1))], 1)
    low, iou = S.mask_decoder(sd, emb, S.embed_prompts(sd, co, lab, 256), multimask=False)
    np.testing.assert_allclose(low[:, :, ::2, ::2], gp["ptbox_low"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(iou, gp["ptbox_iou"], rtol=0, atol=5e-5)
    mny = q["many"] * sc
    co = np.concatenate([mny, np.zeros((2, 1, 2))], 1)
    lab = np.concatenate([q["many_labels"], np.full((2, 1), -1)], 1)
    low, iou = S.mask_decoder(sd, emb, S.embed_prompts(sd, co, lab, 256))
    np.testing.assert_allclose(low[:, :, ::2, ::2], gp["many_low"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(iou, gp["many_iou"], rtol=0, atol=5e-5)
    co = np.concatenate([mny[:, :3], bx[:2]], 1)
    lab = np.concatenate([q["many_labels"][:, :3], np.tile([2, 3], (2, 1))], 1)
    low, iou = S.mask_decoder(sd, emb, S.embed_prompts(sd, co, lab, 256), multimask=False)
    np.testing.assert_allclose(low[:, :, ::2, ::2], gp["manybox_low"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(iou, gp["manybox_iou"], rtol=0, atol=5e-5)
    dense = S.embed_masks(sd, gp["mask_in"])
    np.testing.assert_allclose(dense[:, ::7], gp["mask_dense"], rtol=0, atol=2e-5)
    low, iou = S.mask_decoder(sd, emb, S.embed_prompts(sd, bx, np.tile([2, 3], (4, 1)), 256), dense=dense)
    np.testing.assert_allclose(low[:, :, ::2, ::2], gp["maskin_low"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(iou, gp["maskin_iou"], rtol=0, atol=1e-4)


def test_nms_order_with_nan_scores_is_torch_sort_descending():
    """the oracle's NMS takes candidates in the order torch.sort(descending=True) gives (torchvision.ops.nms): NaN first"""
    import torch
    from oracle import sam_oracle as S
    s = np.array([0.3, np.nan, 0.9, 0.3, np.nan, np.inf, -np.inf, 0.1], np.float32)
    far = np.array([[100 * i, 0, 100 * i + 10, 10] for i in range(len(s))], np.int64)      # nothing overlaps: kept = order
    assert S.nms(far, s, 0.5).tolist() == torch.argsort(torch.from_numpy(s), descending=True, stable=True).tolist()
