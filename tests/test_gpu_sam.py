"""GPU parity of the SAM path (image encoder, prompt encoder + mask decoder, fused post-processing,
NMS, whole SamAutomaticMaskGenerator) against reference stage tensors and the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from hybridgl_amd import sam as hsam
from hybridgl_amd import weights
from oracle import sam_oracle as S
from oracle.cases import sam_tiny_case

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "sam_tiny.npz"))


from conftest import PRECISIONS  # noqa: E402


@pytest.fixture(scope="module", params=PRECISIONS)
def tiny(cuda, request):
    """the tiny-geometry SAM in BOTH arithmetic modes: every test below that takes it runs twice"""
    sd = weights.sam_state_dict("tiny", 0)
    return sd, hsam.Sam(sd, weights.SAM_CONFIGS["tiny"], cuda, precision=request.param)


@pytest.fixture(scope="module")
def tiny_f32(cuda):
    """the exact fp32 matrix-core path of the same model (the decoder / encoder have separate code for the two modes)"""
    from hybridgl_amd import ops
    sd = weights.sam_state_dict("tiny", 0)
    m = hsam.Sam(sd, weights.SAM_CONFIGS["tiny"], cuda, precision="f32")
    yield sd, m
    ops.set_precision(ops.default_precision())


def _p01(points_in, img_size):
    return ((points_in + 0.5) / float(img_size)).astype(np.float32)


def test_tiny_encoder_vs_reference(cuda, g, tiny):
    c = sam_tiny_case()
    emb = tiny[1].encode(T(c["resized"], cuda)).cpu().numpy().reshape(16, 16, 256)
    np.testing.assert_allclose(emb[::2, ::2], g["emb_nhwc"], rtol=0, atol=1e-4)
    ref = S.image_encoder(tiny[0], S.preprocess(c["resized"], 256), weights.SAM_CONFIGS["tiny"])
    np.testing.assert_allclose(emb, ref, rtol=0, atol=1e-4)


def test_tiny_decoder_vs_reference(cuda, g, tiny):
    c = sam_tiny_case()
    sd, m = tiny
    emb = S.image_encoder(sd, S.preprocess(c["resized"], 256), weights.SAM_CONFIGS["tiny"])
    low, iou = m.decode_points(T(emb.reshape(256, 256), cuda), T(_p01(c["points_in"], 256), cuda))
    np.testing.assert_allclose(low.cpu().numpy(), g["low_res"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(iou.cpu().numpy(), g["iou"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(m.dense_pe.cpu().numpy()[::5], g["dense_pe"], rtol=0, atol=2e-5)


def test_tiny_decoder_and_encoder_fp32_mode(cuda, g, tiny_f32):
    """HYBRIDGL_PRECISION=f32: same goldens, same tolerances, through the fp32 MFMA kernels"""
    from hybridgl_amd import ops
    c = sam_tiny_case()
    sd, m = tiny_f32
    ops.set_precision("f32")
    try:
        emb = m.encode(T(c["resized"], cuda)).cpu().numpy().reshape(16, 16, 256)
        np.testing.assert_allclose(emb[::2, ::2], g["emb_nhwc"], rtol=0, atol=1e-4)
        ref = S.image_encoder(sd, S.preprocess(c["resized"], 256), weights.SAM_CONFIGS["tiny"])
        low, iou = m.decode_points(T(ref.reshape(256, 256), cuda), T(_p01(c["points_in"], 256), cuda))
        np.testing.assert_allclose(low.cpu().numpy(), g["low_res"], rtol=0, atol=3e-4)
        np.testing.assert_allclose(iou.cpu().numpy(), g["iou"], rtol=0, atol=1e-4)
    finally:
        ops.set_precision(ops.default_precision())


def test_predictor_mirror_vs_reference(cuda, g, tiny):
    """SamPredictor.set_image / predict_torch (predictor.py:17-269) on the tiny model against the reference's stage
    tensors: full-resolution logits, IoU predictions and the error behaviour."""
    c = sam_tiny_case()
    pred = hsam.SamPredictor(tiny[1])
    with pytest.raises(RuntimeError):
        pred.predict_torch(torch.zeros(1, 1, 2), torch.ones(1, 1))
    pred.set_image(c["image"])
    assert pred.original_size == (160, 200) and pred.input_size == c["input_size"]
    pts = pred.transform.apply_coords(c["points"], pred.original_size)
    np.testing.assert_allclose(pts, c["points_in"], rtol=0, atol=1e-12)
    logits, iou, low = pred.predict_torch(torch.from_numpy(pts)[:, None, :], torch.ones(len(pts), 1, dtype=torch.int),
                                          return_logits=True)
    np.testing.assert_allclose(iou.cpu().numpy(), g["iou"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(low.cpu().numpy(), g["low_res"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(logits.cpu().numpy()[:, :, ::4, ::4], g["full_logits"], rtol=0, atol=3e-4)
    masks, _, _ = pred.predict_torch(torch.from_numpy(pts)[:, None, :], torch.ones(len(pts), 1, dtype=torch.int))
    assert masks.dtype == torch.bool and tuple(masks.shape) == (5, 3, 160, 200)
    with pytest.raises(NotImplementedError):       # eleven points per prompt: twelve sparse tokens
        pred.predict_torch(torch.zeros(1, 11, 2), torch.ones(1, 11, dtype=torch.int))
    with pytest.raises(NotImplementedError):       # a mask input alone
        pred.predict_torch(None, None, mask_input=torch.zeros(1, 1, 64, 64))
    m1, i1, l1 = pred.predict(c["points"][:1], np.array([1]))
    assert m1.shape == (3, 160, 200)
    # predict() hands float32 coordinates to the prompt encoder (predictor.py:141-143), the call above float64
    assert (m1 != masks[0].cpu().numpy()).mean() < 1e-3


def test_predictor_prompt_kinds_vs_reference(cuda, tiny, golden_dir):
    """labelled single points, boxes, multimask_output False and predict() (predictor.py:90-243, prompt_encoder.py:73-101,
    mask_decoder.py:99-105) against the reference SamPredictor's outputs (tests/golden/sam_prompts.npz)"""
    from oracle.cases import sam_prompts_case
    gp = np.load(os.path.join(golden_dir, "sam_prompts.npz"))
    c, q = sam_tiny_case(), sam_prompts_case()
    pred = hsam.SamPredictor(tiny[1])
    pred.set_image(c["image"])
    pts = pred.transform.apply_coords(q["points"], pred.original_size)
    bxs = pred.transform.apply_boxes(q["boxes"], pred.original_size)
    np.testing.assert_allclose(bxs, gp["boxes_in"], rtol=0, atol=1e-12)
    calls = (("pts", dict(point_coords=torch.from_numpy(pts)[:, None, :], point_labels=torch.from_numpy(q["labels"])[:, None])),
             ("box", dict(point_coords=None, point_labels=None, boxes=torch.from_numpy(bxs))))
    for tag, kw in calls:
        for mm in (True, False):
            full, iou, low = pred.predict_torch(multimask_output=mm, return_logits=True, **kw)
            k = f"{tag}_{'multi' if mm else 'single'}"
            assert tuple(low.shape) == (4, 3 if mm else 1, 64, 64) and tuple(full.shape) == (4, 3 if mm else 1, 160, 200)
            np.testing.assert_allclose(iou.cpu().numpy(), gp[k + "_iou"], rtol=0, atol=1e-4)
            np.testing.assert_allclose(low.cpu().numpy()[:, :, ::2, ::2], gp[k + "_low"], rtol=0, atol=3e-4)
            np.testing.assert_allclose(full.cpu().numpy()[:, :, ::8, ::8], gp[k + "_full"], rtol=0, atol=3e-4)
    m, iou, low = pred.predict(point_coords=q["one_point"], point_labels=q["one_label"], multimask_output=True, return_logits=True)
    np.testing.assert_allclose(low, gp["predict_pt_low"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(iou, gp["predict_pt_iou"], rtol=0, atol=1e-4)
    m, iou, low = pred.predict(box=q["one_box"], multimask_output=False, return_logits=True)
    assert m.shape == (1, 160, 200) and low.shape == (1, 64, 64) and iou.shape == (1,)
    np.testing.assert_allclose(low, gp["predict_box_low"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(iou, gp["predict_box_iou"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(m[:, ::4, ::4], gp["predict_box_full"], rtol=0, atol=3e-4)
    mb, _, _ = pred.predict(box=q["one_box"], multimask_output=False)
    ref = np.unpackbits(gp["predict_box_mask"], axis=-1)[..., :200].astype(bool)
    assert mb.dtype == bool and (mb != ref).mean() < 2e-3     # logits within 3e-4 of zero may flip isolated pixels
    # three sparse tokens: two points (+ padding), a point and a box; mask inputs (per-prompt dense embeddings)
    prs = pred.transform.apply_coords(q["pairs"], pred.original_size)
    _, iou, low = pred.predict_torch(torch.from_numpy(prs), torch.from_numpy(q["pair_labels"]), return_logits=True)
    np.testing.assert_allclose(iou.cpu().numpy(), gp["pair_iou"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(low.cpu().numpy()[:, :, ::2, ::2], gp["pair_low"], rtol=0, atol=3e-4)
    _, iou, low = pred.predict_torch(torch.from_numpy(pts)[:, None, :], torch.from_numpy(q["labels"])[:, None],
                                     boxes=torch.from_numpy(bxs), multimask_output=False, return_logits=True)
    np.testing.assert_allclose(iou.cpu().numpy(), gp["ptbox_iou"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(low.cpu().numpy()[:, :, ::2, ::2], gp["ptbox_low"], rtol=0, atol=3e-4)
    # more tokens (the general attention kernels; the token -> image attention 7 queries at a time): six points, three + a box
    mny = pred.transform.apply_coords(q["many"], pred.original_size)
    _, iou, low = pred.predict_torch(torch.from_numpy(mny), torch.from_numpy(q["many_labels"]), return_logits=True)
    np.testing.assert_allclose(iou.cpu().numpy(), gp["many_iou"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(low.cpu().numpy()[:, :, ::2, ::2], gp["many_low"], rtol=0, atol=3e-4)
    _, iou, low = pred.predict_torch(torch.from_numpy(mny)[:, :3], torch.from_numpy(q["many_labels"])[:, :3],
                                     boxes=torch.from_numpy(bxs)[:2], multimask_output=False, return_logits=True)
    np.testing.assert_allclose(iou.cpu().numpy(), gp["manybox_iou"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(low.cpu().numpy()[:, :, ::2, ::2], gp["manybox_low"], rtol=0, atol=3e-4)
    dense = tiny[1].embed_masks(T(gp["mask_in"], cuda))
    np.testing.assert_allclose(dense.cpu().numpy()[:, ::7], gp["mask_dense"], rtol=0, atol=2e-5)
    full, iou, low = pred.predict_torch(None, None, boxes=torch.from_numpy(bxs), mask_input=torch.from_numpy(gp["mask_in"]),
                                        return_logits=True)
    np.testing.assert_allclose(iou.cpu().numpy(), gp["maskin_iou"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(low.cpu().numpy()[:, :, ::2, ::2], gp["maskin_low"], rtol=0, atol=5e-4)
    np.testing.assert_allclose(full.cpu().numpy()[:, :, ::8, ::8], gp["maskin_full"], rtol=0, atol=5e-4)
    _, _, low1 = pred.predict(box=q["one_box"], multimask_output=False, return_logits=True)
    _, iou, low = pred.predict(point_coords=q["one_point"], point_labels=np.array([1]), box=q["one_box"], mask_input=low1,
                               multimask_output=True, return_logits=True)
    np.testing.assert_allclose(iou, gp["predict_all_iou"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(low[:, ::2, ::2], gp["predict_all_low"], rtol=0, atol=5e-4)
    # the foreground-point fast path and the labelled path are the same decoder
    a = pred.predict_torch(torch.from_numpy(pts)[:, None, :], torch.ones(4, 1, dtype=torch.int), return_logits=True)
    lab = torch.stack([torch.ones(4, dtype=torch.int32), torch.full((4,), -1, dtype=torch.int32)], 1).to(cuda)
    p01 = pred._coords01(torch.from_numpy(pts))
    low2, iou2 = tiny[1].decode_prompts(pred.features, torch.stack([p01, torch.zeros_like(p01)], 1).contiguous(), lab)
    assert torch.equal(a[2], low2) and torch.equal(a[1], iou2)


@pytest.mark.parametrize("orig,inp,img,low", [((640, 640), (1024, 1024), 1024, 256), ((427, 640), (683, 1024), 1024, 256),
                                               ((160, 200), (205, 256), 256, 64), ((1500, 2000), (768, 1024), 1024, 256),
                                               ((97, 130), (764, 1024), 1024, 256),
                                               ((300, 400), (768, 1024), 1024, 256), ((283, 377), (769, 1024), 1024, 256),
                                               ((400, 300), (1024, 768), 1024, 256)])
def test_postprocess_shared_table_kernel_is_bit_identical(cuda, tiny, orig, inp, img, low):
    """sam_postprocess_sep_kernel (per-tile column tables + the horizontally interpolated patch in LDS) against the per-pixel
    kernel (HGL_SAM_POST_SEP=0): logits, masks, boxes and stability counters bit for bit, over down- and up-scaling size
    ratios (the fifth geometry exceeds the shared tables, and so do the last three -- crops of a crop layer at 2.6 : 1: both
    calls take the per-pixel kernel there.  A 32-column tile variant of the shared-table kernel fits those crops and was
    measured: 3.1 ms per crop against the per-pixel kernel's 2.1)"""
    m = tiny[1]
    rng = np.random.default_rng(orig[0])
    K = 7
    lr = T((rng.standard_normal((K, low, low)) * 3).astype(np.float32), cuda)
    iou = T(rng.random(K).astype(np.float32), cuda)
    old_img = m.img_size
    outs = []
    try:
        m.img_size = img
        for flag in ("1", "0"):
            os.environ["HGL_SAM_POST_SEP"] = flag
            outs.append(m.postprocess(lr, iou, inp, orig, -1e30, 0.1, 1.0, return_logits=True) +
                        m.postprocess(lr, iou, inp, orig, 0.5, 0.1, 1.0)[:4])     # with the IoU filter: no logits of the dropped
    finally:
        m.img_size = old_img
        os.environ.pop("HGL_SAM_POST_SEP", None)
    for x, y in zip(*outs):
        assert torch.equal(torch.nan_to_num(x), torch.nan_to_num(y)) if x.dtype == torch.float32 else torch.equal(x, y)
    assert outs[0][0].any()


def test_tiny_postprocess_vs_reference(cuda, g, tiny):
    c = sam_tiny_case()
    m = tiny[1]
    low = g["low_res"].reshape(15, 64, 64)
    iou = g["iou"].reshape(15)
    masks, boxes, stab, keep, full = m.postprocess(T(low, cuda), T(iou, cuda), c["input_size"], c["orig_size"],
                                                   -1e30, 0.0, 1.0, return_logits=True)
    full = full.cpu().numpy()
    np.testing.assert_allclose(full[:, ::4, ::4], g["full_logits"].reshape(15, 40, 50), rtol=0, atol=2e-5)
    ref_full = S.postprocess_masks(g["low_res"], c["input_size"], c["orig_size"], 256).reshape(15, 160, 200)
    np.testing.assert_allclose(full, ref_full, rtol=0, atol=2e-5)
    # masks / counters are exact functions of the logits this kernel produced
    assert np.array_equal(masks.cpu().numpy().astype(bool), full > 0)
    st, _, _ = S.stability_score(full)
    np.testing.assert_allclose(stab.cpu().numpy(), st, rtol=0, atol=1e-6)
    assert np.array_equal(boxes.cpu().numpy().astype(np.int64), S.mask_to_box(full > 0))
    np.testing.assert_allclose(stab.cpu().numpy(), g["stability"], rtol=0, atol=2e-3)
    assert np.abs(boxes.cpu().numpy() - g["boxes"]).max() <= 1
    assert keep.cpu().numpy().all()


def test_postprocess_filters_and_empty(cuda, tiny):
    """iou filter skips candidates (zero mask, keep=0); empty masks give box 0 and NaN stability -> keep=0."""
    m = tiny[1]
    rng = np.random.default_rng(1)
    low = rng.standard_normal((6, 64, 64)).astype(np.float32) * 3
    low[2] = -5.0   # never above threshold -> empty mask, 0/0 stability
    iou = np.array([0.9, 0.5, 0.9, 0.95, 0.71, 0.7], np.float32)
    masks, boxes, stab, keep, _ = m.postprocess(T(low, cuda), T(iou, cuda), (205, 256), (160, 200), 0.7, 0.1, 1.0)
    keep, masks, boxes = keep.cpu().numpy(), masks.cpu().numpy(), boxes.cpu().numpy()
    assert keep.tolist() == [1, 0, 0, 1, 1, 0]
    assert masks[1].sum() == 0 and masks[5].sum() == 0 and masks[2].sum() == 0
    assert boxes[2].tolist() == [0, 0, 0, 0]
    full = S.postprocess_masks(low[None], (205, 256), (160, 200), 256)[0]
    assert np.array_equal(masks[0].astype(bool), full[0] > 0)


@pytest.mark.parametrize("orig,inp", [((192, 256), (192, 256)), ((96, 128), (192, 256)), ((333, 500), (171, 256))])
def test_postprocess_tile_paths_vs_oracle(cuda, tiny, orig, inp):
    """16-byte aligned rows (W % 16 == 0), ragged rows and up/down-scaling ratios against the oracle's two-stage
    bilinear: logits within 2e-5, masks / boxes / stability exact functions of the produced logits."""
    m = tiny[1]
    rng = np.random.default_rng(orig[0])
    low = rng.standard_normal((5, 64, 64)).astype(np.float32) * 2
    iou = np.full(5, 0.9, np.float32)
    masks, boxes, stab, keep, full = m.postprocess(T(low, cuda), T(iou, cuda), inp, orig, 0.5, 0.0, 1.0,
                                                   return_logits=True)
    full = full.cpu().numpy()
    ref = S.postprocess_masks(low[None], inp, orig, 256)[0]
    np.testing.assert_allclose(full, ref, rtol=0, atol=2e-5)
    assert np.array_equal(masks.cpu().numpy().astype(bool), full > 0)
    assert np.array_equal(boxes.cpu().numpy().astype(np.int64), S.mask_to_box(full > 0))
    st, _, _ = S.stability_score(full)
    np.testing.assert_allclose(stab.cpu().numpy(), st, rtol=0, atol=1e-6)


def test_nms_vs_oracle(cuda):
    rng = np.random.default_rng(2)
    for K in [1, 7, 64, 192, 512, 513, 700]:      # <= 512: the bit-matrix kernel, above: the serial one
        xy = rng.integers(0, 500, size=(K, 2))
        wh = rng.integers(1, 200, size=(K, 2))
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.int32)
        boxes[K // 2] = boxes[0]                     # duplicates
        scores = rng.random(K).astype(np.float32)
        if K > 3:
            scores[3] = scores[1]                    # score tie -> lower index first
        keep = (rng.random(K) > 0.2).astype(np.uint8)
        keep[0] = 1
        idx, n = hsam.nms(T(boxes, cuda), T(scores, cuda), T(keep, cuda), 0.7)
        n = int(n.item())
        got = idx.cpu().numpy()[:n]
        sel = np.nonzero(keep)[0]
        ref = sel[S.nms(boxes[sel].astype(np.int64), scores[sel], 0.7)]
        assert got.tolist() == ref.tolist()


def test_nms_with_nan_scores(cuda):
    """Several NaN scores among the valid candidates (an f16x3 overflow with the filters open): every kernel ranks them as
    torch.sort(descending=True) does -- first, by index -- so the ranks never collide, no index is negative and the kept
    list is the oracle's.  K = 192 / 512: bit-matrix kernel; 700: serial kernel; 1500: the any-K path."""
    rng = np.random.default_rng(12)
    for K in [5, 192, 512, 700, 1500]:
        xy = rng.integers(0, 700, size=(K, 2))
        wh = rng.integers(1, 200, size=(K, 2))
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.int32)
        scores = rng.random(K).astype(np.float32)
        scores[rng.choice(K, size=min(K // 2, 7), replace=False)] = np.nan
        scores[0] = np.nan
        keep = (rng.random(K) > 0.2).astype(np.uint8)
        keep[0] = keep[K - 1] = 1
        sel = np.nonzero(keep)[0]
        ref = sel[S.nms(boxes[sel].astype(np.int64), scores[sel], 0.7)].tolist()
        assert ref[0] == 0                                   # the first NaN leads
        fn = hsam.nms if K <= 1024 else hsam.nms_large
        idx, n = fn(T(boxes, cuda), T(scores, cuda), T(keep, cuda), 0.7)
        got = idx.cpu().numpy()[: int(n.item())].tolist()
        assert min(got) >= 0 and got == ref, K
        if K <= 1024:
            idx2, n2 = hsam.nms_large(T(boxes, cuda), T(scores, cuda), T(keep, cuda), 0.7)
            assert idx2.cpu().numpy()[: int(n2.item())].tolist() == ref, K
    # keep = 1 everywhere, all scores NaN: index order
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [50, 50, 60, 60], [0, 0, 10, 10]], np.int32)
    idx, n = hsam.nms(T(boxes, cuda), T(np.full(4, np.nan, np.float32), cuda), T(np.ones(4, np.uint8), cuda), 0.5)
    assert idx.cpu().numpy()[: int(n.item())].tolist() == [0, 2]


def test_decoder_iou_gate_leaves_the_candidates_unchanged(cuda, tiny):
    """hgl_sam_decode_points_gated: the quality head first, prompts whose three predictions all fail pred_iou_thresh skip the
    output upscaling.  The generator with the threshold at the median prediction returns the SAME proposals with the gate on
    and off (masks, boxes, predictions, order); the gated decoder's iou_pred equals the plain call's bit for bit, its logits
    equal it on every prompt that passes, and about half the prompts are skipped."""
    c = sam_tiny_case()
    sd, m = tiny
    emb = m.encode(T(c["resized"], cuda))
    rng = np.random.default_rng(3)
    p01 = T(rng.random((97, 2)).astype(np.float32), cuda)
    low0, iou0 = m.decode_points(emb, p01)
    thr = float(iou0.max(dim=1).values.median())
    low1, iou1 = m.decode_points(emb, p01, iou_gate=thr)
    assert torch.equal(iou0, iou1)
    passing = (iou0 > thr).any(dim=1)
    assert 0.25 < float(passing.float().mean()) < 0.75
    assert torch.equal(low0[passing], low1[passing])
    kw = dict(points_per_side=6, pred_iou_thresh=thr, stability_score_thresh=0.0, crop_n_layers=0, min_mask_region_area=20,
              box_nms_thresh=0.7)
    gen = hsam.SamAutomaticMaskGenerator(m, **kw)
    a = gen.generate(c["image"])
    gen.iou_gate = False
    b = gen.generate(c["image"])
    assert len(a) == len(b) > 0
    for x, y in zip(a, b):
        assert np.array_equal(x["segmentation"], y["segmentation"]) and x["bbox"] == y["bbox"]
        assert x["predicted_iou"] == y["predicted_iou"] and x["point_coords"] == y["point_coords"]


def test_tiny_generate_vs_reference(cuda, g, tiny):
    c = sam_tiny_case()
    gen = hsam.SamAutomaticMaskGenerator(tiny[1], points_per_side=4, pred_iou_thresh=-1e9,
                                         stability_score_thresh=0.0, crop_n_layers=0, min_mask_region_area=20,
                                         box_nms_thresh=1.5)
    anns = gen.generate(c["image"])
    assert len(anns) == int(g["amg_n"][0])
    ref_masks = np.unpackbits(g["amg_masks"], axis=-1)[..., :200].astype(bool)
    np.testing.assert_allclose([a["predicted_iou"] for a in anns], g["amg_iou"], rtol=0, atol=1e-4)
    mism = np.array([(a["segmentation"] != r).mean() for a, r in zip(anns, ref_masks)])
    assert mism.max() < 2e-3, mism.max()
    np.testing.assert_allclose([a["point_coords"][0] for a in anns], g["amg_points"], rtol=0, atol=1e-9)
    assert np.abs(np.array([a["bbox"] for a in anns]) - g["amg_bbox"]).max() <= 2
    assert all(a["crop_box"] == [0, 0, 200, 160] for a in anns)


def _fusion(mask):
    from hybridgl_amd import _lib
    return _lib.load().hgl_sam_decoder_fusion(mask)


@pytest.mark.parametrize("name,P", [("tiny", 5), ("tiny", 64), ("vit_h_d2", 64), ("vit_h_d2", 7)])
def test_fused_decoder_stages_equal_the_unfused_launches(cuda, name, P):
    """hgl_sam_decoder_fusion: the merged image-side projections (one GEMM for k | v | q, positional encoding as a table),
    the fused image -> token step and the fused upscaling + hyper-network kernel against the launches they replace
    (ConvTranspose GEMM, LayerNorm2d + GELU, ConvTranspose GEMM + GELU, hyper-network products), on the tiny grid (16 x 16:
    a tile spans four grid rows) and the ViT-H grid (64 x 64): the same matrix products in the same order, the LayerNorm sums
    and the 32-channel dot products associated differently -- equal to fp32 rounding (1e-6 of the largest logit)."""
    from hybridgl_amd import ops
    if ops.default_precision() != "f16x3":
        pytest.skip("the fused stages belong to the split-fp16 mode")
    cfg = weights.SAM_CONFIGS[name]
    m = hsam.Sam(weights.sam_state_dict(name, 0), cfg, cuda)
    g = cfg["img_size"] // cfg["patch_size"]
    rng = np.random.default_rng(P)
    emb = T(rng.standard_normal((g * g, 256)).astype(np.float32), cuda)
    p01 = T(rng.random((P, 2)).astype(np.float32), cuda)
    old = _fusion(-1)
    try:
        _fusion(0)
        low0, iou0 = m.decode_points(emb, p01)
        _fusion(old if old > 0 else 0x7fffffff)
        low1, iou1 = m.decode_points(emb, p01)
    finally:
        _fusion(old)
    assert torch.isfinite(low0).all() and float(low0.abs().max()) > 0
    # bit 0 alone (the fused tail): same matrix products, sums associated differently
    _fusion(1)
    low2, iou2 = m.decode_points(emb, p01)
    _fusion(old)
    assert torch.equal(iou0, iou2)
    assert float((low0 - low2).abs().max()) <= 2e-6 * float(low0.abs().max()), float((low0 - low2).abs().max())
    # bit 2 on top of bits 0-1 (image -> token attention + out-projection + norm4 in one launch): the attention and the
    # product are those of the separate launches, the LayerNorm sums over 256 channels are associated differently
    _fusion(3)
    low3, iou3 = m.decode_points(emb, p01)
    _fusion(7)
    low7, iou7 = m.decode_points(emb, p01)
    _fusion(old)
    assert float((iou3 - iou7).abs().max()) <= 4e-6 * max(1.0, float(iou3.abs().max())), float((iou3 - iou7).abs().max())
    assert float((low3 - low7).abs().max()) <= 4e-6 * float(low3.abs().max()), float((low3 - low7).abs().max())
    # bit 4: the token -> image attention in key chunks (exponentials in base 2, partials merged in a second pass)
    _fusion(7 + 16)
    low23, iou23 = m.decode_points(emb, p01)
    _fusion(old)
    assert float((iou7 - iou23).abs().max()) <= 1e-5 * max(1.0, float(iou7.abs().max())), float((iou7 - iou23).abs().max())
    assert float((low7 - low23).abs().max()) <= 1e-5 * float(low7.abs().max()), float((low7 - low23).abs().max())
    # bit 5: the token -> image attention of layer 1 and the final one on the RAW image-token planes (the 7 tokens projected
    # through W_k / W_v instead of the image tokens: (q W_k)(keys + pe)^T and (P keys) W_v^T + b_v): the same quantities in real
    # arithmetic, every product the split-fp16 triple -- equal to the projected path to fp32 rounding
    _fusion(7 + 16 + 32)
    low55, iou55 = m.decode_points(emb, p01)
    _fusion(old)
    assert torch.isfinite(low55).all()
    assert float((iou23 - iou55).abs().max()) <= 2e-5 * max(1.0, float(iou23.abs().max())), float((iou23 - iou55).abs().max())
    assert float((low23 - low55).abs().max()) <= 2e-5 * float(low23.abs().max()), float((low23 - low55).abs().max())
    # all stages: the merged projections add the positional encoding AFTER the product ((keys + pe) W = keys W + pe W)
    assert float((iou0 - iou1).abs().max()) <= 2e-5 * max(1.0, float(iou0.abs().max()))
    assert float((low0 - low1).abs().max()) <= 2e-5 * float(low0.abs().max()), float((low0 - low1).abs().max())
    del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n", [576, 1100])
def test_decoder_large_prompt_batches_equal_small_ones(cuda, n):
    """points_per_batch is a memory knob (automatic_mask_generator.py:244-255): 576 / 1100 prompts in one decoder launch (the
    PhraseCut configuration of bench.py uses 1024: token-side GEMMs of 7168 rows on the small-tile kernel, one workgroup per
    prompt in the raw-token attention, a positional GEMM of 57k rows) give the rows of launches of 192, bit for bit, at the ViT-H
    decoder size.  Launches of at most 128 prompts cut the key range of the token -> image attention into eight pieces per
    prompt (one workgroup per prompt would leave most of the chip idle): their rows are the same sums in another order --
    equal to fp32 rounding, and bit for bit among themselves (64 of 64 == 64 of 128)."""
    name = "vit_h_d2"
    cfg = weights.SAM_CONFIGS[name]
    m = hsam.Sam(weights.sam_state_dict(name, 0), cfg, cuda)
    g = cfg["img_size"] // cfg["patch_size"]
    rng = np.random.default_rng(5)
    emb = T(rng.standard_normal((g * g, 256)).astype(np.float32), cuda)
    p01 = T(rng.random((n, 2)).astype(np.float32), cuda)
    low, iou = m.decode_points(emb, p01)
    for s in (0, 192, n - 192):
        l2, i2 = m.decode_points(emb, p01[s:s + 192].contiguous())
        assert torch.equal(l2, low[s:s + 192]) and torch.equal(i2, iou[s:s + 192]), s
    assert torch.isfinite(low).all()
    scale = float(low.abs().max())
    l64, i64 = m.decode_points(emb, p01[64:128].contiguous())
    assert float((l64 - low[64:128]).abs().max()) <= 2e-5 * max(1.0, scale) and float((i64 - iou[64:128]).abs().max()) <= 2e-5
    l128, i128 = m.decode_points(emb, p01[:128].contiguous())
    assert torch.equal(l128[64:], l64) and torch.equal(i128[64:], i64)
    del m, low
    torch.cuda.empty_cache()


def test_decoder_on_a_24x24_grid_vs_oracle(cuda):
    """576 image tokens: the last key chunk of the token -> image attention holds 64 keys, a 64-token tile of the fused
    image -> token step spans 2 2/3 grid rows, the fused tail does not apply (the four launches run); labelled prompts with
    three tokens take the unfused image -> token step.  Against the oracle (pinned at 16 x 16 by the reference)."""
    name = "tiny24"
    cfg = weights.SAM_CONFIGS[name]
    sd = weights.sam_state_dict(name, 3)
    m = hsam.Sam(sd, cfg, cuda)
    g = cfg["img_size"] // cfg["patch_size"]
    rng = np.random.default_rng(24)
    emb = rng.standard_normal((g, g, 256)).astype(np.float32)
    pts = (rng.random((9, 2)) * cfg["img_size"]).astype(np.float64)
    p01 = T(((pts + 0.5) / cfg["img_size"]).astype(np.float32), cuda)
    low, iou = m.decode_points(T(emb.reshape(g * g, 256), cuda), p01)
    ref_low, ref_iou = S.mask_decoder(sd, emb, S.embed_points(sd, pts, cfg["img_size"]))
    scale = float(np.abs(ref_low).max())
    np.testing.assert_allclose(low.cpu().numpy(), ref_low, rtol=0, atol=2e-4 * max(1.0, scale))
    np.testing.assert_allclose(iou.cpu().numpy(), ref_iou, rtol=0, atol=1e-4)
    # two points + padding (three sparse tokens), single-mask output
    co = np.concatenate([pts[:8].reshape(4, 2, 2), np.zeros((4, 1, 2))], 1)
    lab = np.array([[1, 0, -1], [0, 0, -1], [1, 1, -1], [0, 1, -1]])
    c01 = T(((co + 0.5) / cfg["img_size"]).astype(np.float32), cuda)
    low, iou = m.decode_prompts(T(emb.reshape(g * g, 256), cuda), c01, T(lab.astype(np.int32), cuda), first_mask=0)
    ref_low, ref_iou = S.mask_decoder(sd, emb, S.embed_prompts(sd, co, lab, cfg["img_size"]), multimask=False)
    np.testing.assert_allclose(low.cpu().numpy()[:, :1], ref_low, rtol=0, atol=2e-4 * max(1.0, float(np.abs(ref_low).max())))
    np.testing.assert_allclose(iou.cpu().numpy()[:, :1], ref_iou, rtol=0, atol=1e-4)
    # eight points + padding (T = 14: the token -> image attention runs 7 queries at a time, the general kernel on 14 keys)
    co = np.concatenate([(rng.random((3, 8, 2)) * cfg["img_size"]), np.zeros((3, 1, 2))], 1)
    lab = np.concatenate([rng.integers(0, 2, (3, 8)), np.full((3, 1), -1)], 1)
    c01 = T(((co + 0.5) / cfg["img_size"]).astype(np.float32), cuda)
    low, iou = m.decode_prompts(T(emb.reshape(g * g, 256), cuda), c01, T(lab.astype(np.int32), cuda))
    ref_low, ref_iou = S.mask_decoder(sd, emb, S.embed_prompts(sd, co, lab, cfg["img_size"]))
    np.testing.assert_allclose(low.cpu().numpy(), ref_low, rtol=0, atol=2e-4 * max(1.0, float(np.abs(ref_low).max())))
    np.testing.assert_allclose(iou.cpu().numpy(), ref_iou, rtol=0, atol=1e-4)
    del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mode", ["uncompressed_rle", "coco_rle"])
def test_generate_rle_output_modes(cuda, tiny, mode):
    """output_mode (automatic_mask_generator.py:176-182): the records of the RLE modes decode to the binary_mask records"""
    from hybridgl_amd import refer_io
    c = sam_tiny_case()
    kw = dict(points_per_side=4, pred_iou_thresh=-1e9, stability_score_thresh=0.0, crop_n_layers=0, min_mask_region_area=20,
              box_nms_thresh=1.5)
    plain = hsam.SamAutomaticMaskGenerator(tiny[1], **kw).generate(c["image"])
    anns = hsam.SamAutomaticMaskGenerator(tiny[1], output_mode=mode, **kw).generate(c["image"])
    assert len(anns) == len(plain) > 0
    for a, b in zip(anns, plain):
        seg = a["segmentation"]
        assert seg["size"] == [160, 200] and isinstance(seg["counts"], list if mode == "uncompressed_rle" else str)
        assert np.array_equal(refer_io.gt_mask_from_rle(seg)[0].astype(bool), b["segmentation"])
        assert a["area"] == b["area"] and a["bbox"] == b["bbox"]
    with pytest.raises(AssertionError):
        hsam.SamAutomaticMaskGenerator(tiny[1], output_mode="polygons", **kw)


def test_tiny_generate_with_deciding_thresholds_vs_reference(cuda, g, tiny):
    """The generator with filters that DECIDE (Hybridgl_main.py:67-73 runs pred_iou 0.7 / NMS 0.7 on trained weights): a
    6 x 6 grid, pred_iou_thresh at the 45 % quantile of the predicted IoUs, box_nms_thresh = 0.7 on sparse masks (mask
    threshold at a high logit quantile, so that boxes differ), min_mask_region_area = 20.  The reference keeps 44 of 108
    candidates; the same records must come out here (a record within 1e-3 of a threshold may flip)."""
    c = sam_tiny_case()
    iou_thr, stab_thr, nms_thr = (float(v) for v in g["dec_thr"])
    m = tiny[1]
    old = m.mask_threshold
    m.mask_threshold = float(g["dec_mask_threshold"][0])
    try:
        gen = hsam.SamAutomaticMaskGenerator(m, points_per_side=6, pred_iou_thresh=iou_thr, stability_score_thresh=stab_thr,
                                             stability_score_offset=0.25, crop_n_layers=0, min_mask_region_area=20,
                                             box_nms_thresh=nms_thr)
        anns = gen.generate(c["image"])
    finally:
        m.mask_threshold = old
    n_ref = len(g["dec_iou"])
    assert 0 < n_ref < int(g["dec_n_open"][0]) // 2          # the filters removed more than half
    ref_masks = np.unpackbits(g["dec_masks"], axis=-1)[..., :200].astype(bool)
    near = int((np.abs(g["dec_iou"] - iou_thr) < 1e-3).sum())
    assert abs(len(anns) - n_ref) <= 2 + near, (len(anns), n_ref)
    matched = same_pos = 0
    for i in range(n_ref):
        for j, a in enumerate(anns):
            if np.abs(np.array(a["point_coords"][0]) - g["dec_points"][i]).max() < 1e-9 and \
                    abs(a["predicted_iou"] - g["dec_iou"][i]) < 1e-4 and (a["segmentation"] != ref_masks[i]).mean() < 2e-3:
                assert np.abs(np.array(a["bbox"]) - g["dec_bbox"][i]).max() <= 2
                matched += 1
                same_pos += int(i == j)
                break
    assert matched >= n_ref - 2 - near, (matched, n_ref)
    assert same_pos >= n_ref - 6, (same_pos, n_ref)          # output order (NMS order) agrees


def test_nms_large_vs_oracle_and_small(cuda):
    """the bit-matrix NMS (any K) against the oracle's greedy NMS and, for K <= 1024, against the one-workgroup
    kernel: identical kept lists (score ties and duplicates included)."""
    rng = np.random.default_rng(4)
    for K in [1, 65, 700, 1025, 5000]:
        xy = rng.integers(0, 900, size=(K, 2))
        wh = rng.integers(1, 250, size=(K, 2))
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.int32)
        boxes[K // 2] = boxes[0]
        scores = rng.random(K).astype(np.float32)
        if K > 70:
            scores[70] = scores[3]
            scores[K - 1] = scores[3]
        keep = (rng.random(K) > 0.2).astype(np.uint8)
        keep[0] = 1
        idx, n = hsam.nms_large(T(boxes, cuda), T(scores, cuda), T(keep, cuda), 0.7)
        got = idx.cpu().numpy()[: int(n.item())]
        sel = np.nonzero(keep)[0]
        ref = sel[S.nms(boxes[sel].astype(np.int64), scores[sel], 0.7)]
        assert got.tolist() == ref.tolist(), K
        if K <= 1024:
            idx2, n2 = hsam.nms(T(boxes, cuda), T(scores, cuda), T(keep, cuda), 0.7)
            assert idx2.cpu().numpy()[: int(n2.item())].tolist() == got.tolist()
    # all-equal scores (the cross-crop pass scores every mask of a crop alike): index order decides
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [50, 50, 60, 60], [0, 0, 10, 10]], np.int32)
    idx, n = hsam.nms_large(T(boxes, cuda), T(np.ones(4, np.float32), cuda), T(np.ones(4, np.uint8), cuda), 0.5)
    assert idx.cpu().numpy()[: int(n.item())].tolist() == [0, 2]


def test_crop_helpers_vs_reference(cuda, golden_dir):
    """generate_crop_boxes / build_all_layer_point_grids / is_box_near_crop_edge against reference outputs."""
    import os
    gc = np.load(os.path.join(golden_dir, "sam_crops.npz"))
    cb, li = hsam.generate_crop_boxes((240, 320), 2, 512 / 1500)
    assert np.array_equal(np.array(cb), gc["crop_boxes_240x320_l2"]) and np.array_equal(np.array(li), gc["crop_layers_240x320_l2"])
    cb, _ = hsam.generate_crop_boxes((640, 480), 1, 512 / 1500)
    assert np.array_equal(np.array(cb), gc["crop_boxes_640x480_l1"])
    grids = hsam.build_all_layer_point_grids(16, 2, 2)
    assert [len(x) for x in grids] == gc["grid_sizes_16_2_2"].tolist()
    np.testing.assert_allclose(grids[2], gc["grid_l2_16_2_2"], rtol=0, atol=0)
    crop = gc["edge_crop"].tolist()
    bx = gc["edge_boxes"].astype(np.int32)
    keep = hsam.box_near_crop_edge(T(bx, cuda), T(np.ones(len(bx), np.uint8), cuda), crop, [0, 0, 320, 240])
    assert np.array_equal(keep.cpu().numpy() == 0, gc["edge_near"])


@pytest.mark.parametrize("tag,min_area", [("a", 0), ("b", 3)])
def test_tiny_generate_crops_vs_reference(cuda, golden_dir, tiny, tag, min_area):
    """SamAutomaticMaskGenerator with one crop layer (automatic_mask_generator.py:197-267): per-crop point grids,
    crop-edge filter, per-crop NMS, uncrop, cross-crop NMS, small-region clean-up -- against a reference run."""
    import os
    from oracle.cases import sam_crops_case
    gc = np.load(os.path.join(golden_dir, "sam_crops.npz"))
    c = sam_crops_case()
    m = tiny[1]
    old = m.mask_threshold
    m.mask_threshold = float(gc["mask_threshold"][0])
    try:
        gen = hsam.SamAutomaticMaskGenerator(m, points_per_side=c["points_per_side"], pred_iou_thresh=-1e9,
                                             stability_score_thresh=0.0, box_nms_thresh=c["box_nms_thresh"],
                                             crop_n_layers=c["crop_n_layers"], crop_nms_thresh=c["crop_nms_thresh"],
                                             crop_n_points_downscale_factor=c["downscale"], min_mask_region_area=min_area)
        anns = gen.generate(c["image"])
    finally:
        m.mask_threshold = old
    n_ref = int(gc[tag + "_n"][0])
    ref_masks = np.unpackbits(gc[tag + "_masks"], axis=-1)[..., :320].astype(bool)
    # thresholded noise: a logit within 1e-5 of the threshold may flip a pixel and with it an NMS decision, so the
    # records are matched by (point, crop) and a few are allowed to differ
    assert abs(len(anns) - n_ref) <= 3, (len(anns), n_ref)
    key = lambda p, cb: (round(float(p[0]), 6), round(float(p[1]), 6), tuple(int(v) for v in cb))
    ref = {}
    for i in range(n_ref):
        ref.setdefault(key(gc[tag + "_points"][i], gc[tag + "_crop_box"][i]), []).append(i)
    matched = same_pos = 0
    for j, a in enumerate(anns):
        for i in ref.get(key(a["point_coords"][0], a["crop_box"]), []):
            # the three masks of a point share the key: the predicted IoU tells them apart
            if abs(a["predicted_iou"] - gc[tag + "_iou"][i]) < 1e-4 and (a["segmentation"] != ref_masks[i]).mean() < 1e-4 \
                    and np.abs(np.array(a["bbox"]) - gc[tag + "_bbox"][i]).max() <= 1:
                assert a["area"] == int(a["segmentation"].sum())
                s_ref = gc[tag + "_stab"][i]
                assert (np.isnan(s_ref) and np.isnan(a["stability_score"])) or abs(a["stability_score"] - s_ref) < 0.35
                matched += 1
                same_pos += int(i == j)
                break
    assert matched >= n_ref - 3, (matched, n_ref)
    assert same_pos >= n_ref - 12, (same_pos, n_ref)      # output order (NMS order) agrees


def test_vit_h_two_blocks_full_width(cuda):
    """ViT-H width (1280, 16 heads of 80, 64x64 tokens, 25 padded windows, 127-entry rel-pos tables):
    one windowed + one global block + neck against the oracle at full size."""
    cfg = weights.SAM_CONFIGS["vit_h_d2"]
    sd = weights.sam_state_dict("vit_h_d2", 0)
    m = hsam.Sam(sd, cfg, cuda)
    from hybridgl_amd.synth import synth_image
    img = synth_image(683, 1024, 5)      # a 427x640 image resized: exercises the zero padding
    emb = m.encode(T(img, cuda)).cpu().numpy().reshape(64, 64, 256)
    ref = S.image_encoder(sd, S.preprocess(img, 1024), cfg)
    np.testing.assert_allclose(emb, ref, rtol=0, atol=2e-4)
    emb2 = m.encode(T(img, cuda)).cpu().numpy().reshape(64, 64, 256)
    assert np.array_equal(emb, emb2)      # run-to-run bit reproducible


def test_vit_b_two_blocks_full_width(cuda):
    """ViT-B width (768, 12 heads of 64): one windowed + one global block + neck against the oracle at full size
    (the head-dim-64 paths of the rel-pos and attention kernels, the 14x14 window without the matrix-core bias)."""
    cfg = weights.SAM_CONFIGS["vit_b_d2"]
    sd = weights.sam_state_dict("vit_b_d2", 0)
    m = hsam.Sam(sd, cfg, cuda)
    from hybridgl_amd.synth import synth_image
    img = synth_image(768, 1024, 6)
    emb = m.encode(T(img, cuda)).cpu().numpy().reshape(64, 64, 256)
    ref = S.image_encoder(sd, S.preprocess(img, 1024), cfg)
    np.testing.assert_allclose(emb, ref, rtol=0, atol=2e-4)


def test_full_size_decoder_properties(cuda):
    """64 prompts x 64x64 embedding (BASELINE size): finite, deterministic, prompt-batch independent."""
    cfg = weights.SAM_CONFIGS["vit_h_d2"]
    sd = weights.sam_state_dict("vit_h_d2", 0)
    m = hsam.Sam(sd, cfg, cuda)
    rng = np.random.default_rng(7)
    emb = T(rng.standard_normal((4096, 256)).astype(np.float32), cuda)
    pts = hsam.build_point_grid(8) * 1024.0
    p01 = T(((pts + 0.5) / 1024.0).astype(np.float32), cuda)
    low, iou = m.decode_points(emb, p01)
    assert torch.isfinite(low).all() and torch.isfinite(iou).all()
    low2, iou2 = m.decode_points(emb, p01)
    assert torch.equal(low, low2) and torch.equal(iou, iou2)
    sub_low, sub_iou = m.decode_points(emb, p01[10:13].contiguous())
    np.testing.assert_allclose(sub_low.cpu().numpy(), low[10:13].cpu().numpy(), rtol=0, atol=2e-4)
    # oracle on three prompts at full grid size
    rl, ri = S.mask_decoder(sd, emb.cpu().numpy().reshape(64, 64, 256), S.embed_points(sd, pts[10:13], 1024))
    np.testing.assert_allclose(sub_low.cpu().numpy(), rl, rtol=0, atol=5e-4)
    np.testing.assert_allclose(sub_iou.cpu().numpy(), ri, rtol=0, atol=2e-4)
    # post-process all 192 candidates at 640x640 and compare three with the oracle
    masks, boxes, stab, keep, _ = m.postprocess(low.flatten(0, 1), iou.flatten(), (1024, 1024), (640, 640))
    full = S.postprocess_masks(low[10:11].cpu().numpy(), (1024, 1024), (640, 640))[0]
    got = masks[30:33].cpu().numpy().astype(bool)
    assert ((got != (full > 0)).mean(axis=(1, 2)) < 1e-4).all()


def test_full_size_decoder_all_prompts_vs_torch_cpu_restatement(cuda):
    """The 64 prompts x 64x64 embedding of BASELINE's size with NO sampling: every prompt's three low-res logit planes and IoU
    predictions against oracle/torch_cpu.py (pinned to the numpy oracle by tests/test_torch_cpu_baseline.py), then all 192
    post-processed candidates: masks (<= 1e-4 of the pixels of any candidate may differ: a logit within rounding of the
    threshold), stability scores and boxes of the candidates whose masks agree exactly."""
    from oracle import torch_cpu as TC
    cfg = weights.SAM_CONFIGS["vit_h_d2"]
    sd = weights.sam_state_dict("vit_h_d2", 0)
    m = hsam.Sam(sd, cfg, cuda)
    rng = np.random.default_rng(7)
    emb = rng.standard_normal((4096, 256)).astype(np.float32)
    pts = hsam.build_point_grid(8) * 1024.0
    p01 = T(((pts + 0.5) / 1024.0).astype(np.float32), cuda)
    low, iou = m.decode_points(T(emb, cuda), p01)
    before = torch.get_num_threads()
    torch.set_num_threads(min(16, before))
    try:
        with torch.no_grad():
            rl, ri = TC.mask_decoder(TC.to_torch(sd), torch.from_numpy(emb.reshape(64, 64, 256)),
                                     torch.from_numpy(S.embed_points(sd, pts, 1024)))
            rb, rstab, rbox = TC.postprocess_and_stats(rl, (1024, 1024), (640, 640))
    finally:
        torch.set_num_threads(before)
    np.testing.assert_allclose(low.cpu().numpy(), rl.numpy(), rtol=0, atol=5e-4)
    np.testing.assert_allclose(iou.cpu().numpy(), ri.numpy(), rtol=0, atol=2e-4)
    masks, boxes, stab, keep, _ = m.postprocess(low.flatten(0, 1), iou.flatten(), (1024, 1024), (640, 640))
    got = masks.cpu().numpy().astype(bool)
    diff = (got != rb.numpy()).mean(axis=(1, 2))
    assert (diff < 1e-4).all(), diff.max()
    same = diff == 0
    assert same.sum() >= 150                       # nearly all of the 192 candidates agree in every pixel
    np.testing.assert_allclose(stab.cpu().numpy()[same], rstab.numpy()[same], rtol=0, atol=1e-3)


@pytest.mark.parametrize("mode", ["holes", "islands"])
def test_remove_small_regions_vs_oracle(cuda, mode):
    """device connected components (8-connectivity) vs the oracle on noisy + structured masks,
    incl. all-small islands (keep the first largest), untouched masks, empty and full masks."""
    rng = np.random.default_rng(3)
    H, W = 97, 130
    from hybridgl_amd.synth import synth_masks
    masks = synth_masks(10, H, W, 8)
    masks[0] ^= rng.random((H, W)) > 0.97             # salt-and-pepper holes and islands
    masks[1] = rng.random((H, W)) > 0.5               # pure noise: many tiny components
    masks[2] = False; masks[2, 5:8, 5:8] = True; masks[2, 50:52, 60:63] = True   # only small islands
    masks[3] = False                                   # empty
    masks[4] = True                                    # full
    masks[5] = False; masks[5, 10:13, 10:13] = True; masks[5, 40:43, 40:43] = True   # tie: equal areas
    masks[6, ::2, ::2] = False                         # checkerboard holes inside the blob (diagonal links)
    # the threshold is strict (amg.py: `s < area_thresh`): a component of EXACTLY 20 pixels stays, one of 19 goes
    masks[7] = False; masks[7, 20:70, 20:90] = True
    masks[7, 2:6, 2:7] = True; masks[7, 2:6, 100:105] = True; masks[7, 5, 104] = False          # islands of 20 and 19
    masks[7, 30:34, 30:35] = False; masks[7, 50:54, 60:65] = False; masks[7, 53, 64] = True     # holes of 20 and 19
    out, changed = hsam.remove_small_regions(T(masks.astype(np.uint8), cuda), 20, mode)
    out, changed = out.cpu().numpy().astype(bool), changed.cpu().numpy()
    for i in range(len(masks)):
        ref, ch = S.remove_small_regions(masks[i], 20, mode)
        assert bool(changed[i]) == ch, (i, mode)
        assert np.array_equal(out[i], ref), (i, mode)
    boxes = hsam.mask_boxes(T(out.astype(np.uint8), cuda)).cpu().numpy()
    assert np.array_equal(boxes.astype(np.int64), S.mask_to_box(out))


def test_remove_small_regions_full_size(cuda):
    """64 masks of 640x640 at once (BASELINE size) against the oracle on a few of them."""
    from hybridgl_amd.synth import synth_masks
    rng = np.random.default_rng(4)
    masks = synth_masks(64, 640, 640, 9)
    masks ^= rng.random(masks.shape) > 0.999
    for mode in ("holes", "islands"):
        out, changed = hsam.remove_small_regions(T(masks.astype(np.uint8), cuda), 800, mode)
        out = out.cpu().numpy().astype(bool)
        for i in (0, 17, 63):
            ref, ch = S.remove_small_regions(masks[i], 800, mode)
            assert np.array_equal(out[i], ref) and bool(changed[i]) == ch


@pytest.mark.parametrize("mode", ["holes", "islands"])
def test_remove_small_regions_speckled_full_size(cuda, mode):
    """Speckle at full size -- what a random-weight SAM hands to the clean-up in the benchmark: blobs of a few pixels at
    densities around the percolation threshold of 8-connectivity (tens of thousands of components per mask, one giant
    component that collects most runs: the wave-carried area sums and the per-row statistics of the count / stats passes)."""
    rng = np.random.default_rng(11)
    H = W = 640
    masks = np.zeros((6, H, W), dtype=bool)
    for i, p in enumerate((0.3, 0.41, 0.5, 0.59, 0.7, 0.5)):
        coarse = rng.random((H // 2 + 1, W // 2 + 1)) < p            # 2 x 2 blobs
        masks[i] = np.repeat(np.repeat(coarse, 2, axis=0), 2, axis=1)[:H, :W]
    masks[5] ^= rng.random((H, W)) > 0.9                             # + single-pixel noise
    out, changed = hsam.remove_small_regions(T(masks.astype(np.uint8), cuda), 800, mode)
    out, changed = out.cpu().numpy().astype(bool), changed.cpu().numpy()
    for i in range(len(masks)):
        ref, ch = S.remove_small_regions(masks[i], 800, mode)
        assert bool(changed[i]) == ch, (i, mode)
        assert np.array_equal(out[i], ref), (i, mode)


@pytest.mark.parametrize("shape", [(160, 200, 256), (427, 640, 1024), (640, 640, 1024), (1500, 2000, 1024)])
def test_resize_longest_side_bit_exact_vs_pillow(cuda, shape):
    from PIL import Image
    H, W, L = shape
    img = np.random.default_rng(H + W).integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    nh, nw = hsam.get_preprocess_shape(H, W, L)
    ref = np.array(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    got = hsam.resize_longest_side(T(img, cuda), L).cpu().numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_encoder_batch_of_two_equals_single_images(cuda):
    """hgl_sam_encode_batch: two images stacked along the token rows (windows of both images in one attention launch,
    row maps with per-image offsets, position embedding shared through the residual modulo) == two single-image
    passes.  Tiny geometry (fp32 GEMMs, windowed + global blocks) and the ViT-B width with 2 blocks (split-fp16 path,
    pad-row skipping, split-K in the single-image pass only)."""
    from hybridgl_amd.synth import synth_image
    for name, tol in [("tiny", 1e-5), ("vit_b_d2", 2e-4)]:
        cfg = weights.SAM_CONFIGS[name]
        m = hsam.Sam(weights.sam_state_dict(name, 0), cfg, cuda)
        S = cfg["img_size"]
        a = T(synth_image(S, S, 3), cuda)
        b = T(synth_image(S * 3 // 4, S, 4), cuda)          # a shorter image: zero-padded rows in its preprocess
        ea, eb = m.encode(a), m.encode(b)
        both = m.encode_batch([a, b])
        assert both.shape == (2,) + tuple(ea.shape)
        np.testing.assert_allclose(both[0].cpu().numpy(), ea.cpu().numpy(), rtol=0, atol=tol)
        np.testing.assert_allclose(both[1].cpu().numpy(), eb.cpu().numpy(), rtol=0, atol=tol)
        np.testing.assert_allclose(m.encode_batch([b])[0].cpu().numpy(), eb.cpu().numpy(), rtol=0, atol=0)
        del m
        torch.cuda.empty_cache()


def test_encoder_batch_of_eight_equals_single_images(cuda):
    """the benchmark's encoder pass: EIGHT images stacked along the token rows (M = 8 x 4096 at the ViT-H width, two
    blocks: one windowed, one global) == eight single-image passes, image by image"""
    from hybridgl_amd.synth import synth_image
    cfg = weights.SAM_CONFIGS["vit_h_d2"]
    m = hsam.Sam(weights.sam_state_dict("vit_h_d2", 0), cfg, cuda)
    imgs = [T(synth_image(1024 if i % 3 else 683, 1024, 20 + i), cuda) for i in range(8)]
    both = m.encode_batch(imgs)
    assert both.shape == (8, 4096, 256) and torch.isfinite(both).all()
    for i in (0, 3, 7):
        np.testing.assert_allclose(both[i].cpu().numpy(), m.encode(imgs[i]).cpu().numpy(), rtol=0, atol=2e-4)
    assert torch.equal(both, m.encode_batch(imgs))          # run-to-run bit reproducible


@pytest.mark.parametrize("mode", ["holes", "islands"])
def test_remove_small_regions_boxes_equal_mask_boxes_of_the_output(cuda, mode):
    """hgl_remove_small_regions_boxes: the boxes that come out of the clean-up's last pass (row ballots -> one word per row, folded
    per mask by box_rows_kernel) are batched_mask_to_box (utils/amg.py:303-346) of the masks it wrote -- blobs, speckle, an empty mask, a mask that
    the clean-up empties, a full mask, ragged width (not a multiple of 64)."""
    rng = np.random.default_rng(5)
    H, W = 97, 150
    m = np.zeros((7, H, W), dtype=np.uint8)
    m[0, 10:40, 20:90] = 1; m[0, 20:25, 30:35] = 0; m[0, 60:62, 100:102] = 1
    m[1] = rng.random((H, W)) < 0.5
    m[2] = 0
    m[3, 5:7, 140:150] = 1                      # small island only: islands keeps the largest, holes leaves it
    m[4] = 1
    m[5] = rng.random((H, W)) < 0.08
    m[6, :, 149] = 1; m[6, 96, :] = 1
    mt = T(m, cuda)
    out, ch, bx = hsam.remove_small_regions_boxes(mt, 30, mode)
    out0, ch0 = hsam.remove_small_regions(mt, 30, mode)
    assert torch.equal(out, out0) and torch.equal(ch, ch0)
    assert torch.equal(bx, hsam.mask_boxes(out))
