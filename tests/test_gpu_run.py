"""HybridGLPipeline.run -- the product's evaluation loop at the grouped, two-stream rate (Hybridgl_main.py:45,79-230
taken several images at a time) -- against the per-ref step() it replaces: same metric rows bit for bit, on REFER-format
data with images of different sizes, several refs per image, ragged proposal counts and SAM's OWN masks feeding CLIP."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dataset(root, n_images=9):
    """REFER directory layout with `n_images` images of different sizes and 1-3 consecutive refs per image (18 refs)."""
    from PIL import Image
    from hybridgl_amd.synth import synth_image
    (root / "refcoco").mkdir(parents=True)
    img_dir = root / "images/mscoco/images/train2014"
    img_dir.mkdir(parents=True)
    images, anns, refs = [], [], []
    rid = 200
    for i in range(n_images):
        h, w = 96 + 16 * (i % 4), 128 + 24 * (i % 3)
        name = f"COCO_train2014_{i:012d}.png"
        Image.fromarray(synth_image(h, w, 70 + i)).save(img_dir / name)
        images.append({"id": 10 + i, "file_name": name, "height": h, "width": w})
        for j in range(1 + i % 3):
            aid = 1000 + rid
            x0 = 8 + 20 * j
            anns.append({"id": aid, "image_id": 10 + i, "category_id": 1,
                         "segmentation": [[x0, 12, x0 + 70, 15, x0 + 60, h - 10, x0 + 5, h - 20]], "bbox": [0, 0, 1, 1]})
            sents = [{"sent_id": 2 * rid, "raw": "the cat on left", "tokens": []}]
            if (i + j) % 2:
                sents.append({"sent_id": 2 * rid + 1, "raw": "a big dog", "tokens": []})
            refs.append({"ref_id": rid, "ann_id": aid, "image_id": 10 + i, "category_id": 1, "split": "val",
                         "sent_ids": [s["sent_id"] for s in sents], "sentences": sents})
            rid += 1
    json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "thing"}]},
              open(root / "refcoco/instances.json", "w"))
    pickle.dump(refs, open(root / "refcoco/refs(unc).p", "wb"))
    parse = {}
    for r in refs:
        parse[str(r["sent_ids"][0])] = {"noun_phrase": "the cat", "other_nouns": ["left"], "dirflag": "left", "relaflag": "left"}
        if len(r["sent_ids"]) > 1:
            parse[str(r["sent_ids"][1])] = {"noun_phrase": "dog", "other_nouns": [], "dirflag": "none", "relaflag": "big"}
    json.dump(parse, open(root / "parse.json", "w"))
    return refs


@pytest.fixture(scope="module")
def models(cuda):
    from hybridgl_amd.backbone import CLIPViTFM
    from hybridgl_amd.gem import create_gem_model
    from hybridgl_amd.sam import SamAutomaticMaskGenerator, sam_model_registry
    model = CLIPViTFM("ViT-B/16", seed=0, device=cuda)
    gem = create_gem_model("ViT-B/16", clip=model)
    # thresholds that DECIDE: the 0.7 box NMS of the reference on a 5x5 grid, an IoU threshold near the median of the
    # tiny model's predictions -> a different number of proposals for every image
    sam = sam_model_registry["tiny"](device=cuda)
    gen = SamAutomaticMaskGenerator(sam, points_per_side=5, pred_iou_thresh=-1.0,
                                    stability_score_thresh=0.0, box_nms_thresh=0.7, min_mask_region_area=2)
    # random weights give noise logits whose masks at threshold 0 all span the image (one survivor of the NMS): a mask
    # threshold at the 99.95 % quantile leaves a handful of pixels per candidate, so the boxes differ and the NMS decides
    from hybridgl_amd.sam import SamPredictor
    from hybridgl_amd.synth import synth_image
    pred = SamPredictor(sam)
    pred.set_image(torch.from_numpy(synth_image(96, 128, 70)).to(cuda))
    pts = torch.from_numpy(pred.transform.apply_coords(gen.point_grids[0] * np.array([[128, 96]]), (96, 128)))[:, None, :]
    logits, _, _ = pred.predict_torch(pts, torch.ones((len(pts), 1), dtype=torch.int64), return_logits=True)
    sam.mask_threshold = float(torch.quantile(logits.flatten()[:4_000_000].float(), 0.9995))
    return model, gem, gen


def _args(root, golden_dir, heatmap="device"):
    from hybridgl_amd import main as drv
    return drv.default_argument_parser().parse_args([
        "--real", "--refer_data_root", str(root), "--dataset", "refcoco", "--split", "val",
        "--bpe_vocab", os.path.join(golden_dir, "tiny_bpe_vocab.txt.gz"), "--parse_json", str(root / "parse.json"),
        "--heatmap", heatmap])


def test_run_equals_per_ref_steps_on_a_refer_set(cuda, golden_dir, tmp_path, models):
    """18 refs over 9 images of 9 different sizes: (a) step() ref by ref on one stream, items prepared on the calling
    thread; (b) run() in groups of 8 images fed by 4 loader threads; (c) groups of 3.  The metric rows (dataset position,
    sentence, I, U, I_final, U_final) are identical, and so is the report."""
    from hybridgl_amd import main as drv
    from hybridgl_amd.loader import Prefetcher
    from hybridgl_amd.pipeline import HybridGLPipeline
    model, gem, gen = models
    root = tmp_path / "refer_data"
    refs = _dataset(root)
    assert len(refs) >= 16
    args = _args(root, golden_dir)
    rr = drv.RealRefs(args, cuda, "unc", 77)
    mk = lambda: HybridGLPipeline(model, fusion_mode="G2L", masking_block=9, mask_generator=gen, use_sam_masks=True,
                                  gem_model=gem)
    a = mk()
    counts = []
    for i in rr.jobs():
        a.step(rr.load(i))
        counts.append(int(a._cache_ref.masks.shape[0]))
    torch.cuda.synchronize()
    assert len(set(counts)) >= 3, f"the proposal counts should be ragged, got {counts}"
    b = mk()
    n = b.run(Prefetcher(rr.jobs(), rr.load, workers=4, depth=12, device=cuda), group=8)
    torch.cuda.synchronize()
    assert n == len(refs) and b.skipped == 0
    rows_a, rows_b = a.partial_rows(), b.partial_rows()
    assert rows_a.shape == (sum(len(r["sent_ids"]) for r in refs), 6)
    assert np.array_equal(rows_a, rows_b)
    assert a.metrics() == b.metrics()
    c = mk()
    c.run(Prefetcher(rr.jobs(), rr.load, workers=2, depth=4, device=cuda), group=3, collect=True)
    torch.cuda.synchronize()
    assert np.array_equal(rows_a, c.partial_rows())
    assert len(c.collected) == len(refs)
    # the k1 / k2 clamp state after the loop is the per-ref one as well (same order of refs)
    assert (a.k1, a.k2) == (b.k1, b.k2) == (c.k1, c.k2)


def test_run_reuses_images_that_come_back(cuda, golden_dir, tmp_path, models):
    """The refs of an image are not always neighbours in the loader's order: with the same images coming back after other
    images (the dataset's refs re-ordered round-robin over the images), run() takes proposals, hybrid and GEM features of a
    returning image from its image cache -- fewer proposal stages, identical rows to the per-ref loop, cache on or off."""
    from hybridgl_amd import main as drv
    from hybridgl_amd.pipeline import HybridGLPipeline
    model, gem, gen = models
    root = tmp_path / "refer_data"
    refs = _dataset(root)
    # re-order: first refs of all images, then second refs, then third refs
    import pickle
    by_img = {}
    for r in refs:
        by_img.setdefault(r["image_id"], []).append(r)
    order = [rs[k] for k in range(3) for rs in by_img.values() if k < len(rs)]
    assert [r["image_id"] for r in order] != [r["image_id"] for r in refs]
    pickle.dump(order, open(root / "refcoco/refs(unc).p", "wb"))
    args = _args(root, golden_dir)
    rr = drv.RealRefs(args, cuda, "unc", 77)
    mk = lambda **kw: HybridGLPipeline(model, fusion_mode="G2L", masking_block=9, mask_generator=gen, use_sam_masks=True,
                                       gem_model=gem, **kw)
    a = mk()
    for i in rr.jobs():
        a.step(rr.load(i))
    b, c = mk(image_cache=32), mk(image_cache=0)
    for p, g in ((b, 4), (c, 4)):
        assert p.run((rr.load(i) for i in rr.jobs()), group=g) == len(order)
    torch.cuda.synchronize()
    ra, rb, rc = a.partial_rows(), b.partial_rows(), c.partial_rows()
    assert np.array_equal(ra, rc), (ra[(ra != rc).any(axis=1)][:4].tolist(), rc[(ra != rc).any(axis=1)][:4].tolist())
    assert np.array_equal(ra, rb), (ra[(ra != rb).any(axis=1)][:4].tolist(), rb[(ra != rb).any(axis=1)][:4].tolist())
    assert b.cache_hits == len(order) - len(by_img) and c.cache_hits == 0


def test_run_skips_images_without_proposals_and_goes_on(cuda, golden_dir, tmp_path, models):
    """An IoU threshold no candidate passes on some images: their refs are counted as skipped (the reference would fail
    at torch.stack([])), the rest of the group is scored; rows equal the per-ref loop's, which skips the same refs."""
    from hybridgl_amd import main as drv
    from hybridgl_amd.pipeline import EmptyProposals, HybridGLPipeline
    from hybridgl_amd.sam import SamAutomaticMaskGenerator
    model, gem, gen0 = models
    root = tmp_path / "refer_data"
    refs = _dataset(root, n_images=6)
    args = _args(root, golden_dir, heatmap="given")
    rr = drv.RealRefs(args, cuda, "unc", 77)
    # find a threshold between the images' best predicted IoUs
    best = []
    for i in rr.jobs():
        r = rr.load(i)
        _, _, iou, _, order, n, _ = gen0.propose(r.sam_img)
        best.append(float(iou.max()))
    thr = float(np.median(sorted(set(best))))
    gen = SamAutomaticMaskGenerator(gen0.model, points_per_side=5, pred_iou_thresh=thr, stability_score_thresh=0.0,
                                    box_nms_thresh=0.7, min_mask_region_area=2)
    mk = lambda: HybridGLPipeline(model, mask_generator=gen, use_sam_masks=True, k_clamp="per_ref")
    a, skipped = mk(), 0
    for i in rr.jobs():
        try:
            a.step(rr.load(i))
        except EmptyProposals:
            skipped += 1
    b = mk()
    n = b.run((rr.load(i) for i in rr.jobs()), group=4)
    torch.cuda.synchronize()
    assert 0 < skipped < len(refs) and b.skipped == skipped and n == len(refs) - skipped
    assert np.array_equal(a.partial_rows(), b.partial_rows())


def test_run_on_given_proposals_equals_step_group(cuda):
    """Without a mask generator run() is the grouped CLIP + scoring stage over the items' own masks (scope A)."""
    from hybridgl_amd.backbone import CLIPViTFM
    from hybridgl_amd.pipeline import HybridGLPipeline, synthetic_ref
    model = CLIPViTFM("ViT-B/16", seed=0, device=cuda)
    refs = [synthetic_ref(i, cuda, N=6 + i, H=120, W=160)[0] for i in range(5)]
    a, b = HybridGLPipeline(model), HybridGLPipeline(model)
    for r in refs:
        a.step(r)
    assert b.run(iter(refs), group=4) == 5
    torch.cuda.synchronize()
    assert np.array_equal(a.partial_rows(), b.partial_rows())


def test_hybrid_forward_over_segments_equals_per_image_calls(cuda):
    """hgl_clip_hybrid_forward_segments: the proposals of three images of different sizes in one forward == the three
    forwards, row for row (every mask row is independent); wrong run lengths are refused."""
    from hybridgl_amd._lib import HybridGLError
    from hybridgl_amd.backbone import CLIPViTFM
    from hybridgl_amd.synth import synth_masks
    model = CLIPViTFM("ViT-B/16", seed=0, device=cuda)
    rng = np.random.default_rng(3)
    sizes, ns = [(120, 160), (97, 130), (200, 64)], [9, 1, 14]
    masks = [torch.from_numpy(synth_masks(n, h, w, 40 + i)).to(cuda) for i, ((h, w), n) in enumerate(zip(sizes, ns))]
    loc = torch.from_numpy(rng.standard_normal((sum(ns), 3, 224, 224)).astype(np.float32)).to(cuda)
    glo = torch.from_numpy(rng.standard_normal((sum(ns), 3, 224, 224)).astype(np.float32)).to(cuda)
    for mode in ("G2L", "L2G", "G2L&L2G"):
        whole = model(loc, glo, masks, masking_block=9, fusion_mode=mode)
        o = 0
        for mk, n in zip(masks, ns):
            # 9, 1 and 14 masks alone are small-M launches that take other GEMM kernels than the stacked 24: rounding-level
            part = model(loc[o:o + n], glo[o:o + n], mk, masking_block=9, fusion_mode=mode)
            assert torch.allclose(whole[o:o + n], part, rtol=0, atol=2e-5), mode
            o += n
    with pytest.raises(HybridGLError, match="mask runs hold"):
        model(loc, glo, masks[:2], masking_block=9, fusion_mode="G2L")


@pytest.fixture(scope="module")
def full(cuda):
    """ViT-B/16 + GEM + SAM ViT-H with seeded weights (full size)"""
    from hybridgl_amd.backbone import CLIPViTFM
    from hybridgl_amd.gem import create_gem_model
    from hybridgl_amd.sam import sam_model_registry
    model = CLIPViTFM("ViT-B/16", seed=0, device=cuda)
    yield model, create_gem_model("ViT-B/16", clip=model), sam_model_registry["default"](seed=0, device=cuda)
    torch.cuda.empty_cache()


def test_full_size_dependent_groups_are_composition_independent(cuda, full):
    """BASELINE size (640x640, SAM ViT-H, ViT-B/16, 64 of SAM's own masks per image through clean-up into CLIP): run() in
    groups of 4 and in groups of 2 file the same rows bit for bit (no encoder pass of these sizes uses split-K), run to run
    as well; the proposals CLIP scored are SAM's (not the items' seeded masks)."""
    from hybridgl_amd.pipeline import HybridGLPipeline, synthetic_ref
    from hybridgl_amd.sam import SamAutomaticMaskGenerator
    model, gem, sam = full
    gen = SamAutomaticMaskGenerator(sam, points_per_side=8, pred_iou_thresh=-1e30,
                                    stability_score_thresh=0.0, box_nms_thresh=2.0, min_mask_region_area=800)
    refs = [synthetic_ref(i, cuda, N=64, sam_img_size=1024, gem=True, device_blur=True)[0] for i in range(8)]
    mk = lambda: HybridGLPipeline(model, mask_generator=gen, use_sam_masks=True, gem_model=gem)
    rows = []
    for g in (4, 2, 4):
        p = mk()
        assert p.run(iter(refs), group=g, proposal_cap=64, collect=True) == 8
        torch.cuda.synchronize()
        rows.append(p.partial_rows())
        assert all(h.shape == (64, 512) for h, _, _ in p.collected)
    assert np.array_equal(rows[0], rows[1]) and np.array_equal(rows[0], rows[2])
    assert rows[0].shape == (24, 6) and (rows[0][:, 3] > 0).all()


def test_phrasecut_configuration_at_full_size(cuda, full):
    """BASELINE configs[4] (Hybridgl_main_PhraseCut.py:56-62): 64 x 64 points + one crop layer (downscale 2), min area 100, on
    a 480 x 640 image with SAM ViT-H -- 5 encoder passes, 128 decoder batches, per-crop NMS over 12 288 / 3 072 candidates
    (bit-matrix kernels), cross-crop NMS, small-region clean-up, last NMS.  No oracle can run this size in seconds, so the
    checks are the properties the reference's generator guarantees: determinism, output records consistent with each other
    (boxes = batched_mask_to_box of the masks, crop boxes from the crop list, layer-1 masks inside their crop), clean-up
    idempotent on a sample, and independence of the composition of the CLIP batch the proposals are scored in."""
    from hybridgl_amd import sam as hsam
    from hybridgl_amd.pipeline import HybridGLPipeline, synthetic_ref
    from hybridgl_amd.synth import synth_image
    from oracle import sam_oracle as S
    model, gem, sam = full
    # thresholds: the reference's 0.86 / 0.92 reject everything a random-weight model predicts; they sit at values that keep a
    # part of the candidates, so that every filter decides
    H, W = 480, 640
    img = torch.from_numpy(synth_image(H, W, 9)).to(cuda)
    probe = hsam.SamAutomaticMaskGenerator(sam, points_per_side=8, pred_iou_thresh=-1e30, stability_score_thresh=0.0, box_nms_thresh=2.0)
    _, _, iou, stab, _, _, _ = probe.propose(img)
    iou_thr = float(torch.quantile(iou, 0.5))
    gen = hsam.SamAutomaticMaskGenerator(sam, points_per_side=64, pred_iou_thresh=iou_thr, stability_score_thresh=0.0,
                                         crop_n_layers=1, crop_n_points_downscale_factor=2, min_mask_region_area=100)
    a = gen.generate_device_crops(img)
    b = gen.generate_device_crops(img)
    torch.cuda.synchronize()
    n = a[0].shape[0]
    assert n > 0 and all(torch.equal(x, y) for x, y in zip(a[:4], b[:4])) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])
    masks, xywh, iou_k, stab_k, pts, cbs = a
    assert masks.shape == (n, H, W) and xywh.shape == (n, 4) and (iou_k >= iou_thr).all()
    crop_list, layers = hsam.generate_crop_boxes((H, W), 1, 512 / 1500)
    assert len(crop_list) == 5 and set(map(tuple, cbs.tolist())) <= set(map(tuple, crop_list))
    # boxes are the boxes of the masks (XYXY inclusive rule -> XYWH), for every record
    bx = hsam.mask_boxes(masks.contiguous()).long()
    assert torch.equal(torch.stack([bx[:, 0], bx[:, 1], bx[:, 2] - bx[:, 0], bx[:, 3] - bx[:, 1]], 1), xywh)
    mk = masks.cpu().numpy().astype(bool)
    sel = np.linspace(0, n - 1, min(n, 6)).astype(int)
    for i in sel:
        x0, y0, x1, y1 = cbs[i]
        inside = np.zeros((H, W), bool)
        inside[y0:y1, x0:x1] = True
        assert not (mk[i] & ~inside).any()                      # a crop's mask never leaves its crop (uncrop_masks pads with 0)
        assert x0 <= pts[i][0] <= x1 and y0 <= pts[i][1] <= y1   # its prompt point lies in the crop
        # the clean-up already ran: running it again changes nothing (utils/amg.py:267-291)
        m1, ch1 = S.remove_small_regions(mk[i], 100, "holes")
        m2, ch2 = S.remove_small_regions(m1, 100, "islands")
        assert np.array_equal(m2, mk[i]) or mk[i].sum() < 100    # (a mask whose largest island is small is kept as it is)
    # the same image twice in one group / alone: same rows
    ref = synthetic_ref(50, cuda, N=8, H=H, W=W, n_sent=8, sam_img_size=1024, gem=True, device_blur=True)[0]
    ref2 = synthetic_ref(51, cuda, N=8, H=H, W=W, n_sent=8, sam_img_size=1024, gem=True, device_blur=True)[0]
    mkp = lambda: HybridGLPipeline(model, fusion_mode="G2L&L2G", mask_generator=gen, use_sam_masks=True, gem_model=gem)
    p1, p2 = mkp(), mkp()
    assert p1.run(iter([ref, ref2]), group=2, proposal_cap=128) == 2
    assert p2.run(iter([ref, ref2]), group=1, proposal_cap=128) == 2
    torch.cuda.synchronize()
    r1, r2 = p1.partial_rows(), p2.partial_rows()
    assert r1.shape == (16, 6) and np.array_equal(r1, r2)


def test_crop_layers_in_groups_equal_image_by_image(cuda, models):
    """SamAutomaticMaskGenerator.crops_begin / crops_mid / crops_post / crops_finish (three host syncs per GROUP of images)
    against generate_device_crops (one per crop + two per image): identical masks, boxes, scores, for images of different
    sizes, with thresholds that decide."""
    from hybridgl_amd.sam import SamAutomaticMaskGenerator, sam_model_registry
    from hybridgl_amd.synth import synth_image
    sam = models[2].model      # the tiny SAM whose mask threshold leaves sparse masks: boxes differ, the NMS passes decide
    imgs = [torch.from_numpy(synth_image(h, w, 11 + i)).to(cuda) for i, (h, w) in enumerate(((150, 200), (200, 150), (120, 120)))]
    gen = SamAutomaticMaskGenerator(sam, points_per_side=6, points_per_batch=16, pred_iou_thresh=0.0, stability_score_thresh=0.0,
                                    crop_n_layers=1, crop_n_points_downscale_factor=2, min_mask_region_area=10)
    # thresholds that decide: the IoU threshold at the median prediction of the first image, sparse masks for the NMS passes
    probe = gen.generate_device_crops(imgs[0])
    assert len(probe[2]) > 4
    gen.pred_iou_thresh = float(torch.quantile(probe[2], 0.3))
    one = [gen.generate_device_crops(im) for im in imgs]
    grp = gen.generate_crops_group(imgs)
    assert len(grp) == 3 and sum(a[0].shape[0] for a in one) > 0 and 0 < one[0][0].shape[0] < probe[0].shape[0]
    for a, b in zip(one, grp):
        assert a[0].shape == b[0].shape
        for x, y in zip(a[:4], b[:4]):      # stability of an empty mask is 0 / 0: NaN in both
            assert torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0)) if x.is_floating_point() else torch.equal(x, y)
    # an IoU threshold nothing passes: empty outputs of the right shapes, no failure
    gen.pred_iou_thresh = 1e9
    for (m, bx, iou, stab), im in zip(gen.generate_crops_group(imgs), imgs):
        assert m.shape == (0,) + tuple(im.shape[:2]) and bx.shape == (0, 4) and iou.shape == stab.shape == (0,)


def test_phrasecut_from_disk_groups_equal_image_by_image(cuda, tmp_path):
    """python -m hybridgl_amd.main --dataset phrasecut on a tree in the published VGPhraseCut layout (synth.write_phrasecut_tree:
    image files smaller than their annotation included): the grouped loop (crop layers through the three-sync protocol,
    loader threads, device transforms) reports exactly what the image-by-image loop (--group 1: step()) reports; every
    phrase is scored against its own ground truth."""
    from hybridgl_amd import main as drv, synth
    root = str(tmp_path / "pc")
    info = synth.write_phrasecut_tree(root, n_images=5, phrases_per_image=3, sizes=((120, 160), (160, 120), (96, 128)), resized_files=0.5)
    base = ["--real", "--dataset", "phrasecut", "--split", "test", "--phrasecut_root", root, "--bpe_vocab", os.path.join(root, "bpe.txt.gz"),
            "--parse_json", os.path.join(root, "parse.json"), "--sam_model", "tiny", "--points_per_side", "4", "--points_per_batch", "16",
            "--pred_iou_thresh", "0.0", "--stability_score_thresh", "0", "--min_mask_region_area", "10", "--fusion_mode", "G2L&L2G",
            "--proposal_cap", "24", "--k_clamp", "per_ref", "--result_dir", str(tmp_path / "log")]
    args = drv.default_argument_parser().parse_args(base + ["--group", "2"])
    model, gen, gem = drv.build_models(args, cuda)
    assert gen.crop_n_layers == 1 and len(gen.point_grids) == 2
    m_grp, st = drv.evaluate(args, model, gen, gem, cuda)
    assert st["refs"] == info["images"] and m_grp["n_sentences"] == info["phrases"] and st["skipped"] == 0
    uncapped = [a for a in base if a not in ("--proposal_cap", "24")]
    with pytest.raises(SystemExit):       # step() applies no proposal cap: the flag combination is refused, not silently ignored
        drv.evaluate(drv.default_argument_parser().parse_args(base + ["--group", "1"]), model, gen, gem, cuda)
    args1 = drv.default_argument_parser().parse_args(uncapped + ["--group", "1"])
    m_one, st1 = drv.evaluate(args1, model, gen, gem, cuda)
    assert m_one["n_sentences"] == info["phrases"]
    args_nc = drv.default_argument_parser().parse_args(uncapped + ["--group", "3"])
    m_nc, _ = drv.evaluate(args_nc, model, gen, gem, cuda)
    assert m_nc == m_one
