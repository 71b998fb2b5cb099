"""Host logic of hybridgl_amd/transforms.py and of the synthetic REFER tree (no GPU): the Pillow coefficient tables the
device resampler is fed (checked through a numpy statement of the same integer arithmetic against Pillow itself), the
normalisation table against the host transform, the tree writer against the REFER reader."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hybridgl_amd import synth, transforms as T   # noqa: E402


@pytest.mark.parametrize("hw", [(480, 640), (375, 500), (97, 131), (448, 448), (700, 300)])
def test_pillow_tables_reproduce_pillow(hw):
    from PIL import Image
    img = np.random.default_rng(hw[0]).integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
    for filt, pf in (("bicubic", Image.BICUBIC), ("bilinear", Image.BILINEAR)):
        want = np.asarray(Image.fromarray(img).resize((448, 448), pf))
        assert np.array_equal(T.pil_resample_numpy(img, 448, 448, filt), want), filt
    # the tables of the bilinear filter are the ones SAM's ResizeLongestSide path has used since round 1
    from hybridgl_amd.sam import pil_bilinear_coeffs
    for a, b in zip(T.pil_coeffs(hw[1], 448, "bilinear"), pil_bilinear_coeffs(hw[1], 448)):
        assert np.array_equal(a, b)


def test_normalisation_tables_are_the_host_transforms():
    from hybridgl_amd.gem import OPENAI_MEAN, OPENAI_STD, get_gem_img_transform
    img = np.random.default_rng(3).integers(0, 256, (120, 160, 3), dtype=np.uint8)
    lut = T.normalize_lut(T.IMAGENET_MEAN, T.IMAGENET_STD)
    assert np.array_equal(np.stack([lut[c][img[..., c]] for c in range(3)]), synth.imagenet_normalize(img))
    lut = T.normalize_lut(OPENAI_MEAN, OPENAI_STD)
    r = T.pil_resample_numpy(img, 448, 448, "bicubic")
    assert np.array_equal(np.stack([lut[c][r[..., c]] for c in range(3)]), get_gem_img_transform()(img).numpy())


def test_synthetic_refer_tree_reads_back(tmp_path):
    from hybridgl_amd.refer_io import ReferDataset
    from hybridgl_amd.tokenizer import SimpleTokenizer, tokenize
    info = synth.write_refer_tree(str(tmp_path), n_images=12, sizes=((60, 80), (80, 60)), far_refs=0.5)
    ds = ReferDataset(str(tmp_path), "refcoco", "unc", "val")
    assert len(ds) == info["refs"] and 24 <= info["refs"] <= 36 and info["sentences"] == 3 * info["refs"]
    ids = [ds.image_id(i) for i in range(len(ds))]
    runs = sum(1 for i in range(len(ids)) if i == 0 or ids[i] != ids[i - 1])
    assert runs > len(set(ids)) == 12                 # some refs come back after other images
    data, annot, sents = ds[0]
    assert data["sam_img"].shape in ((60, 80, 3), (80, 60, 3)) and annot.shape == data["sam_img"].shape[:2]
    assert 0 < annot.sum() < annot.size and len(sents) == 3
    assert np.array_equal(ds.image(0), data["sam_img"]) and np.array_equal(ds.target(0), annot)
    tk = SimpleTokenizer(str(tmp_path / "bpe.txt.gz"))
    tok = tokenize(sents, tokenizer=tk)
    assert tok.shape == (3, 77) and (tok.argmax(axis=1) >= 3).all() and tok.max() == tk.eot
    import json
    parse = json.load(open(tmp_path / "parse.json"))
    assert set(parse) == {str(s) for r in ds.ref_ids for s in ds.refer.Refs[r]["sent_ids"]}


def test_phrasecut_tree_reads_back(tmp_path):
    """hybridgl_amd/phrasecut_io.py on a tree in the published VGPhraseCut layout: one item per image, every phrase with the
    polygons of all its instances, the image brought to the annotation's size, ground truth as dataset_phrasecut.py:108-122
    rasterises it; the seen / unseen filters."""
    from hybridgl_amd.phrasecut_io import COCO_CLASSES, PhraseCutDataset, RefVGLoader, cv_resize_linear_u8
    info = synth.write_phrasecut_tree(str(tmp_path), n_images=6, phrases_per_image=4, sizes=((60, 80), (80, 60)), resized_files=0.5)
    ds = PhraseCutDataset(str(tmp_path), "test")
    assert len(ds) == 6 and ds.refvg_loader.img_ids == sorted(ds.refvg_loader.img_ids)
    n_resized = 0
    for i in range(6):
        it = ds[i]
        assert it["sam_img"].shape == (it["height"], it["width"], 3) and len(it["phrases"]) == 4
        n_resized += it["file_img"] is not None
        gt = ds.gt_mask(it, 1)
        assert gt.shape == (it["height"], it["width"]) and gt.dtype == bool and 0 < gt.sum() < gt.size
    assert 0 < n_resized < 6
    assert sum(len(ds[i]["phrases"]) for i in range(6)) == info["phrases"]
    with pytest.raises(FileNotFoundError):
        RefVGLoader(str(tmp_path), "val")
    seen, unseen = PhraseCutDataset(str(tmp_path), "test", seen_mode=True), PhraseCutDataset(str(tmp_path), "test", unseen_mode=True)
    for i in range(6):
        a, b = seen[i], unseen[i]
        assert len(a["phrases"] if a else []) + len(b["phrases"] if b else []) == 4
    names = {t["phrase_structure"]["name"] for ts in ds.refvg_loader.ImgReferTasks.values() for t in ts}
    assert names & set(COCO_CLASSES) and names - set(COCO_CLASSES)
    # the cv2.resize restatement: identity at equal size, constants stay constant, a ramp stays within one grey level of
    # the real-valued interpolation (parity with opencv-python itself is unpinned: absent offline)
    img = np.random.default_rng(0).integers(0, 256, (30, 40, 3), dtype=np.uint8)
    assert cv_resize_linear_u8(img, 40, 30) is img
    assert (cv_resize_linear_u8(np.full((30, 40, 3), 77, np.uint8), 80, 45) == 77).all()
    ramp = np.tile(np.arange(0, 200, 5, dtype=np.uint8)[None, :, None], (30, 1, 3))
    x = np.clip((np.arange(80) + 0.5) * 0.5 - 0.5, 0, 39)
    assert np.abs(cv_resize_linear_u8(ramp, 80, 30)[0, :, 0] - np.interp(x, np.arange(40), np.arange(0, 200, 5))).max() <= 1.0


def test_dataset_defaults_follow_the_reference_scripts():
    from hybridgl_amd import main as drv
    a = drv.resolve_defaults(drv.default_argument_parser().parse_args(["--dataset", "refcocog"]))
    assert (a.points_per_side, a.pred_iou_thresh, a.stability_score_thresh, a.min_mask_region_area, a.crop_n_layers, a.group) == \
           (8, 0.7, 0.7, 800, 0, 16)                                             # Hybridgl_main.py:67-73
    b = drv.resolve_defaults(drv.default_argument_parser().parse_args(["--dataset", "phrasecut", "--group", "2"]))
    assert (b.points_per_side, b.pred_iou_thresh, b.stability_score_thresh, b.min_mask_region_area, b.crop_n_layers,
            b.crop_n_points_downscale_factor, b.group) == (64, 0.86, 0.92, 100, 1, 2, 2)   # Hybridgl_main_PhraseCut.py:56-62
    assert (a.split, b.split) == ("val", "test")                                 # Hybridgl_main_PhraseCut.py:42: split='test'
    import pytest
    with pytest.raises(SystemExit):      # the ref-by-ref path keeps every proposal: a cap there would be silently ignored
        drv.resolve_defaults(drv.default_argument_parser().parse_args(["--group", "1", "--proposal_cap", "64"]))
    assert T.resize_shorter_side(480, 640, 800) == (800, 1066) and T.resize_shorter_side(640, 480, 800) == (1066, 800)
