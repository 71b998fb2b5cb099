"""Driver on REFER-format data: annotations -> native GT masks -> SAM proposals -> CLIP hybrid
-> scoring -> IoU, end to end on the device, with a synthetic dataset in the reference's directory layout."""
import json
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dataset(root):
    from PIL import Image
    from hybridgl_amd.synth import synth_image
    (root / "refcoco").mkdir(parents=True)
    img_dir = root / "images/mscoco/images/train2014"
    img_dir.mkdir(parents=True)
    images, anns, refs = [], [], []
    for i in range(2):
        h, w = 120 + 20 * i, 160
        name = f"COCO_train2014_{i:012d}.png"
        Image.fromarray(synth_image(h, w, 50 + i)).save(img_dir / name)
        images.append({"id": 10 + i, "file_name": name, "height": h, "width": w})
        for j in range(2):      # two refs per image: the second reuses the image's cached proposals / features
            aid, rid = 100 + 2 * i + j, 200 + 2 * i + j
            anns.append({"id": aid, "image_id": 10 + i, "category_id": 1,
                         "segmentation": [[10 + 30 * j, 12, 90 + 30 * j, 15, 80 + 30 * j, 100, 15 + 30 * j, 90]], "bbox": [0, 0, 1, 1]})
            refs.append({"ref_id": rid, "ann_id": aid, "image_id": 10 + i, "category_id": 1, "split": "val",
                         "sent_ids": [2 * rid, 2 * rid + 1],
                         "sentences": [{"sent_id": 2 * rid, "raw": "the cat on left", "tokens": []},
                                       {"sent_id": 2 * rid + 1, "raw": "a big dog", "tokens": []}]})
    json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "thing"}]},
              open(root / "refcoco/instances.json", "w"))
    pickle.dump(refs, open(root / "refcoco/refs(unc).p", "wb"))
    parse = {str(2 * 200): {"noun_phrase": "the cat", "other_nouns": ["left"], "dirflag": "left", "relaflag": "none"},
             str(2 * 200 + 1): {"noun_phrase": "dog", "other_nouns": [], "dirflag": "none", "relaflag": "big"}}
    json.dump(parse, open(root / "parse.json", "w"))
    return refs


@pytest.mark.parametrize("heatmap", ["device", "given"])
def test_driver_runs_refer_data_end_to_end(cuda, golden_dir, tmp_path, heatmap):
    from hybridgl_amd import main as drv
    root = tmp_path / "refer_data"
    refs = _dataset(root)
    args = drv.default_argument_parser().parse_args([
        "--real", "--refer_data_root", str(root), "--dataset", "refcoco", "--split", "val", "--sam_model", "tiny",
        "--bpe_vocab", os.path.join(golden_dir, "tiny_bpe_vocab.txt.gz"), "--parse_json", str(root / "parse.json"),
        "--pred_iou_thresh", "-1", "--stability_score_thresh", "0", "--min_mask_region_area", "20",
        "--points_per_side", "4", "--result_dir", str(tmp_path / "log"), "--heatmap", heatmap])
    m = drv.main(args)
    assert m["n_sentences"] == 2 * len(refs)
    assert m["cum"][1] > 0 and 0.0 <= m["oIoU"] <= 100.0 and 0.0 <= m["oIoU_final"] <= 100.0
    log = open(tmp_path / "log" / "result_log_refcoco_val.txt").read()
    assert "pure hybridgl:" in log and "hybridgl w/ spatial guidance:" in log and "refcoco / val / unc" in log


def test_phrasecut_item_with_crop_layers(cuda, golden_dir):
    """PhraseCut-shaped item (Hybridgl_main_PhraseCut.py:56-62,67-119): crop-layer proposals, one hybrid forward per
    image, every phrase scored against its own ground truth."""
    import torch
    from hybridgl_amd import refer_io, sam as hsam
    from hybridgl_amd.backbone import CLIPViTFM
    from hybridgl_amd.pipeline import HybridGLPipeline
    from hybridgl_amd.synth import synth_image
    from hybridgl_amd.tokenizer import SimpleTokenizer
    img = synth_image(150, 200, 3)
    tk = SimpleTokenizer(os.path.join(golden_dir, "tiny_bpe_vocab.txt.gz"))
    polys = [[[[(20, 20), (90, 25), (80, 100), (25, 90)]]],
             [[[(100, 30), (180, 30), (180, 120), (100, 120)]], [[(10, 110), (40, 110), (40, 140)]]]]
    ref = refer_io.phrasecut_item(img, ["the cat on left", "a big dog"], polys, cuda, tk,
                                  parse={"a big dog": {"noun_phrase": "dog", "relaflag": "big"}}, image_id=7)
    want = refer_io.phrasecut_polygons_to_mask([p for inst in polys[1] for p in inst], 200, 150)
    assert np.array_equal(ref.sentences[1].target.cpu().numpy().astype(bool), want) and want.sum() > 7000
    model = CLIPViTFM(model_name="ViT-B/16", device=cuda).eval()
    tiny = hsam.sam_model_registry["tiny"](device=cuda)
    gen = hsam.SamAutomaticMaskGenerator(tiny, points_per_side=4, pred_iou_thresh=0.0, stability_score_thresh=0.0,
                                         crop_n_layers=1, crop_n_points_downscale_factor=2, min_mask_region_area=10)
    pipe = HybridGLPipeline(model, fusion_mode="G2L&L2G", masking_block=9, mask_generator=gen, use_sam_masks=True)
    pipe.step(ref)
    m = pipe.metrics()
    assert m["n_sentences"] == 2 and m["cum"][1] > 0 and m["cum"][3] > 0
    # union >= the phrase's own ground-truth area for each sentence
    (a0, b0), (a1, b1) = [(x.cpu().numpy(), y.cpu().numpy()) for x, y in pipe.iu_log]
    assert int(a0[1]) >= int(ref.sentences[0].target.sum().item()) and int(a1[1]) >= int(want.sum())
