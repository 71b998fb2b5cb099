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


def test_driver_runs_refer_data_end_to_end(cuda, golden_dir, tmp_path):
    from hybridgl_amd import main as drv
    root = tmp_path / "refer_data"
    refs = _dataset(root)
    args = drv.default_argument_parser().parse_args([
        "--real", "--refer_data_root", str(root), "--dataset", "refcoco", "--split", "val", "--sam_model", "tiny",
        "--bpe_vocab", os.path.join(golden_dir, "tiny_bpe_vocab.txt.gz"), "--parse_json", str(root / "parse.json"),
        "--pred_iou_thresh", "-1", "--stability_score_thresh", "0", "--min_mask_region_area", "20",
        "--points_per_side", "4", "--result_dir", str(tmp_path / "log")])
    m = drv.main(args)
    assert m["n_sentences"] == 2 * len(refs)
    assert m["cum"][1] > 0 and 0.0 <= m["oIoU"] <= 100.0 and 0.0 <= m["oIoU_final"] <= 100.0
    log = open(tmp_path / "log" / "result_log_refcoco_val.txt").read()
    assert "pure hybridgl:" in log and "hybridgl w/ spatial guidance:" in log and "refcoco / val / unc" in log
