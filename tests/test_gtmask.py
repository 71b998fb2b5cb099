"""Ground-truth mask codec + REFER reader.  Host code: runs without a GPU.

hgl_gt_mask_* (C++ in libhybridgl.so) and the Python oracle are checked against vectors produced by the
reference's own maskApi.c (tests/golden/gtmask.npz, oracle/gen_gtmask_golden.py) and, where the compiled
reference is present (oracle/_ref), against it directly on random polygons.  Bar: bit-exact."""
import json
import os
import pickle

import numpy as np
import pytest

from hybridgl_amd import refer_io
from oracle import gtmask_oracle as G


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "gtmask.npz"))


def _polys(gold, i):
    xy, npts = gold[f"c{i}_xy"], gold[f"c{i}_npts"]
    out, p = [], 0
    for k in npts:
        out.append(xy[p:p + 2 * k].tolist())
        p += 2 * k
    return out


def test_polygon_masks_match_reference_vectors(gold):
    for i in range(int(gold["n_cases"][0])):
        H, W = (int(v) for v in gold[f"c{i}_size"])
        m, area = refer_io.gt_mask_from_polygons(_polys(gold, i), H, W)
        assert np.array_equal(m, gold[f"c{i}_mask"]), i
        assert area == int(gold[f"c{i}_area"][0]), i


def test_oracle_matches_reference_vectors(gold):
    for i in range(int(gold["n_cases"][0])):
        H, W = (int(v) for v in gold[f"c{i}_size"])
        if H * W > 40000:
            continue      # pure-Python loops: small cases only
        m, area = G.poly_to_mask(_polys(gold, i), H, W)
        assert np.array_equal(m, gold[f"c{i}_mask"]) and area == int(gold[f"c{i}_area"][0]), i


def test_rle_counts_and_strings(gold):
    strs = [str(s) for s in gold["r_strings"]]
    for j in range(int(gold["n_rle"][0])):
        H, W = (int(v) for v in gold[f"r{j}_size"])
        want = gold[f"r{j}_mask"]
        m, area = refer_io.gt_mask_from_rle({"size": [H, W], "counts": gold[f"r{j}_counts"].tolist()})
        assert np.array_equal(m, want) and area == int(want.sum())
        m2, area2 = refer_io.gt_mask_from_rle({"size": [H, W], "counts": strs[j]})
        assert np.array_equal(m2, want) and area2 == int(want.sum())
        assert G.rle_string_to_counts(strs[j]) == gold[f"r{j}_counts"].tolist()
        assert np.array_equal(G.counts_to_mask(gold[f"r{j}_counts"].tolist(), H, W), want)


def test_overlapping_polygons_count_twice(gold):
    """REFER.getMask sums the polygons' masks; the dataset keeps count == 1 (dataset_refer_bert.py:118-121)."""
    m, _ = refer_io.gt_mask_from_polygons([[2, 2, 20, 2, 20, 15, 2, 15], [10, 8, 35, 8, 35, 28, 10, 28]], 30, 40)
    assert m.max() == 2 and (m == 2).sum() > 0


@pytest.mark.skipif(not G.have_ref(), reason="oracle/_ref not built (needs /root/reference at build time)")
def test_fuzz_against_compiled_reference():
    ref = G.RefMaskApi()
    rng = np.random.default_rng(5)
    for t in range(400):
        H, W = int(rng.integers(3, 200)), int(rng.integers(3, 200))
        polys = []
        for _ in range(int(rng.integers(1, 4))):
            k = int(rng.integers(1, 14))
            xy = rng.random(2 * k) * np.tile([W + 20, H + 20], k) - 10
            if t % 3 == 0:
                xy = np.round(xy * 2) / 2
            polys.append(xy.tolist())
        want, wa = ref.poly_to_mask(polys, H, W)
        got, ga = refer_io.gt_mask_from_polygons(polys, H, W)
        assert np.array_equal(got, want) and ga == wa, t
    for t in range(50):
        H, W = int(rng.integers(2, 120)), int(rng.integers(2, 120))
        m = (rng.random((H, W)) < rng.random()).astype(np.uint8)
        s, cnts = ref.encode_to_string(m)
        assert np.array_equal(refer_io.gt_mask_from_rle({"size": [H, W], "counts": s})[0], m)
        assert np.array_equal(refer_io.gt_mask_from_rle({"size": [H, W], "counts": cnts})[0], m)


def test_rle_output_modes_against_reference_vectors(gold):
    """SamAutomaticMaskGenerator's output modes uncompressed_rle / coco_rle (automatic_mask_generator.py:176-182,
    utils/amg.py:107-153,294-300): mask -> counts -> compressed string, against counts and strings the reference's
    maskApi.c produced (tests/golden/gtmask.npz), and back."""
    from hybridgl_amd import sam as hsam
    strs = [str(s) for s in gold["r_strings"]]
    for j in range(int(gold["n_rle"][0])):
        H, W = (int(v) for v in gold[f"r{j}_size"])
        mask = gold[f"r{j}_mask"].astype(bool)
        rle = hsam.mask_to_rle(mask)
        assert rle == {"size": [H, W], "counts": gold[f"r{j}_counts"].tolist()}
        assert hsam.area_from_rle(rle) == int(mask.sum())
        assert np.array_equal(hsam.rle_to_mask(rle), mask)
        assert hsam.coco_encode_rle(rle) == {"size": [H, W], "counts": strs[j]}
    # masks that start with foreground carry a leading 0 count (utils/amg.py:131); all-zero / all-one masks
    m = np.ones((3, 2), bool)
    assert hsam.mask_to_rle(m)["counts"] == [0, 6] and hsam.mask_to_rle(~m)["counts"] == [6]
    m[1, 0] = False
    assert hsam.mask_to_rle(m)["counts"] == [0, 1, 1, 4]


@pytest.mark.skipif(not G.have_ref(), reason="oracle/_ref not built (needs /root/reference at build time)")
def test_rle_encoding_fuzz_against_compiled_reference():
    """hgl_rle_encode_mask / hgl_rle_to_string == rleEncode / rleToString of the reference's maskApi.c, incl. long runs
    (negative differences between counts two places apart: the sign digit of the string format)"""
    from hybridgl_amd import sam as hsam
    ref = G.RefMaskApi()
    rng = np.random.default_rng(8)
    for t in range(120):
        H, W = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        if t % 4 == 0:       # blobs: long runs
            m = np.zeros((H, W), np.uint8)
            for _ in range(int(rng.integers(0, 5))):
                y0, x0 = int(rng.integers(0, H)), int(rng.integers(0, W))
                m[y0:y0 + int(rng.integers(1, H + 1)), x0:x0 + int(rng.integers(1, W + 1))] = 1
        else:
            m = (rng.random((H, W)) < rng.random() ** 2).astype(np.uint8)
        s, cnts = ref.encode_to_string(m)
        rle = hsam.mask_to_rle(m)
        assert rle["counts"] == cnts, t
        assert hsam.coco_encode_rle(rle)["counts"] == s, t


def test_error_reporting():
    from hybridgl_amd._lib import HybridGLError
    with pytest.raises(HybridGLError):
        refer_io.gt_mask_from_polygons([[1.0, 2.0]], 0, 5)


def test_refer_reader_on_a_synthetic_dataset(tmp_path):
    """refs(unc).p + instances.json + images laid out as the reference expects: split filters, index tables,
    item layout, GT = pixels covered by exactly one polygon."""
    from PIL import Image
    root = tmp_path / "refer_data"
    (root / "refcoco").mkdir(parents=True)
    img_dir = root / "images/mscoco/images/train2014"
    img_dir.mkdir(parents=True)
    rng = np.random.default_rng(0)
    images, anns, refs = [], [], []
    for i in range(3):
        h, w = 40 + 8 * i, 60 + 4 * i
        name = f"COCO_train2014_{i:012d}.jpg"
        Image.fromarray(rng.integers(0, 255, size=(h, w, 3), dtype=np.uint8)).save(img_dir / name.replace(".jpg", ".png"))
        images.append({"id": 100 + i, "file_name": name.replace(".jpg", ".png"), "height": h, "width": w})
        seg = [[5, 5, 30, 6, 28, 25, 6, 24]] if i != 1 else [[5, 5, 30, 5, 30, 25, 5, 25], [20, 15, 45, 15, 45, 35, 20, 35]]
        anns.append({"id": 500 + i, "image_id": 100 + i, "category_id": 1 + (i % 2), "segmentation": seg, "bbox": [5, 5, 25, 20]})
        refs.append({"ref_id": 900 + i, "ann_id": 500 + i, "image_id": 100 + i, "category_id": 1 + (i % 2),
                     "split": ["val", "testA", "testB"][i], "sent_ids": [2 * i, 2 * i + 1],
                     "sentences": [{"sent_id": 2 * i, "raw": f"the left thing {i}", "tokens": ["the", "left", "thing"]},
                                   {"sent_id": 2 * i + 1, "raw": f"object number {i}", "tokens": ["object", "number"]}]})
    rle_m = np.zeros((12, 10), np.uint8)
    rle_m[3:9, 2:7] = 1
    anns.append({"id": 777, "image_id": 100, "category_id": 2,
                 "segmentation": {"size": [12, 10], "counts": [int(v) for v in _counts(rle_m)]}, "bbox": [2, 3, 5, 6]})
    json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "person"}, {"id": 2, "name": "dog"}]},
              open(root / "refcoco/instances.json", "w"))
    pickle.dump(refs, open(root / "refcoco/refs(unc).p", "wb"))

    R = refer_io.REFER(str(root), "refcoco", "unc")
    assert R.getRefIds(split="val") == [900] and R.getRefIds(split="testA") == [901] and R.getRefIds(split="test") == [901, 902]
    assert R.getImgIds([901]) == [101] and R.Cats[2] == "dog" and R.sentToRef[3]["ref_id"] == 901
    assert R.refToAnn[902]["id"] == 502 and [r["ref_id"] for r in R.imgToRefs[100]] == [900]
    ds = refer_io.ReferDataset(str(root), "refcoco", "unc", split="testA")
    data, annot, sents = ds[0]
    assert sents == ["the left thing 1", "object number 1"] and data["cat_name"] == "dog"
    assert data["sam_img"].shape == (48, 64, 3) and annot.shape == (48, 64) and annot.dtype == np.uint8
    full = R.getMask(R.Refs[901])
    assert full["mask"].max() == 2 and np.array_equal(annot, (full["mask"] == 1).astype(np.uint8))
    assert full["area"] == int((full["mask"] >= 1).sum() + (full["mask"] == 2).sum())
    # RLE-annotated object
    R.refToAnn[900] = R.Anns[777]
    R.Imgs[100] = dict(R.Imgs[100], height=12, width=10)
    got = R.getMask(R.Refs[900])
    assert np.array_equal(got["mask"], rle_m) and got["area"] == int(rle_m.sum())


def _counts(mask):
    flat = mask.T.ravel()
    cnts, cur, run = [], 0, 0
    for v in flat:
        if v == cur:
            run += 1
        else:
            cnts.append(run)
            cur, run = v, 1
    cnts.append(run)
    return cnts
