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
