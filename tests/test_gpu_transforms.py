"""hybridgl_amd/transforms.py: the reference's per-item dataset transforms on the device -- ToTensor + Normalize
(data/dataset_refer_bert.py:155-158) and gem.get_gem_img_transform (Hybridgl_main.py:39: bicubic Resize((448, 448)) +
ToTensor + Normalize) -- must be the host transforms BIT FOR BIT (Pillow's resampler itself and numpy's fp32 arithmetic
are the checkers here), and the evaluator's loader must hand the loop the same tensors with them as without."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hw", [(480, 640), (427, 640), (375, 500), (640, 640), (97, 131), (448, 448), (1000, 900), (33, 2000)])
def test_device_transforms_are_the_host_transforms_bit_for_bit(cuda, hw):
    from PIL import Image
    from hybridgl_amd import synth, transforms as T
    from hybridgl_amd.gem import get_gem_img_transform
    rng = np.random.default_rng(hw[0] * 7 + hw[1])
    img = rng.integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
    d = torch.from_numpy(img).to(cuda)
    assert np.array_equal(T.to_tensor_normalize(d).cpu().numpy(), synth.imagenet_normalize(img))
    for filt, pf in (("bicubic", Image.BICUBIC), ("bilinear", Image.BILINEAR)):
        want = np.asarray(Image.fromarray(img).resize((448, 448), pf))
        assert np.array_equal(T.pil_resize_u8(d, 448, 448, filt).cpu().numpy(), want), filt
    want = np.asarray(Image.fromarray(img).resize((211, 307), Image.BICUBIC))      # up- and down-scaling axes at once
    assert np.array_equal(T.pil_resize_u8(d, 307, 211, "bicubic").cpu().numpy(), want)
    assert np.array_equal(T.gem_img_transform(d).cpu().numpy(), get_gem_img_transform()(img).numpy())


def test_loader_items_are_identical_with_device_and_host_transforms(cuda, golden_dir, tmp_path):
    """RealRefs.load with the transforms on the device == with the reference's host transforms: every tensor of every
    item, and the refs of one image share one decode / upload."""
    from hybridgl_amd import main as drv, synth
    from hybridgl_amd.loader import Prefetcher
    root = str(tmp_path / "tree")
    info = synth.write_refer_tree(root, n_images=6, sizes=((120, 160), (160, 120), (97, 131)))
    args = drv.default_argument_parser().parse_args([
        "--real", "--refer_data_root", root, "--dataset", "refcoco", "--split", "val", "--bpe_vocab", os.path.join(root, "bpe.txt.gz"),
        "--parse_json", os.path.join(root, "parse.json")])
    a = drv.RealRefs(args, cuda, "unc", 77, device_transforms=True)
    b = drv.RealRefs(args, cuda, "unc", 77, device_transforms=False, image_lru=0)
    items_a = list(Prefetcher(a.jobs(), a.load, workers=4, depth=8, device=cuda))
    items_b = [b.load(i) for i in b.jobs()]
    torch.cuda.synchronize()
    assert len(items_a) == len(items_b) == info["refs"]
    assert a.decoded == info["images"] and b.decoded == info["refs"]
    for x, y in zip(items_a, items_b):
        for f in ("sam_img", "image_norm", "tensor_img", "tokens", "target"):
            assert torch.equal(getattr(x, f), getattr(y, f)), f
        assert x.image_id == y.image_id and x.index == y.index and x.token_len == y.token_len
        assert [(s.sentence_row, s.noun_phrase_row, s.other_noun_rows, s.dirflag, s.relaflag, s.gem_row) for s in x.sentences] == \
               [(s.sentence_row, s.noun_phrase_row, s.other_noun_rows, s.dirflag, s.relaflag, s.gem_row) for s in y.sentences]


def test_evaluate_from_disk_reports_loader_statistics(cuda, golden_dir, tmp_path):
    """hybridgl_amd.main.evaluate on a tree written by synth.write_refer_tree (tiny SAM): every ref is scored, the report
    equals the one of the same items fed resident, the loader statistics are filled in."""
    from hybridgl_amd import main as drv, synth
    from hybridgl_amd.pipeline import HybridGLPipeline
    root = str(tmp_path / "tree")
    info = synth.write_refer_tree(root, n_images=10, sizes=((120, 160), (160, 120), (97, 131)), far_refs=0.4)
    args = drv.default_argument_parser().parse_args([
        "--real", "--refer_data_root", root, "--dataset", "refcoco", "--split", "val", "--bpe_vocab", os.path.join(root, "bpe.txt.gz"),
        "--parse_json", os.path.join(root, "parse.json"), "--sam_model", "tiny", "--points_per_side", "4", "--pred_iou_thresh", "-1",
        "--stability_score_thresh", "0", "--min_mask_region_area", "20", "--group", "4", "--proposal_cap", "12"])
    model, gen, gem = drv.build_models(args, cuda)
    m, st = drv.evaluate(args, model, gen, gem, cuda)
    assert st["refs"] == info["refs"] and m["n_sentences"] == info["sentences"] and st["skipped"] == 0
    assert st["images_decoded"] <= info["refs"] and st["image_cache_hits"] > 0 and st["groups"] >= 2
    assert st["seconds"] > 0 and st["loader_make_s"] > 0 and 0 <= st["loader_wait_s"] <= st["seconds"]
    rr = drv.RealRefs(args, cuda, "unc", 77)
    pipe = HybridGLPipeline(model, mask_generator=gen, use_sam_masks=True, gem_model=gem)
    pipe.run((rr.load(i) for i in rr.jobs()), group=4, proposal_cap=12)
    assert pipe.metrics() == m


@pytest.mark.parametrize("hw,out", [((800, 1066), (480, 640)), ((600, 800), (600, 800)), ((97, 131), (200, 333)), ((1066, 800), (500, 375))])
def test_plain_bilinear_tensor_resize_vs_oracle(cuda, hw, out):
    """hgl_resize_bilinear == F.interpolate(bilinear, align_corners=False, antialias=False) as the oracle restates it (pinned
    against torch in tests/golden/resize.npz): T.Resize on a TENSOR, Hybridgl_main_PhraseCut.py:69-70.  Bit for bit."""
    from hybridgl_amd import transforms as T
    from oracle import clip_oracle as O
    x = np.random.default_rng(hw[0] + out[1]).standard_normal((3,) + hw).astype(np.float32)
    got = T.resize_bilinear(torch.from_numpy(x).to(cuda), out).cpu().numpy()
    assert got.shape == (3,) + out and np.array_equal(got, O.bilinear_resize(x, out[0], out[1]))
    import torch.nn.functional as F
    want = F.interpolate(torch.from_numpy(x)[None], size=out, mode="bilinear", align_corners=False)[0].numpy()
    assert np.abs(got - want).max() < 1e-5          # torch's CPU kernel contracts some of the products into fmas


@pytest.mark.parametrize("file_hw,annot_hw", [((480, 640), (480, 640)), ((360, 480), (480, 640)), ((640, 427), (640, 427))])
def test_phrasecut_image_norm_vs_host_chain(cuda, file_hw, annot_hw):
    """image['image'] of the PhraseCut loop (data/dataset_phrasecut.py:49-51 + Hybridgl_main_PhraseCut.py:69-70): T.Resize(800) on
    the PIL image -> ToTensor -> Normalize -> T.Resize((height, width)) on the tensor, all on the device, against the same
    chain on the host (Pillow's own resize, numpy normalisation, the oracle's bilinear): bit for bit."""
    from PIL import Image
    from hybridgl_amd import synth, transforms as T
    from oracle import clip_oracle as O
    img = np.random.default_rng(file_hw[0]).integers(0, 256, file_hw + (3,), dtype=np.uint8)
    nh, nw = T.resize_shorter_side(file_hw[0], file_hw[1], 800)
    assert min(nh, nw) == 800
    host = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    want = O.bilinear_resize(synth.imagenet_normalize(host), annot_hw[0], annot_hw[1])
    got = T.phrasecut_image_norm(torch.from_numpy(img).to(cuda), annot_hw[0], annot_hw[1]).cpu().numpy()
    assert got.shape == (3,) + annot_hw and np.array_equal(got, want)
