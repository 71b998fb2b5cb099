"""Consumers of the pin-when-available fixtures (oracle/gen_thirdparty_golden.py): outputs of the REAL opencv-python /
torchvision / gem_torch packages on seeded inputs.  None of the three is installed in the build image, so the fixtures are
normally absent and these tests skip, saying which command creates them; on a box with the reference's environment.yaml one
command turns the "parity unpinned" rows (cv2 blur and connected components, torchvision NMS and
resize, the GEM heat-map) into pinned ones.  `test_consumers_run_on_selftest_fixtures` exercises the same consumer code on
fixtures written from the oracle (plumbing only: it pins nothing)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import clip_oracle as O
from oracle import cv_oracle as CV
from oracle import gem_oracle as GO
from oracle import sam_oracle as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOW = "absent: run `python oracle/gen_thirdparty_golden.py` where {} is installed (the reference's environment.yaml)"


def fixture(golden_dir, name, pkg):
    p = os.path.join(golden_dir, name)
    if not os.path.exists(p):
        pytest.skip(f"tests/golden/{name} " + HOW.format(pkg))
    return np.load(p)


# ---- the consumers (also used by the GPU tests below) ----
def check_cv_blur(g, blur):
    n = 0
    while f"img{n}" in g:
        assert np.array_equal(blur(g[f"img{n}"]), g[f"out{n}"]), f"cv2.GaussianBlur case {n}"
        n += 1
    assert n >= 1


def check_cv_cc(g, clean):
    n = 0
    while f"m{n}" in g:
        m, thr = g[f"m{n}"], int(g[f"thr{n}"])
        for mode in ("holes", "islands"):
            out, changed = clean(m, thr, mode)
            assert np.array_equal(out.astype(np.uint8), g[f"out{n}_{mode}"]), f"remove_small_regions case {n} {mode}"
            assert np.array_equal(np.asarray(changed, dtype=bool), g[f"changed{n}_{mode}"].astype(bool))
        n += 1
    assert n >= 1


def check_tv_nms(g, nms):
    n = 0
    while f"boxes{n}" in g:
        keep = nms(g[f"boxes{n}"], g[f"scores{n}"], g[f"idxs{n}"], float(g[f"thr{n}"]))
        assert np.array_equal(np.asarray(keep, dtype=np.int64), g[f"keep{n}"]), f"batched_nms case {n}"
        n += 1
    assert n >= 1


def oracle_clean(m, thr, mode):
    res = [S.remove_small_regions(m[k].astype(bool), thr, mode) for k in range(len(m))]
    return np.stack([r[0] for r in res]), [r[1] for r in res]


def oracle_batched_nms(boxes, scores, idxs, thr):
    off = idxs.astype(np.float32)[:, None] * (boxes.max() + 1)      # torchvision's batched_nms: per-class coordinate offsets
    return S.nms(boxes + off, scores, thr)


# ---- CPU: the oracle against the real packages ----
def test_oracle_blur_equals_cv2(golden_dir):
    check_cv_blur(fixture(golden_dir, "cv_blur.npz", "opencv-python"), lambda img: CV.gaussian_blur_u8(img, 15))


def test_oracle_remove_small_regions_equals_cv2(golden_dir):
    check_cv_cc(fixture(golden_dir, "cv_cc.npz", "opencv-python"), oracle_clean)


def test_oracle_nms_equals_torchvision(golden_dir):
    check_tv_nms(fixture(golden_dir, "tv_nms.npz", "torchvision"), oracle_batched_nms)


def test_oracle_resize_equals_torchvision(golden_dir):
    g = fixture(golden_dir, "tv_resize.npz", "torchvision")
    n = 0
    while f"x{n}" in g:
        x, plain, aa = g[f"x{n}"], g[f"plain{n}"], g[f"aa{n}"]
        np.testing.assert_allclose(O.bilinear_resize(x, *plain.shape[-2:]), plain, rtol=0, atol=1e-6)
        np.testing.assert_allclose(GO.resize_bilinear_aa(x, *aa.shape[-2:]), aa, rtol=0, atol=2e-6)
        n += 1
    assert n >= 1


def test_oracle_gem_heatmap_equals_gem_torch(golden_dir):
    g = fixture(golden_dir, "gem_b16.npz", "gem_torch (+ the OpenAI ViT-B/16 checkpoint)")
    ckpt = os.environ.get("HYBRIDGL_CLIP_CKPT", "")
    if not os.path.exists(ckpt):
        pytest.skip("gem_b16.npz is present but HYBRIDGL_CLIP_CKPT does not name the OpenAI ViT-B-16.pt the fixture was made with")
    from hybridgl_amd.backbone import load_clip_state_dict
    sd = {k: v.float().numpy() for k, v in load_clip_state_dict(ckpt).items()}
    feat, _ = GO.gem_vit_forward(sd, g["tensor_img"])
    from hybridgl_amd.tokenizer import SimpleTokenizer, tokenize
    tok = tokenize([f"a photo of a {p}." for p in g["prompts"]], tokenizer=SimpleTokenizer(os.environ.get("HYBRIDGL_BPE_VOCAB") or None))
    text = O.encode_text(sd, tok)
    heat = GO.gem_heatmap(feat[0], text, g["tensor_img"].shape[-1])
    np.testing.assert_allclose(heat, g["heat"][0], rtol=0, atol=2e-3)


# ---- GPU: the HIP path against the real packages ----
@pytest.mark.gpu
def test_gpu_blur_equals_cv2(cuda, golden_dir):
    import torch
    from hybridgl_amd import ops
    g = fixture(golden_dir, "cv_blur.npz", "opencv-python")
    check_cv_blur(g, lambda img: ops.gaussian_blur_u8(torch.from_numpy(img).to(cuda), 15).cpu().numpy())


@pytest.mark.gpu
def test_gpu_remove_small_regions_equals_cv2(cuda, golden_dir):
    import torch
    from hybridgl_amd import sam as hsam
    g = fixture(golden_dir, "cv_cc.npz", "opencv-python")

    def clean(m, thr, mode):
        out, ch = hsam.remove_small_regions(torch.from_numpy(m).to(cuda), thr, mode)
        return out.cpu().numpy(), ch.cpu().numpy().astype(bool)
    check_cv_cc(g, clean)


@pytest.mark.gpu
def test_gpu_nms_equals_torchvision(cuda, golden_dir):
    import torch
    from hybridgl_amd import sam as hsam
    g = fixture(golden_dir, "tv_nms.npz", "torchvision")

    def nms(boxes, scores, idxs, thr):
        off = idxs.astype(np.float32)[:, None] * (boxes.max() + 1)
        b = torch.from_numpy((boxes + off).astype(np.float32)).to(cuda)
        order, n = hsam.nms_large(b, torch.from_numpy(scores).to(cuda), torch.ones(len(b), dtype=torch.uint8, device=cuda), thr)
        return order[: int(n.item())].cpu().numpy()
    check_tv_nms(g, nms)


# ---- plumbing: the consumers above on fixtures of the same layout written from the oracle ----
def test_consumers_run_on_selftest_fixtures(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_thirdparty_golden.py"), "--selftest", str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    check_cv_blur(np.load(tmp_path / "cv_blur.npz"), lambda img: CV.gaussian_blur_u8(img, 15))
    check_cv_cc(np.load(tmp_path / "cv_cc.npz"), oracle_clean)
    check_tv_nms(np.load(tmp_path / "tv_nms.npz"), oracle_batched_nms)
    # without the packages the generator pins nothing and says so
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_thirdparty_golden.py"), "--out", str(tmp_path / "real")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "package(s) pinned" in r.stdout
