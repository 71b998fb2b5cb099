"""Pin-when-available hooks for the third-party arithmetic the reference calls but this image does not hold
(opencv-python 4.10.0.84, torchvision 0.15.2, gem_torch 1.0.1 over open_clip_torch 2.24.0; VERDICT r04
"what's missing" 2-3).  The oracle restates their published algorithms (oracle/cv_oracle.py, oracle/sam_oracle.py nms /
remove_small_regions, oracle/gem_oracle.py) and says "parity unpinned" where it does; this script turns that into ONE command
on any box that has the reference's environment.yaml installed:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_thirdparty_golden.py [--only cv2,tv,gem] [--out tests/golden]

For every package that imports it runs the REAL package on seeded inputs and writes a small fixture (inputs + outputs):

    cv2          -> cv_blur.npz   cv2.GaussianBlur(img, (15, 15), 0) on uint8 images (Hybridgl_main.py:99)
                    cv_comp.npz   the bitwise_and / add compositing of Hybridgl_main.py:103-113 on one image and masks
                    cv_cc.npz     cv2.connectedComponentsWithStats(mask, 8) -> remove_small_regions (utils/amg.py:267-291)
    torchvision  -> tv_nms.npz    torchvision.ops.boxes.batched_nms (automatic_mask_generator.py:251-257)
                    tv_resize.npz TF.resize on a float tensor (model/backbone.py:160), T.Resize(antialias=True) (Hybridgl_main.py:201)
    gem          -> gem_b16.npz   gem.create_gem_model('ViT-B/16', 'openai') heat-maps of a seeded image for three prompts
                                  (Hybridgl_main.py:36-39, 200-201); needs the OpenAI checkpoint the package downloads

tests/test_thirdparty_pins.py consumes whichever fixtures exist (the oracle on the CPU, the HIP path on the GPU) and skips the
rest with this file's name in the reason.  `--selftest DIR` writes fixtures of the same layout from the ORACLE instead (not
pins: plumbing only -- it lets the consumer tests run here, where none of the packages exists).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hybridgl_amd.synth import synth_image, synth_masks  # noqa: E402

BLUR_SIZES = [(97, 130), (480, 640), (33, 17)]
CC_CASES = [(0, 60, 80, 30), (1, 97, 131, 10), (2, 64, 64, 800)]       # (seed, H, W, area threshold)


def speckle(seed, H, W, n=4):
    """n masks [n,H,W] uint8: a blob with holes and islands, speckle at two densities, an empty one"""
    rng = np.random.default_rng(1000 + seed)
    m = np.zeros((n, H, W), dtype=np.uint8)
    m[0] = synth_masks(1, H, W, 7 + seed)[0]
    m[0][rng.random((H, W)) < 0.02] ^= 1
    m[1] = rng.random((H, W)) < 0.45
    m[2] = rng.random((H, W)) < 0.08
    return m


def nms_case(seed, n=200, n_idx=3):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, 500, (n, 2))
    wh = rng.uniform(5, 200, (n, 2))
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    boxes[::7] = boxes[1::7][: len(boxes[::7])]                # exact duplicates
    scores = rng.uniform(0, 1, n).astype(np.float32)
    scores[::11] = scores[1::11][: len(scores[::11])]          # tied scores
    idxs = rng.integers(0, n_idx, n).astype(np.int64)
    return boxes, scores, idxs


# ----------------------------------------------------------------------------- the real packages
def gen_cv2(out, cv2):
    blur = {}
    for i, (H, W) in enumerate(BLUR_SIZES):
        img = synth_image(H, W, 50 + i)
        blur[f"img{i}"] = img
        blur[f"out{i}"] = cv2.GaussianBlur(img, (15, 15), 0)
    blur["version"] = np.array(cv2.__version__)
    np.savez_compressed(os.path.join(out, "cv_blur.npz"), **blur)
    # Hybridgl_main.py:99-113: blurred background + sharp foreground, per mask
    H, W = 120, 160
    img = synth_image(H, W, 60)
    masks = synth_masks(3, H, W, 61).astype(np.uint8)
    blurred = cv2.GaussianBlur(img, (15, 15), 0)
    comp = []
    for m in masks:
        m255 = (m * 255).astype(np.uint8)
        fg = cv2.bitwise_and(img, img, mask=m255)
        bg = cv2.bitwise_and(blurred, blurred, mask=cv2.bitwise_not(m255))
        comp.append(cv2.add(fg, bg))
    np.savez_compressed(os.path.join(out, "cv_comp.npz"), img=img, masks=masks, blurred=blurred, out=np.stack(comp),
                        version=np.array(cv2.__version__))
    cc = {}
    for ci, (seed, H, W, thr) in enumerate(CC_CASES):
        m = speckle(seed, H, W)
        cc[f"m{ci}"] = m
        cc[f"thr{ci}"] = np.array(thr)
        for mode in ("holes", "islands"):
            outs, changed, ncomp = [], [], []
            for k in range(len(m)):
                # utils/amg.py:267-291, statement for statement, on the real cv2
                mask = m[k].astype(bool)
                correct_holes = mode == "holes"
                working = (correct_holes ^ mask).astype(np.uint8)
                n_labels, regions, stats, _ = cv2.connectedComponentsWithStats(working, 8)
                sizes = stats[:, -1][1:]
                small = [i + 1 for i, s in enumerate(sizes) if s < thr]
                ncomp.append(n_labels - 1)
                if len(small) == 0:
                    outs.append(mask)
                    changed.append(False)
                    continue
                fill = [0] + small
                if not correct_holes:
                    fill = [i for i in range(n_labels) if i not in fill]
                    if len(fill) == 0:
                        fill = [int(np.argmax(sizes)) + 1]
                outs.append(np.isin(regions, fill))
                changed.append(True)
            cc[f"out{ci}_{mode}"] = np.stack(outs).astype(np.uint8)
            cc[f"changed{ci}_{mode}"] = np.array(changed)
            cc[f"ncomp{ci}_{mode}"] = np.array(ncomp)
    cc["version"] = np.array(cv2.__version__)
    np.savez_compressed(os.path.join(out, "cv_cc.npz"), **cc)
    print("cv2", cv2.__version__, "-> cv_blur.npz, cv_comp.npz, cv_cc.npz")


def gen_tv(out, torchvision):
    import torch
    from torchvision.ops.boxes import batched_nms
    import torchvision.transforms as T
    import torchvision.transforms.functional as TF
    d = {}
    for ci, thr in enumerate((0.7, 0.3, 0.95)):
        boxes, scores, idxs = nms_case(ci)
        keep = batched_nms(torch.from_numpy(boxes), torch.from_numpy(scores), torch.from_numpy(idxs), thr)
        d[f"boxes{ci}"], d[f"scores{ci}"], d[f"idxs{ci}"], d[f"thr{ci}"], d[f"keep{ci}"] = boxes, scores, idxs, np.array(thr), keep.numpy()
    d["version"] = np.array(torchvision.__version__)
    np.savez_compressed(os.path.join(out, "tv_nms.npz"), **d)
    rng = np.random.default_rng(5)
    r = {}
    for ci, (H, W, oh, ow) in enumerate([(97, 130, 14, 14), (640, 480, 14, 14), (28, 28, 97, 130)]):
        x = rng.random((3, H, W)).astype(np.float32)
        r[f"x{ci}"] = x
        r[f"plain{ci}"] = TF.resize(torch.from_numpy(x), (oh, ow)).numpy()                       # model/backbone.py:160
        r[f"aa{ci}"] = T.Resize((oh, ow), antialias=True)(torch.from_numpy(x)).numpy()            # Hybridgl_main.py:201
    r["version"] = np.array(torchvision.__version__)
    np.savez_compressed(os.path.join(out, "tv_resize.npz"), **r)
    print("torchvision", torchvision.__version__, "-> tv_nms.npz, tv_resize.npz")


GEM_PROMPTS = ["cat", "left dog", "the red car"]


def gen_gem(out, gem):
    import torch
    from PIL import Image
    model = gem.create_gem_model(model_name="ViT-B/16", pretrained="openai", device="cpu")        # Hybridgl_main.py:36-38
    tf = gem.get_gem_img_transform()
    img = synth_image(375, 500, 77)
    x = tf(Image.fromarray(img)).unsqueeze(0)
    with torch.no_grad():
        heat = model(x, GEM_PROMPTS)                                                                # [1, n_prompts, h, w]
    import hashlib
    sd = model.model.state_dict() if hasattr(model, "model") else model.state_dict()
    dig = hashlib.sha256()
    for k in sorted(sd):
        dig.update(k.encode())
        dig.update(sd[k].detach().cpu().float().numpy().tobytes()[:4096])
    np.savez_compressed(os.path.join(out, "gem_b16.npz"), img=img, tensor_img=x.numpy(), prompts=np.array(GEM_PROMPTS),
                        heat=heat.numpy(), weights_digest=np.array(dig.hexdigest()), version=np.array(getattr(gem, "__version__", "?")))
    print("gem -> gem_b16.npz (weights digest", dig.hexdigest()[:12], ")")


# ----------------------------------------------------------------------------- plumbing self-test (NOT pins)
def selftest(out):
    """the same files from the ORACLE: lets tests/test_thirdparty_pins.py exercise its consumers where no package exists"""
    import torch
    from oracle import cv_oracle as CV
    from oracle import sam_oracle as S
    from oracle import gem_oracle as GO
    from oracle import clip_oracle as O
    blur = {}
    for i, (H, W) in enumerate(BLUR_SIZES):
        img = synth_image(H, W, 50 + i)
        blur[f"img{i}"], blur[f"out{i}"] = img, CV.gaussian_blur_u8(img, 15)
    blur["version"] = np.array("oracle-selftest")
    np.savez_compressed(os.path.join(out, "cv_blur.npz"), **blur)
    cc = {}
    for ci, (seed, H, W, thr) in enumerate(CC_CASES):
        m = speckle(seed, H, W)
        cc[f"m{ci}"], cc[f"thr{ci}"] = m, np.array(thr)
        for mode in ("holes", "islands"):
            res = [S.remove_small_regions(m[k].astype(bool), thr, mode) for k in range(len(m))]
            cc[f"out{ci}_{mode}"] = np.stack([r[0] for r in res]).astype(np.uint8)
            cc[f"changed{ci}_{mode}"] = np.array([r[1] for r in res])
    cc["version"] = np.array("oracle-selftest")
    np.savez_compressed(os.path.join(out, "cv_cc.npz"), **cc)
    d = {}
    for ci, thr in enumerate((0.7, 0.3, 0.95)):
        boxes, scores, idxs = nms_case(ci)
        off = idxs.astype(np.float32)[:, None] * (boxes.max() + 1)          # batched_nms: per-class offsets
        keep = S.nms(boxes + off, scores, thr)
        d[f"boxes{ci}"], d[f"scores{ci}"], d[f"idxs{ci}"], d[f"thr{ci}"], d[f"keep{ci}"] = boxes, scores, idxs, np.array(thr), np.asarray(keep)
    d["version"] = np.array("oracle-selftest")
    np.savez_compressed(os.path.join(out, "tv_nms.npz"), **d)
    rng = np.random.default_rng(5)
    r = {}
    for ci, (H, W, oh, ow) in enumerate([(97, 130, 14, 14), (640, 480, 14, 14), (28, 28, 97, 130)]):
        x = rng.random((3, H, W)).astype(np.float32)
        r[f"x{ci}"], r[f"plain{ci}"], r[f"aa{ci}"] = x, O.bilinear_resize(x, oh, ow), GO.resize_bilinear_aa(x, oh, ow)
    r["version"] = np.array("oracle-selftest")
    np.savez_compressed(os.path.join(out, "tv_resize.npz"), **r)
    print("selftest fixtures (from the oracle, not pins) ->", out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="cv2,tv,gem")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--selftest", default="", help="write fixtures of the same layout from the oracle into this directory (plumbing only)")
    a = ap.parse_args()
    if a.selftest:
        os.makedirs(a.selftest, exist_ok=True)
        selftest(a.selftest)
        sys.exit(0)
    os.makedirs(a.out, exist_ok=True)
    done = 0
    for name in a.only.split(","):
        try:
            if name == "cv2":
                import cv2
                gen_cv2(a.out, cv2)
            elif name == "tv":
                import torchvision
                gen_tv(a.out, torchvision)
            elif name == "gem":
                import gem
                gen_gem(a.out, gem)
            else:
                print("unknown:", name)
                continue
            done += 1
        except ImportError as e:
            print(f"{name}: not importable here ({e}); its fixtures stay absent and the consumer tests skip")
    print(f"{done} package(s) pinned")
