"""VGPhraseCut annotations -> the inputs of the hot path (host side).

The reference's PhraseCut evaluator (Hybridgl_main_PhraseCut.py:40-43, data/dataset_phrasecut.py:9-122) reads the dataset
through `PhraseCutDataset.utils.refvg_loader.RefVGLoader` -- the loader of the published VGPhraseCut release, an EMPTY
git submodule in the reference tree.  What is restated here is (a) the part of that loader `dataset_phrasecut.py` calls --
`RefVGLoader(split).img_ids`, `.get_img_ref_data(image_id)` with the keys it reads (`image_id`, `width`, `height`,
`task_ids`, `phrases`, `gt_Polygons`, `gt_boxes`, `img_ins_cats`) -- over the published file layout

    <root>/image_data_split.json   [{"image_id", "width", "height", "split", ...}, ...]
    <root>/refer_<split>.json      [{"task_id", "image_id", "phrase", "phrase_structure": {"name", ...},
                                     "instance_boxes": [[x, y, w, h], ...], "Polygons": [[[[x, y], ...], ...], ...]}, ...]
    <root>/images/<image_id>.jpg

and (b) the item layout of `PhraseCutDataset.__getitem__` (:36-104): one item per IMAGE with all its phrases, the image
resized to the annotation's (width, height) (`cv2.resize`, :55 -- a no-op for every image whose file has the size the
annotation states), per-phrase ground truth from the instance polygons rasterised with Pillow (:108-122), the seen / unseen
category filters (:63-66).  The loader package itself is absent, so (a) is pinned by this file-layout description only.
"""
import json
import os

import numpy as np

from .refer_io import phrasecut_polygons_to_mask

# data/dataset_phrasecut.py:15-28
COCO_CLASSES = ['person', 'bicycle', 'car', 'motorcycle', 'airplane', 'bus', 'train', 'truck', 'boat', 'traffic light',
                'fire hydrant', 'stop sign', 'parking meter', 'bench', 'bird', 'cat', 'dog', 'horse', 'sheep', 'cow', 'elephant',
                'bear', 'zebra', 'giraffe', 'backpack', 'umbrella', 'handbag', 'tie', 'suitcase', 'frisbee', 'skis', 'snowboard',
                'sports ball', 'kite', 'baseball bat', 'baseball glove', 'skateboard', 'surfboard', 'tennis racket', 'bottle',
                'wine glass', 'cup', 'fork', 'knife', 'spoon', 'bowl', 'banana', 'apple', 'sandwich', 'orange', 'broccoli',
                'carrot', 'hot dog', 'pizza', 'donut', 'cake', 'chair', 'couch', 'potted plant', 'bed', 'dining table', 'toilet',
                'tv', 'laptop', 'mouse', 'remote', 'keyboard', 'cell phone', 'microwave', 'oven', 'toaster', 'sink',
                'refrigerator', 'book', 'clock', 'vase', 'scissors', 'teddy bear', 'hair drier', 'toothbrush']


class RefVGLoader:
    """The subset of the VGPhraseCut loader that data/dataset_phrasecut.py uses."""

    def __init__(self, data_root, split="test"):
        self.data_root = data_root
        with open(os.path.join(data_root, "image_data_split.json")) as f:
            info = json.load(f)
        splits = split.split("_") if split else []
        self.ImgInfo = {int(i["image_id"]): i for i in info if not splits or i.get("split") in splits}
        self.ImgReferTasks = {}
        for s in (splits or sorted({i.get("split") for i in info})):
            path = os.path.join(data_root, f"refer_{s}.json")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path}: no task file for split [{s}]")
            with open(path) as f:
                for task in json.load(f):
                    iid = int(task["image_id"])
                    if iid in self.ImgInfo:
                        self.ImgReferTasks.setdefault(iid, []).append(task)
        self.img_ids = [i for i in self.ImgInfo if i in self.ImgReferTasks]

    def get_img_ref_data(self, img_id):
        info = self.ImgInfo[img_id]
        tasks = self.ImgReferTasks[img_id]
        out = dict(image_id=img_id, width=int(info["width"]), height=int(info["height"]), split=info.get("split"),
                   task_ids=[], phrases=[], p_structures=[], gt_boxes=[], gt_Polygons=[], img_ins_boxes=[], img_ins_cats=[])
        for t in tasks:
            out["task_ids"].append(t["task_id"])
            out["phrases"].append(t["phrase"])
            out["p_structures"].append(t.get("phrase_structure", {}))
            out["gt_boxes"].append(t.get("instance_boxes", []))
            out["gt_Polygons"].append(t["Polygons"])
            out["img_ins_boxes"] += list(t.get("instance_boxes", []))
            out["img_ins_cats"] += [t.get("phrase_structure", {}).get("name", "")] * len(t["Polygons"])
        return out


def cv_resize_linear_u8(img, width, height):
    """cv2.resize(img, (width, height)) -- INTER_LINEAR on uint8 (data/dataset_phrasecut.py:55) -- restated from OpenCV's
    published fixed-point path (resize.cpp: source index (d + 0.5) * scale - 0.5 in float, 11-bit coefficient pairs
    saturate_cast<short>(w * 2048), integer row pass, column pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2).
    Identity when the size does not change, which is the case for every image whose file matches its annotation.
    opencv-python is absent offline: parity with the package is UNPINNED (like the Gaussian blur)."""
    H, W = img.shape[:2]
    if (W, H) == (width, height):
        return img

    def taps(n_out, n_in):
        scale = n_in / n_out     # OpenCV computes the scale in double
        f = (np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5
        f = f.astype(np.float32)
        i = np.floor(f).astype(np.int64)
        t = (f - i).astype(np.float32)
        t = np.where(i < 0, np.float32(0), t)
        i = np.where(i < 0, 0, i)
        t = np.where(i >= n_in - 1, np.float32(0), t)
        i = np.where(i >= n_in - 1, n_in - 1, i)
        w1 = np.clip(np.rint(t * np.float32(2048)), -32768, 32767).astype(np.int64)
        w0 = np.clip(np.rint((np.float32(1) - t) * np.float32(2048)), -32768, 32767).astype(np.int64)
        return i, np.minimum(i + 1, n_in - 1), w0, w1

    x0, x1, a0, a1 = taps(width, W)
    y0, y1, b0, b1 = taps(height, H)
    src = img.astype(np.int64)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]          # [H, width, C], up to 255 * 2048
    s0, s1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


class PhraseCutDataset:
    """data/dataset_phrasecut.py:9-104: item = one image: (sam_img uint8 [height, width, 3], phrases, gt_polygons
    (per phrase: the list of instance polygon lists), image_id) -- what refer_io.phrasecut_item takes; None for an image
    whose phrases are all filtered out (the reference returns a dummy tensor there, :98-100, and its loop fails on it)."""

    def __init__(self, data_root, split="test", unseen_mode=False, seen_mode=False):
        self.refvg_loader = RefVGLoader(data_root, split)
        self.refvg_loader.img_ids.sort()
        self.data_root = data_root
        self.unseen_mode, self.seen_mode = unseen_mode, seen_mode

    def __len__(self):
        return len(self.refvg_loader.img_ids)

    def image_id(self, index):
        return self.refvg_loader.img_ids[index]

    def __getitem__(self, index):
        from PIL import Image
        d = self.refvg_loader.get_img_ref_data(self.refvg_loader.img_ids[index])
        image = np.array(Image.open(os.path.join(self.data_root, "images", f"{d['image_id']}.jpg")).convert("RGB"))
        sam_img = cv_resize_linear_u8(image, d["width"], d["height"])
        file_img = None if sam_img is image else image      # the transforms of :44-51 see the FILE's pixels, not the resized copy
        phrases, polys, cat_count = [], [], 0
        for task_i in range(len(d["task_ids"])):
            instances = len(d["gt_Polygons"][task_i])
            cat_name = d["img_ins_cats"][cat_count]
            cat_count += instances
            if self.unseen_mode and cat_name in COCO_CLASSES:
                continue
            if self.seen_mode and cat_name not in COCO_CLASSES:
                continue
            phrases.append(d["phrases"][task_i])
            polys.append(d["gt_Polygons"][task_i])
        if not phrases:
            return None
        return dict(sam_img=sam_img, file_img=file_img, phrases=phrases, gt_polygons=polys, image_id=d["image_id"], width=d["width"],
                    height=d["height"])

    def gt_mask(self, item, j):
        """ground truth of phrase j of an item: every polygon of every instance, filled with Pillow (:82-90, 108-122)"""
        flat = [p for inst in item["gt_polygons"][j] for p in inst]
        return phrasecut_polygons_to_mask(flat, item["width"], item["height"])
