"""REFER annotations -> the inputs of the hot path (host side).

Mirrors the reference's loader: `REFER` (refer/refer.py:40-291: refs(<splitBy>).p + instances.json, the index
tables, getRefIds / getImgIds / loadRefs / getMask) and the item layout of `ReferDataset`
(data/dataset_refer_bert.py:18-163).  Ground-truth masks come from the native codec in libhybridgl.so
(hgl_gt_mask_*), not from pycocotools.
"""
import ctypes as C
import json
import os
import pickle

import numpy as np

from . import _lib
from ._lib import check


def gt_mask_from_polygons(polygons, height, width):
    """list of flat [x0,y0,x1,y1,...] polygons -> (count image [H,W] uint8, summed area); mask.frPyObjects +
    mask.decode + np.sum(axis=2) of refer/refer.py:283-291."""
    lib = _lib.load()
    flat = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float64).ravel() for p in polygons])
                                if len(polygons) else np.zeros(0, np.float64))
    npts = np.ascontiguousarray([len(p) // 2 for p in polygons], dtype=np.int32)
    out = np.empty((height, width), np.uint8)
    area = C.c_int64(0)
    check(lib.hgl_gt_mask_from_polygons(flat.ctypes.data, npts.ctypes.data, len(polygons), int(height), int(width),
                                        out.ctypes.data, C.addressof(area)), "hgl_gt_mask_from_polygons")
    return out, int(area.value)


def gt_mask_from_rle(rle):
    """COCO RLE dict {'size': [h, w], 'counts': list | str | bytes} -> (mask [H,W] uint8, area)."""
    lib = _lib.load()
    h, w = int(rle["size"][0]), int(rle["size"][1])
    out = np.empty((h, w), np.uint8)
    area = C.c_int64(0)
    counts = rle["counts"]
    if isinstance(counts, (bytes, str)):
        s = counts if isinstance(counts, bytes) else counts.encode("ascii")
        check(lib.hgl_gt_mask_from_rle_string(C.c_char_p(s), h, w, out.ctypes.data, C.addressof(area)),
              "hgl_gt_mask_from_rle_string")
    else:
        c = np.ascontiguousarray(counts, dtype=np.uint32)
        check(lib.hgl_gt_mask_from_rle_counts(c.ctypes.data, len(c), h, w, out.ctypes.data, C.addressof(area)),
              "hgl_gt_mask_from_rle_counts")
    return out, int(area.value)


class REFER:
    """refer/refer.py:40-291 (the subset the evaluation loop uses)."""

    def __init__(self, data_root, dataset="refcoco", splitBy="unc"):
        self.DATA_DIR = os.path.join(data_root, dataset)
        if dataset in ("refcoco", "refcoco+", "refcocog"):
            self.IMAGE_DIR = os.path.join(data_root, "images/mscoco/images/train2014")
        elif dataset == "refclef":
            self.IMAGE_DIR = os.path.join(data_root, "images/saiapr_tc-12")
        else:
            raise ValueError(f"No refer dataset is called [{dataset}]")
        with open(os.path.join(self.DATA_DIR, "refs(" + splitBy + ").p"), "rb") as f:
            refs = pickle.load(f)
        with open(os.path.join(self.DATA_DIR, "instances.json"), "r") as f:
            instances = json.load(f)
        self.data = {"dataset": dataset, "refs": refs, "images": instances["images"],
                     "annotations": instances["annotations"], "categories": instances["categories"]}
        self.createIndex()

    def createIndex(self):
        self.Anns = {a["id"]: a for a in self.data["annotations"]}
        self.Imgs = {i["id"]: i for i in self.data["images"]}
        self.Cats = {c["id"]: c["name"] for c in self.data["categories"]}
        self.imgToAnns = {}
        for a in self.data["annotations"]:
            self.imgToAnns.setdefault(a["image_id"], []).append(a)
        self.Refs, self.imgToRefs, self.refToAnn, self.annToRef, self.catToRefs = {}, {}, {}, {}, {}
        self.Sents, self.sentToRef, self.sentToTokens = {}, {}, {}
        for ref in self.data["refs"]:
            self.Refs[ref["ref_id"]] = ref
            self.imgToRefs.setdefault(ref["image_id"], []).append(ref)
            self.catToRefs.setdefault(ref["category_id"], []).append(ref)
            self.refToAnn[ref["ref_id"]] = self.Anns[ref["ann_id"]]
            self.annToRef[ref["ann_id"]] = ref
            for sent in ref["sentences"]:
                self.Sents[sent["sent_id"]] = sent
                self.sentToRef[sent["sent_id"]] = ref
                self.sentToTokens[sent["sent_id"]] = sent["tokens"]

    def getRefIds(self, image_ids=(), cat_ids=(), ref_ids=(), split=""):
        """refer/refer.py:140-167 (testA/testB/testC match by the split's last letter, 'test' by substring)."""
        as_list = lambda v: list(v) if isinstance(v, (list, tuple)) else [v]
        image_ids, cat_ids, ref_ids = as_list(image_ids), as_list(cat_ids), as_list(ref_ids)
        if image_ids:
            refs = [r for i in image_ids for r in self.imgToRefs[i]]
        else:
            refs = self.data["refs"]
        if cat_ids:
            refs = [r for r in refs if r["category_id"] in cat_ids]
        if ref_ids:
            refs = [r for r in refs if r["ref_id"] in ref_ids]
        if split:
            if split in ("testA", "testB", "testC"):
                refs = [r for r in refs if split[-1] in r["split"]]
            elif split in ("testAB", "testBC", "testAC"):
                refs = [r for r in refs if r["split"] == split]
            elif split == "test":
                refs = [r for r in refs if "test" in r["split"]]
            elif split in ("train", "val"):
                refs = [r for r in refs if r["split"] == split]
            else:
                raise ValueError(f"No such split [{split}]")
        return [r["ref_id"] for r in refs]

    def getImgIds(self, ref_ids=()):
        ref_ids = list(ref_ids) if isinstance(ref_ids, (list, tuple)) else [ref_ids]
        if ref_ids:
            return list(set(self.Refs[r]["image_id"] for r in ref_ids))
        return list(self.Imgs.keys())

    def loadRefs(self, ref_ids=()):
        if isinstance(ref_ids, (list, tuple)):
            return [self.Refs[r] for r in ref_ids]
        return [self.Refs[ref_ids]]

    def getMask(self, ref):
        """refer/refer.py:277-291 -> {'mask': count image uint8 [H,W], 'area': int}."""
        ann = self.refToAnn[ref["ref_id"]]
        image = self.Imgs[ref["image_id"]]
        seg = ann["segmentation"]
        if isinstance(seg, list) and len(seg) > 0 and isinstance(seg[0], list):
            m, area = gt_mask_from_polygons(seg, image["height"], image["width"])
        else:
            m, area = gt_mask_from_rle(seg)
        return {"mask": m, "area": area}


class ReferDataset:
    """Item layout of data/dataset_refer_bert.py:103-163: (data dict, annot uint8 [H,W], sentence_raw list).
    `annot` keeps the pixels covered by exactly one polygon (:118-121)."""

    def __init__(self, refer_data_root, dataset="refcoco", splitBy="unc", split="val"):
        self.refer = REFER(refer_data_root, dataset, splitBy)
        self.ref_ids = self.refer.getRefIds(split=split)
        self.Cat_dict = self.refer.Cats
        self.sentence_raws = [[s["raw"] for s in self.refer.Refs[r]["sentences"]] for r in self.ref_ids]
        self.cat_names = [self.Cat_dict[self.refer.Refs[r]["category_id"]] for r in self.ref_ids]

    def __len__(self):
        return len(self.ref_ids)

    def image_id(self, index):
        return self.refer.Refs[self.ref_ids[index]]["image_id"]

    def image(self, index):
        """the decoded RGB image of item `index`, uint8 [H, W, 3] (data/dataset_refer_bert.py:107-110)"""
        from PIL import Image
        img_info = self.refer.Imgs[self.image_id(index)]
        return np.array(Image.open(os.path.join(self.refer.IMAGE_DIR, img_info["file_name"])).convert("RGB"))

    def target(self, index):
        """ground truth of item `index`: the pixels covered by exactly one polygon (:112-121), uint8 [H, W]"""
        return (self.refer.getMask(self.refer.Refs[self.ref_ids[index]])["mask"] == 1).astype(np.uint8)

    def __getitem__(self, index, sam_img=None):
        rid = self.ref_ids[index]
        ref = self.refer.Refs[rid]
        img_info = self.refer.Imgs[ref["image_id"]]
        if sam_img is None:
            sam_img = self.image(index)
        annot = self.target(index)
        data = dict(sam_img=sam_img, height=sam_img.shape[0], width=sam_img.shape[1], file_name=img_info["file_name"],
                    cat_name=self.cat_names[index], img_id=[ref["image_id"]], ref_id=rid,
                    sent_ids=list(ref["sent_ids"]))
        return data, annot, self.sentence_raws[index]


def phrasecut_polygons_to_mask(polygons, w, h):
    """data/dataset_phrasecut.py:108-122: instance polygons [[(x, y), ...], ...] -> bool [h, w]; vertices are
    truncated to int and filled with Pillow's polygon rasteriser (outline + fill), exactly as the reference does."""
    from PIL import Image, ImageDraw
    acc = np.zeros((h, w))
    for polygon in polygons:
        if len(polygon) < 2:
            continue
        pts = [(int(x), int(y)) for x, y in polygon]
        img = Image.new("L", (w, h), 0)
        ImageDraw.Draw(img).polygon(pts, outline=1, fill=1)
        acc += np.array(img)
    return acc > 0


def phrasecut_item(sam_img, phrases, gt_polygons, device, tokenizer, parse=None, heatmaps=None, image_id=None,
                   context_length=77):
    """One PhraseCut dataset item (data/dataset_phrasecut.py:36-104; Hybridgl_main_PhraseCut.py:67-119): one image,
    all its phrases, one ground-truth mask per phrase (gt_polygons[i] = the list of instance polygon lists of
    phrase i, flattened as the reference does) -> pipeline.RefBatch whose sentences carry their own targets."""
    import torch
    from . import ops, synth
    from .pipeline import RefBatch, Sentence
    from .tokenizer import tokenize
    H, W = sam_img.shape[:2]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    strings, sents = [], []
    for i, phrase in enumerate(phrases):
        rec = (parse or {}).get(phrase, {})
        row = len(strings)
        others = list(rec.get("other_nouns", []))   # bare phrases; Hybridgl_main_PhraseCut.py:147 encodes 'a photo of ' + other_noun
        strings += [phrase, rec.get("noun_phrase", phrase)] + ["a photo of " + o for o in others]
        flat = [p for inst in gt_polygons[i] for p in inst]
        gt = phrasecut_polygons_to_mask(flat, W, H)
        attn = t(np.asarray(heatmaps[i], np.float32)) if heatmaps is not None else torch.ones((H, W), dtype=torch.float32, device=device)
        sents.append(Sentence(row, row + 1, list(range(row + 2, row + 2 + len(others))), rec.get("dirflag", "none"),
                              rec.get("relaflag", "none"), len(others), attn, t(gt.astype(np.uint8))))
    tokens = tokenize(strings, context_length=context_length, tokenizer=tokenizer)
    return RefBatch(t(sam_img), ops.gaussian_blur_u8(t(sam_img)), t(synth.imagenet_normalize(sam_img)),
                    torch.zeros((1, H, W), dtype=torch.bool, device=device), torch.zeros((1, 4), dtype=torch.int64, device=device),
                    t(tokens), sents[0].target, sents, None, image_id)
