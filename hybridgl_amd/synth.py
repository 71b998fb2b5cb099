"""Synthetic, seeded inputs of the benchmark / parity workload.

Pure numpy, deterministic for a given seed on any machine: images (smooth noise), mask
proposals (ellipses / rectangles covering 1-40 % of the image), boxes (the reference's
inclusive XYWH rule, utils/amg.py:303-346), a smooth positive heat-map standing in for the
external GEM output, and token id rows.
"""
import numpy as np


def synth_image(H, W, seed):
    """uint8 [H,W,3]: sum of 8 low-frequency sinusoids per channel + U(0,16) jitter."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.zeros((H, W, 3), dtype=np.float32)
    for c in range(3):
        for _ in range(8):
            fy, fx = rng.uniform(0.5, 4.0, size=2)
            ph = rng.uniform(0, 2 * np.pi)
            amp = rng.uniform(8, 24)
            img[..., c] += amp * np.sin(2 * np.pi * (fy * yy / H + fx * xx / W) + ph)
    img += 128.0
    img += rng.uniform(0, 16, size=img.shape).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_masks(N, H, W, seed):
    """bool [N,H,W]: axis-aligned ellipses (even index) / rectangles (odd), area U(1%,40%)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    out = np.zeros((N, H, W), dtype=bool)
    for n in range(N):
        area = rng.uniform(0.01, 0.40) * H * W
        aspect = rng.uniform(0.5, 2.0)
        if n % 2 == 0:  # ellipse: pi*a*b = area
            a = np.sqrt(area * aspect / np.pi)
            b = area / (np.pi * a)
        else:           # rectangle: (2a)*(2b) = area
            a = np.sqrt(area * aspect) / 2
            b = area / (4 * a)
        a, b = min(a, W / 2 - 1), min(b, H / 2 - 1)
        cx = rng.uniform(a, W - a)
        cy = rng.uniform(b, H - b)
        if n % 2 == 0:
            out[n] = ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1.0
        else:
            out[n] = (np.abs(xx - cx) <= a) & (np.abs(yy - cy) <= b)
        if not out[n].any():
            out[n, int(cy), int(cx)] = True
    return out


def boxes_from_masks(masks):
    """int64 [N,4] XYWH with w = x1-x0, h = y1-y0 of inclusive pixel coordinates
    (batched_mask_to_box + box_xyxy_to_xywh, utils/amg.py:303-346,91-95)."""
    out = np.zeros((len(masks), 4), dtype=np.int64)
    for i, m in enumerate(masks):
        ys, xs = np.nonzero(m)
        if len(ys):
            out[i] = [xs.min(), ys.min(), xs.max() - xs.min(), ys.max() - ys.min()]
    return out


def synth_heatmap(H, W, seed):
    """fp32 [H,W] smooth positive field (stand-in for the GEM heat-map, Hybridgl_main.py:200)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    f = np.zeros((H, W), dtype=np.float32)
    for _ in range(6):
        cy, cx = rng.uniform(0, H), rng.uniform(0, W)
        s = rng.uniform(0.08, 0.3) * max(H, W)
        f += rng.uniform(0.3, 1.0) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
    f += rng.uniform(0, 0.05, size=f.shape).astype(np.float32)
    return f.astype(np.float32)


def synth_tokens(B, context, vocab, seed):
    """int32 [B,context]: SOT, 3..12 random ids, EOT (= vocab-1, the arg-max pooled token), zeros."""
    rng = np.random.default_rng(seed)
    tok = np.zeros((B, context), dtype=np.int32)
    for b in range(B):
        n = int(rng.integers(3, min(13, context - 2)))
        tok[b, 0] = vocab - 2
        tok[b, 1:1 + n] = rng.integers(1, vocab - 2, size=n)
        tok[b, 1 + n] = vocab - 1
    return tok


# parse records cycled by the synthetic workload: (dirflag, relaflag, n_other_nouns)
PARSE_RECORDS = [("left", "left", 1), ("none", "big", 0), ("middle", "none", 2)]


def imagenet_normalize(img_u8):
    """dataset transform (data/dataset_refer_bert.py:155): ToTensor + Normalize(ImageNet) -> [3,H,W]."""
    x = img_u8.astype(np.float32) / np.float32(255)
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)
    return ((x - mean) / std).transpose(2, 0, 1).astype(np.float32).copy()


def box_blur_u8(img, k=15):
    """Deterministic stand-in for cv2.GaussianBlur(img,(15,15),0) (OpenCV is absent offline and
    its fixed-point kernel is unpinned): separable Gaussian, sigma per OpenCV's
    rule 0.3*((k-1)*0.5-1)+0.8, reflect-101 borders, round-half-up to uint8."""
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    r = k // 2
    x = np.arange(-r, r + 1, dtype=np.float64)
    g = np.exp(-(x * x) / (2 * sigma * sigma))
    g /= g.sum()
    a = img.astype(np.float64)
    pad = np.pad(a, ((r, r), (0, 0), (0, 0)), mode="reflect")
    a = sum(g[i] * pad[i:i + img.shape[0]] for i in range(k))
    pad = np.pad(a, ((0, 0), (r, r), (0, 0)), mode="reflect")
    a = sum(g[i] * pad[:, i:i + img.shape[1]] for i in range(k))
    return np.clip(np.floor(a + 0.5), 0, 255).astype(np.uint8)


# ---- a REFER tree on disk (the evaluator's real feed: JPEG decode, GEM transform, BPE, polygon rasterisation, uploads) -------
_NOUNS = ("man woman girl boy person child player dog cat horse zebra giraffe elephant bear sheep cow bird bus car truck train "
          "bike motorcycle boat plane bench chair couch table bed laptop phone book clock vase cup bowl bottle pizza sandwich "
          "cake banana apple orange donut umbrella kite surfboard skateboard racket bat glove hat shirt jacket").split()
_ADJS = "red blue green white black yellow brown striped tall short young old big small dark bright wooden empty".split()
_SPATIAL = (("on the left", "left", "none"), ("on the right", "right", "none"), ("in the middle", "middle", "none"),
            ("at the top", "up", "none"), ("at the bottom", "down", "none"), ("to the left of the {o}", "none", "left"),
            ("behind the {o}", "none", "behind"), ("next to the {o}", "none", "none"), ("bigger than the {o}", "none", "big"),
            ("smaller than the {o}", "none", "small"), ("inside the {o}", "none", "within"), ("", "none", "none"))
REFER_IMAGE_SIZES = ((480, 640), (640, 480), (427, 640), (640, 427), (375, 500), (500, 375), (480, 640), (640, 640))


def synth_sentence(rng):
    """one referring expression + the parse record the (external) spaCy step would produce for it:
    (raw, {"noun_phrase", "other_nouns", "dirflag", "relaflag"})"""
    noun, adj = _NOUNS[int(rng.integers(len(_NOUNS)))], _ADJS[int(rng.integers(len(_ADJS)))]
    other = _NOUNS[int(rng.integers(len(_NOUNS)))]
    tail, dirflag, relaflag = _SPATIAL[int(rng.integers(len(_SPATIAL)))]
    np_ = f"the {adj} {noun}"
    uses_other = "{o}" in tail
    raw = (np_ + " " + tail.format(o=other)).strip()
    return raw, {"noun_phrase": np_, "other_nouns": [other] if uses_other else [], "dirflag": dirflag, "relaflag": relaflag}


def write_bpe_merges(path, n_merges=600):
    """A byte-level BPE merges file in the format of OpenAI's bpe_simple_vocab_16e6.txt.gz (header line + one 'left right'
    merge per line) learnt from the synthetic sentences' word list with a plain most-frequent-pair trainer, so that -- as
    with the real vocabulary on real captions -- nearly every word is one token.  Our own data."""
    import collections
    import gzip
    from .tokenizer import byte_table
    enc, _ = byte_table()
    words = collections.Counter()
    corpus = list(_NOUNS) + list(_ADJS) + "the a photo of on left right in middle at top bottom to behind next bigger smaller than inside".split()
    for w in corpus:
        sym = [enc[b] for b in w.encode("utf-8")]
        sym[-1] += "</w>"
        words[tuple(sym)] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        (a, b), _c = max(sorted(pairs.items()), key=lambda kv: kv[1])
        merges.append((a, b))
        new = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and w[i] == a and w[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(w[i])
                    i += 1
            new[tuple(out)] += c
        words = new
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(("#version: synthetic\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n").encode("utf-8"))
    return len(merges)


def write_refer_tree(root, n_images=200, dataset="refcoco", splitBy="unc", split="val", sentences_per_ref=3, seed=0,
                     sizes=REFER_IMAGE_SIZES, fmt="jpg", far_refs=0.1):
    """A dataset in the directory layout refer/refer.py:40-76 reads -- <root>/<dataset>/refs(<splitBy>).p, instances.json,
    <root>/images/mscoco/images/train2014/*.jpg -- plus <root>/parse.json (the parse records, keyed by sent_id) and
    <root>/bpe.txt.gz: COCO-sized JPEGs of mixed sizes, 2-3 refs per image (RefCOCO: 2.6), `sentences_per_ref` sentences
    per ref, polygon ground truth.  The refs of an image are neighbours in the refs file except a fraction `far_refs`
    of them, which come back ~20 images later (the per-image cache of the loop has to hold them).  Returns
    {"refs", "images", "sentences"} counts."""
    import json
    import os
    import pickle
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, dataset), exist_ok=True)
    img_dir = os.path.join(root, "images/mscoco/images/train2014")
    os.makedirs(img_dir, exist_ok=True)
    images, anns, refs, parse, late = [], [], [], {}, []
    sent_id = 0
    for i in range(n_images):
        h, w = sizes[i % len(sizes)]
        name = f"COCO_train2014_{i:012d}.{fmt}"
        # a cheap image with JPEG-realistic entropy: low-frequency field + per-pixel jitter (synth_image costs 0.3 s at 640 x 480)
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        img = np.empty((h, w, 3), np.float32)
        for c in range(3):
            fy, fx, ph = rng.uniform(0.5, 4.0), rng.uniform(0.5, 4.0), rng.uniform(0, 6.28)
            img[..., c] = 128 + 60 * np.sin(6.28 * (fy * yy / h + fx * xx / w) + ph)
        img += rng.uniform(0, 24, size=img.shape).astype(np.float32)
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(img_dir, name), quality=90)
        images.append({"id": 1000 + i, "file_name": name, "height": h, "width": w})
        for j in range(2 + int(rng.integers(2))):
            aid, rid = 10 * (1000 + i) + j, 100000 + 10 * i + j
            cx, cy = rng.uniform(0.25, 0.75) * w, rng.uniform(0.25, 0.75) * h
            ax, ay = rng.uniform(0.08, 0.24) * w, rng.uniform(0.08, 0.24) * h
            th = np.linspace(0, 2 * np.pi, 24, endpoint=False)
            rad = 1 + 0.15 * np.sin(3 * th + rng.uniform(0, 6.28))
            poly = np.stack([cx + ax * rad * np.cos(th), cy + ay * rad * np.sin(th)], axis=1).round(2).ravel().tolist()
            anns.append({"id": aid, "image_id": 1000 + i, "category_id": 1, "segmentation": [poly], "bbox": [0, 0, 1, 1]})
            sents = []
            for _ in range(sentences_per_ref):
                raw, rec = synth_sentence(rng)
                sents.append({"sent_id": sent_id, "raw": raw, "sent": raw, "tokens": raw.split()})
                parse[str(sent_id)] = rec
                sent_id += 1
            ref = {"ref_id": rid, "ann_id": aid, "image_id": 1000 + i, "category_id": 1, "split": split,
                   "sent_ids": [s["sent_id"] for s in sents], "sentences": sents}
            if j > 0 and rng.uniform() < far_refs:
                late.append((len(refs) + 50, ref))
            else:
                refs.append(ref)
        while late and late[0][0] <= len(refs):
            refs.append(late.pop(0)[1])
    refs += [r for _, r in late]
    json.dump({"images": images, "annotations": anns, "categories": [{"id": 1, "name": "thing"}]},
              open(os.path.join(root, dataset, "instances.json"), "w"))
    pickle.dump(refs, open(os.path.join(root, dataset, f"refs({splitBy}).p"), "wb"))
    json.dump(parse, open(os.path.join(root, "parse.json"), "w"))
    write_bpe_merges(os.path.join(root, "bpe.txt.gz"))
    return {"refs": len(refs), "images": len(images), "sentences": sent_id}


def write_phrasecut_tree(root, n_images=8, split="test", phrases_per_image=8, seed=0, sizes=((480, 640), (640, 480), (375, 500)),
                         resized_files=0.25):
    """A dataset in the published VGPhraseCut layout (hybridgl_amd/phrasecut_io.py): <root>/image_data_split.json,
    <root>/refer_<split>.json, <root>/images/<image_id>.jpg, plus <root>/parse.json (parse records keyed by phrase) and
    <root>/bpe.txt.gz.  Every task has 1-3 instances of 1-2 polygons; a fraction `resized_files` of the image FILES is
    stored at 3/4 of the annotated size, so that the cv2.resize step of data/dataset_phrasecut.py:55 has work to do.
    Returns {"images", "phrases"}."""
    import json
    import os
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    infos, tasks, parse = [], [], {}
    n_phr = 0
    for i in range(n_images):
        h, w = sizes[i % len(sizes)]
        iid = 2300000 + 7 * i
        fh, fw = (h * 3 // 4, w * 3 // 4) if rng.uniform() < resized_files else (h, w)
        yy, xx = np.mgrid[0:fh, 0:fw].astype(np.float32)
        img = np.empty((fh, fw, 3), np.float32)
        for c in range(3):
            fy, fx, ph = rng.uniform(0.5, 4.0), rng.uniform(0.5, 4.0), rng.uniform(0, 6.28)
            img[..., c] = 128 + 60 * np.sin(6.28 * (fy * yy / fh + fx * xx / fw) + ph)
        img += rng.uniform(0, 24, size=img.shape).astype(np.float32)
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(root, "images", f"{iid}.jpg"), quality=90)
        infos.append({"image_id": iid, "width": w, "height": h, "split": split, "coco_id": None})
        for j in range(phrases_per_image):
            raw, rec = synth_sentence(rng)
            phrase = raw.replace("the ", "", 1).capitalize() if j % 3 == 0 else raw      # the evaluator lower-cases phrases
            polys, boxes = [], []
            for _ in range(1 + int(rng.integers(3))):
                inst = []
                for _ in range(1 + int(rng.integers(2))):
                    cx, cy = rng.uniform(0.2, 0.8) * w, rng.uniform(0.2, 0.8) * h
                    ax, ay = rng.uniform(0.05, 0.18) * w, rng.uniform(0.05, 0.18) * h
                    th = np.linspace(0, 2 * np.pi, 16, endpoint=False)
                    inst.append(np.stack([cx + ax * np.cos(th), cy + ay * np.sin(th)], axis=1).round(2).tolist())
                pts = np.concatenate([np.asarray(p) for p in inst])
                x0, y0 = pts.min(axis=0)
                x1, y1 = pts.max(axis=0)
                boxes.append([float(x0), float(y0), float(x1 - x0), float(y1 - y0)])
                polys.append(inst)
            noun = rec["noun_phrase"].split()[-1]
            tasks.append({"task_id": f"{iid}__{j}", "image_id": iid, "phrase": phrase,
                          "phrase_structure": {"name": noun, "attributes": [], "relation_descriptions": [], "type": "name"},
                          "instance_boxes": boxes, "Polygons": polys})
            parse[phrase] = rec
            n_phr += 1
    json.dump(infos, open(os.path.join(root, "image_data_split.json"), "w"))
    json.dump(tasks, open(os.path.join(root, f"refer_{split}.json"), "w"))
    json.dump(parse, open(os.path.join(root, "parse.json"), "w"))
    write_bpe_merges(os.path.join(root, "bpe.txt.gz"))
    return {"images": n_images, "phrases": n_phr}
