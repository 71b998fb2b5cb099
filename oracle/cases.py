"""ORACLE support (test infrastructure): seeded inputs of the golden cases, shared by
oracle/gen_golden.py (which produced tests/golden/*.npz from the reference) and tests/."""
import numpy as np

from hybridgl_amd.synth import synth_masks


def views_for_case(N, res, H, W):
    """Seeded (local, global, masks) of one golden case -- shared with tests/ via import."""
    rng = np.random.default_rng(100 + N)
    loc = rng.standard_normal((N, 3, res, res)).astype(np.float32)
    glo = rng.standard_normal((N, 3, res, res)).astype(np.float32)
    return loc, glo, edge_masks(N, H, W, 200 + N)


def edge_masks(N, H, W, seed):
    """ragged set: seeded blobs + empty-after-resize (1 px), full, thin line, empty."""
    m = synth_masks(N, H, W, seed)
    specials = []
    e = np.zeros((H, W), bool); e[H // 2 + 1, W // 2 + 1] = True; specials.append(e)   # single pixel
    specials.append(np.ones((H, W), bool))                                             # full
    t = np.zeros((H, W), bool); t[:, W // 3] = True; specials.append(t)                # thin column
    specials.append(np.zeros((H, W), bool))                                            # empty
    for i, s in enumerate(specials):
        if i < N:
            m[N - 1 - i] = s
    return m




RESIZE_CASES = [(640, 640, 14, 14), (427, 640, 14, 14), (97, 33, 4, 4),
                (64, 48, 56, 56), (480, 640, 224, 224), (5, 7, 14, 14)]


def resize_case(i):
    """input [C,H,W] fp32 of resize golden case i (first three are 0/1 masks)."""
    H, W, oh, ow = RESIZE_CASES[i]
    rng = np.random.default_rng(30 + i)
    x = rng.random((2 if i != 4 else 1, H, W)).astype(np.float32)
    if i < 3:
        x = (x > 0.7).astype(np.float32)
    return x, (H, W, oh, ow)


TAIL_CASES = [("none", "none", False), ("left", "left", True), ("big", "middle", False),
              ("within", "right", True), ("small", "none", True), ("up", "left", False),
              ("down", "none", True), ("right", "none", False)]


def tail_case(ci, N=12, E=32, H=96, W=128):
    """inputs of scoring-tail golden case ci: hybrid, t_pos, t_neg, masks, boxes, attn, gt."""
    from hybridgl_amd.synth import boxes_from_masks, synth_heatmap
    r = np.random.default_rng(500 + ci)
    hybrid = r.standard_normal((N, E)).astype(np.float32)
    t_pos = r.standard_normal((1, E)).astype(np.float32)
    t_neg = r.standard_normal((1, E)).astype(np.float32)
    masks = synth_masks(N, H, W, 600 + ci)
    boxes = boxes_from_masks(masks)
    attn = synth_heatmap(H, W, 700 + ci)
    gt = masks[(7 * ci) % N]
    return hybrid, t_pos, t_neg, masks, boxes, attn, gt


def sam_tiny_case():
    """inputs of the SAM tiny-geometry golden: a 160x200 RGB image, its PIL-bilinear resize to the
    256 long side (done here exactly as ResizeLongestSide.apply_image does), and 5 prompt points."""
    from PIL import Image
    from hybridgl_amd.synth import synth_image
    img = synth_image(160, 200, 42)
    oh, ow = 160, 200
    scale = 256.0 / max(oh, ow)
    nh, nw = int(oh * scale + 0.5), int(ow * scale + 0.5)
    resized = np.array(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    pts = np.array([[20.5, 30.25], [100.0, 80.0], [199.0, 159.0], [0.0, 0.0], [150.5, 10.0]], dtype=np.float64)
    pts_in = pts.copy()
    pts_in[:, 0] *= nw / ow
    pts_in[:, 1] *= nh / oh
    return dict(image=img, resized=resized, input_size=(nh, nw), orig_size=(oh, ow), points=pts, points_in=pts_in)


def sam_prompts_case():
    """prompts of tests/golden/sam_prompts.npz on the sam_tiny_case image (original frame, 160 x 200): single points with
    labels (0 = background), boxes XYXY, and the prompts of the two predict() calls"""
    pts = np.array([[20.5, 30.25], [100.0, 80.0], [199.0, 159.0], [0.0, 0.0]], dtype=np.float64)
    labels = np.array([0, 1, 0, 0], dtype=np.int32)
    boxes = np.array([[10.0, 20.0, 120.5, 90.0], [0.0, 0.0, 199.0, 159.0], [60.25, 70.0, 61.0, 150.75], [150.0, 5.0, 180.0, 40.0]],
                     dtype=np.float64)
    pairs = np.array([[[20.5, 30.25], [150.0, 100.0]], [[100.0, 80.0], [10.0, 12.5]], [[199.0, 159.0], [60.0, 60.0]],
                      [[0.0, 0.0], [190.5, 20.0]]], dtype=np.float64)
    pair_labels = np.array([[1, 0], [1, 1], [0, 1], [0, 0]], dtype=np.int32)
    rng = np.random.default_rng(11)
    many = np.stack([rng.random((2, 6)) * 199.0, rng.random((2, 6)) * 159.0], -1)          # [2 prompts, 6 points, (x, y)]
    many_labels = np.array([[1, 0, 1, 1, 0, 0], [0, 1, 1, 0, 1, 1]], dtype=np.int32)
    return dict(points=pts, labels=labels, boxes=boxes, one_point=np.array([[33.3, 77.7]]), one_label=np.array([0]),
                one_box=np.array([12.3, 45.6, 130.1, 140.9]), pairs=pairs, pair_labels=pair_labels, many=many,
                many_labels=many_labels)


def sam_crops_case():
    """inputs of the crop-layer generator golden (tests/golden/sam_crops.npz): a 240x320 RGB image run through
    SamAutomaticMaskGenerator(points_per_side=8, crop_n_layers=1, crop_n_points_downscale_factor=2) at the
    tiny geometry.  With seeded random weights the mask logits are pixel noise, so the model's mask_threshold is
    raised until a mask is a handful of pixels: the boxes then differ from mask to mask and the crop-edge filter,
    the per-crop NMS and the cross-crop NMS all have something to decide."""
    from hybridgl_amd.synth import synth_image
    return dict(image=synth_image(240, 320, 77), points_per_side=8, crop_n_layers=1, downscale=2,
                box_nms_thresh=0.7, crop_nms_thresh=0.7, logit_quantile=0.9995)


def tie_case(ci, dup):
    """tail_case(ci) with proposals `dup[1:]` made exact copies of proposal dup[0] (feature row, mask, box): their scores
    tie exactly in both soft-maxes, so the winners depend on how argmax / topk order equal values"""
    hybrid, t_pos, t_neg, masks, boxes, attn, gt = tail_case(ci, 12, 32, 96, 128)
    for j in dup[1:]:
        hybrid[j] = hybrid[dup[0]]
        masks[j] = masks[dup[0]]
        boxes[j] = boxes[dup[0]]
    return hybrid, t_pos, t_neg, masks, boxes, attn, gt


TIE_PLAN = [(0, (2, 5), "none", "none", False), (1, (0, 7, 9), "left", "left", True), (2, (4, 1), "big", "middle", True),
            (3, (11, 3, 6), "within", "right", False), (4, (8, 10), "small", "none", True), (5, (1, 2, 3, 4), "up", "left", False)]


def nan_case(kind):
    """tail_case(0) pushed into the divisions by zero of Hybridgl_main.py:203-223: a constant heat-map (min-max = 0/0: every
    coherence score NaN), a proposal with an EMPTY mask (sum = 0) or a FULL mask ((1 - m).sum() = 0) -- also as the
    proposal with the best CLIP score, so that the NaN reaches the blended top-k scores and torch.argmax's NaN rule decides"""
    import numpy as np
    hybrid, t_pos, t_neg, masks, boxes, attn, gt = tail_case(0, 12, 32, 96, 128)
    masks = masks.copy()
    hn = hybrid / np.linalg.norm(hybrid, axis=1, keepdims=True)
    top = int(np.argmax(hn @ (t_pos[0] / np.linalg.norm(t_pos[0]))))
    if kind == "const_attn":
        attn = np.full_like(attn, 0.5)
    elif kind == "empty_mask":
        masks[3] = 0
    elif kind == "full_mask":
        masks[4] = 1
    elif kind == "empty_top":
        masks[top] = 0
    elif kind == "full_top":
        masks[top] = 1
    elif kind == "zero_box_top":      # relation "within" divides by the area of box i (utils.py:262): 0/0 or x/0
        boxes = boxes.copy()
        boxes[top, 2:] = 0
    elif kind == "nan_row":           # one proposal's feature row is NaN: torch.argmax answers with that proposal
        hybrid = hybrid.copy()
        hybrid[5, 4] = np.nan
    elif kind == "nan_rows2":
        hybrid = hybrid.copy()
        hybrid[7, 2] = np.nan
        hybrid[2, 9] = np.nan
    elif kind == "zero_box_other":
        boxes = boxes.copy()
        boxes[(top + 1) % len(boxes), 2] = 0
        boxes[(top + 5) % len(boxes), 3] = 0
    else:
        raise ValueError(kind)
    return hybrid, t_pos, t_neg, masks, boxes, attn, gt


NAN_PLAN = [("const_attn", "none", "none", False), ("empty_mask", "none", "none", False), ("full_mask", "big", "middle", True),
            ("empty_top", "none", "none", False), ("empty_top", "left", "left", True), ("full_top", "small", "right", False),
            ("zero_box_top", "within", "none", False), ("zero_box_top", "within", "left", True), ("zero_box_other", "within", "none", True),
            ("zero_box_top", "big", "none", False),
            # NaN FEATURES: only the pure-CLIP index is defined (first NaN); every soft-maxed score is NaN after that and
            # which entries torch.topk returns for an all-NaN vector is an implementation detail -- tests compare idx[0] only
            ("nan_row", "none", "none", False), ("nan_rows2", "left", "left", True)]


# (case id, number of other nouns, relation word, direction flag): the text glue of Hybridgl_main.py:146-165 --
# r * sentence + (1 - r) * noun phrase, the MEAN of the other nouns' features (zeros when there are none)
GLUE_PLAN = [(20, 0, "none", "none"), (21, 1, "left", "left"), (22, 2, "big", "middle"), (23, 3, "within", "right"), (24, 1, "small", "none")]


def glue_tokens(ci, n_other):
    from hybridgl_amd import synth
    return synth.synth_tokens(2 + n_other, 16, 512, 7000 + ci)
