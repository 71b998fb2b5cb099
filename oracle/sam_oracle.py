"""ORACLE (test infrastructure, NOT product code): numpy fp32 restatement of the reference's
segment-anything path used by Hybridgl_main.py:66-74,85 (SamAutomaticMaskGenerator over SAM ViT-H).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
Pinned against fixtures produced by importing the reference package itself
(oracle/gen_golden.py -> tests/golden/sam_*.npz).  Paths cited are relative to
third_party/segment-anything/segment_anything/ in the reference tree.
"""
import math

import numpy as np
from scipy.special import erf

from .clip_oracle import F32, bilinear_resize, layer_norm, linear, softmax


def gelu(x):
    """nn.GELU (erf form), modeling/common.py:21-27 MLPBlock act."""
    return (F32(0.5) * x * (F32(1) + erf(x / F32(math.sqrt(2.0))))).astype(F32)


# ----------------------------------------------------------------------------- image encoder
def window_partition(x, ws):
    """modeling/image_encoder.py:243-266.  x: [B,H,W,C] -> [B*nW, ws, ws, C], (Hp, Wp)."""
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = np.pad(x, ((0, 0), (0, ph), (0, pw), (0, 0)))
    Hp, Wp = H + ph, W + pw
    x = x.reshape(B, Hp // ws, ws, Wp // ws, ws, C).transpose(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(w, ws, pad_hw, hw):
    """modeling/image_encoder.py:269-290."""
    Hp, Wp = pad_hw
    H, W = hw
    B = w.shape[0] // (Hp * Wp // ws // ws)
    x = w.reshape(B, Hp // ws, Wp // ws, ws, ws, -1).transpose(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, -1)
    return x[:, :H, :W, :]


def get_rel_pos(q_size, k_size, rel_pos):
    """modeling/image_encoder.py:292-322 (no interpolation needed: table length == 2*size-1)."""
    max_rel = int(2 * max(q_size, k_size) - 1)
    assert rel_pos.shape[0] == max_rel, "rel-pos interpolation is not on the reference's path"
    q = np.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k = np.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (q - k) + (k_size - 1) * max(q_size / k_size, 1.0)
    return rel_pos[rel.astype(np.int64)]


def encoder_attention(x, sd, p, heads):
    """Attention.forward + add_decomposed_rel_pos (modeling/image_encoder.py:224-240,325-361)."""
    B, H, W, D = x.shape
    hd = D // heads
    qkv = linear(x.reshape(B, H * W, D), sd[f"{p}.qkv.weight"], sd[f"{p}.qkv.bias"])
    qkv = qkv.reshape(B, H * W, 3, heads, hd).transpose(2, 0, 3, 1, 4).reshape(3, B * heads, H * W, hd)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q * F32(hd ** -0.5)) @ k.transpose(0, 2, 1)
    Rh = get_rel_pos(H, H, sd[f"{p}.rel_pos_h"])
    Rw = get_rel_pos(W, W, sd[f"{p}.rel_pos_w"])
    rq = q.reshape(B * heads, H, W, hd)
    rel_h = np.einsum("bhwc,hkc->bhwk", rq, Rh).astype(F32)
    rel_w = np.einsum("bhwc,wkc->bhwk", rq, Rw).astype(F32)
    attn = (attn.reshape(B * heads, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :])
    attn = softmax(attn.reshape(B * heads, H * W, H * W).astype(F32), axis=-1)
    o = (attn @ v).reshape(B, heads, H, W, hd).transpose(0, 2, 3, 1, 4).reshape(B, H, W, D)
    return linear(o, sd[f"{p}.proj.weight"], sd[f"{p}.proj.bias"])


def encoder_block(x, sd, p, heads, window):
    """Block.forward (modeling/image_encoder.py:166-182), eps=1e-6 (build_sam.py:70)."""
    short = x
    x = layer_norm(x, sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], 1e-6)
    if window > 0:
        H, W = x.shape[1:3]
        x, pad_hw = window_partition(x, window)
    x = encoder_attention(x, sd, f"{p}.attn", heads)
    if window > 0:
        x = window_unpartition(x, window, pad_hw, (H, W))
    x = short + x
    h = layer_norm(x, sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], 1e-6)
    h = gelu(linear(h, sd[f"{p}.mlp.lin1.weight"], sd[f"{p}.mlp.lin1.bias"]))
    return x + linear(h, sd[f"{p}.mlp.lin2.weight"], sd[f"{p}.mlp.lin2.bias"])


def layer_norm_2d(x, w, b, eps=1e-6):
    """LayerNorm2d over the channel axis of NHWC data (modeling/common.py:30-43)."""
    return layer_norm(x, w, b, eps)


def conv3x3_nhwc(x, w):
    """nn.Conv2d(k=3, padding=1, bias=False) on [H,W,C] with weight [O,C,3,3]."""
    H, W, C = x.shape
    xp = np.pad(x, ((1, 1), (1, 1), (0, 0)))
    cols = np.stack([xp[ky:ky + H, kx:kx + W, :] for ky in range(3) for kx in range(3)], axis=-1)  # [H,W,C,9]
    return (cols.reshape(H * W, C * 9) @ w.reshape(w.shape[0], -1).T).reshape(H, W, -1).astype(F32)


def image_encoder(sd, img, cfg):
    """ImageEncoderViT.forward (modeling/image_encoder.py:106-116). img: [3,S,S] fp32 (pre-processed).
    Returns the embedding in NHWC order [S/16, S/16, out_chans] (the reference is NCHW)."""
    p = "image_encoder"
    w = sd[f"{p}.patch_embed.proj.weight"]
    D, _, ps, _ = w.shape
    g = img.shape[-1] // ps
    cols = img.reshape(3, g, ps, g, ps).transpose(1, 3, 0, 2, 4).reshape(g * g, 3 * ps * ps)
    x = (cols @ w.reshape(D, -1).T + sd[f"{p}.patch_embed.proj.bias"]).reshape(1, g, g, D)
    x = (x + sd[f"{p}.pos_embed"]).astype(F32)
    for i in range(cfg["depth"]):
        window = 0 if i in cfg["global_attn_indexes"] else cfg["window_size"]
        x = encoder_block(x, sd, f"{p}.blocks.{i}", cfg["num_heads"], window)
    x = x[0]
    x = (x.reshape(g * g, D) @ sd[f"{p}.neck.0.weight"].reshape(-1, D).T).reshape(g, g, -1).astype(F32)
    x = layer_norm_2d(x, sd[f"{p}.neck.1.weight"], sd[f"{p}.neck.1.bias"])
    x = conv3x3_nhwc(x, sd[f"{p}.neck.2.weight"])
    return layer_norm_2d(x, sd[f"{p}.neck.3.weight"], sd[f"{p}.neck.3.bias"])


def preprocess(img_u8_hwc, img_size=1024):
    """Sam.preprocess (modeling/sam.py:164-174) on the already-resized image: normalise, zero-pad."""
    mean = np.array([123.675, 116.28, 103.53], dtype=F32)
    std = np.array([58.395, 57.12, 57.375], dtype=F32)
    x = ((img_u8_hwc.astype(F32) - mean) / std).transpose(2, 0, 1)
    out = np.zeros((3, img_size, img_size), dtype=F32)
    out[:, :x.shape[1], :x.shape[2]] = x
    return out


# ----------------------------------------------------------------------------- prompt encoder
def pe_encoding(sd, coords01):
    """PositionEmbeddingRandom._pe_encoding (modeling/prompt_encoder.py:185-192). coords in [0,1]."""
    g = sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    c = (F32(2) * coords01.astype(F32) - F32(1)) @ g
    c = F32(2 * np.pi) * c
    return np.concatenate([np.sin(c), np.cos(c)], axis=-1).astype(F32)


def dense_pe(sd, h, w):
    """PromptEncoder.get_dense_pe -> [h*w, C] (NHWC rows) (modeling/prompt_encoder.py:194-205)."""
    y = ((np.arange(h, dtype=F32) + F32(1)) - F32(0.5)) / F32(h)
    x = ((np.arange(w, dtype=F32) + F32(1)) - F32(0.5)) / F32(w)
    grid = np.stack(np.broadcast_arrays(x[None, :], y[:, None]), axis=-1)  # [h,w,(x,y)]
    return pe_encoding(sd, grid).reshape(h * w, -1)


def embed_points(sd, points_xy, input_size=1024):
    """PromptEncoder._embed_points for one positive point per prompt + the padding point
    (modeling/prompt_encoder.py:73-91). points_xy: float64 [P,2] in input-image pixels -> [P,2,C]."""
    P = points_xy.shape[0]
    pts = np.concatenate([points_xy[:, None, :].astype(np.float64) + 0.5, np.zeros((P, 1, 2))], axis=1)
    pts = (pts / float(input_size)).astype(F32)          # coords normalised in float64, then cast
    emb = pe_encoding(sd, pts)                           # [P,2,C]
    emb[:, 1, :] = sd["prompt_encoder.not_a_point_embed.weight"][0]
    emb[:, 0, :] += sd["prompt_encoder.point_embeddings.1.weight"][0]
    return emb.astype(F32)


def embed_prompts(sd, coords_xy, labels, input_size=1024):
    """PromptEncoder._embed_points / _embed_boxes for prompts of n tokens (modeling/prompt_encoder.py:73-101).
    coords_xy: [P,n,2] input-image pixels (float64 or float32: the arithmetic dtype of `+ 0.5` and the division, as the
    reference takes whatever the caller hands over), labels [P,n]: -1 padding, 0 / 1 points, 2 / 3 box corners -> [P,n,C]."""
    labels = np.asarray(labels)
    pts = ((coords_xy + coords_xy.dtype.type(0.5)) / coords_xy.dtype.type(input_size)).astype(F32)
    emb = pe_encoding(sd, pts)
    emb[labels == -1] = 0.0
    emb[labels == -1] += sd["prompt_encoder.not_a_point_embed.weight"][0]
    for lab in range(4):
        emb[labels == lab] += sd[f"prompt_encoder.point_embeddings.{lab}.weight"][0]
    return emb.astype(F32)


# ----------------------------------------------------------------------------- mask decoder
def dec_attention(sd, p, q, k, v, heads):
    """modeling/transformer.py:185-240 Attention."""
    q = linear(q, sd[f"{p}.q_proj.weight"], sd[f"{p}.q_proj.bias"])
    k = linear(k, sd[f"{p}.k_proj.weight"], sd[f"{p}.k_proj.bias"])
    v = linear(v, sd[f"{p}.v_proj.weight"], sd[f"{p}.v_proj.bias"])
    B, Nq, C = q.shape
    hd = C // heads
    sep = lambda t: t.reshape(t.shape[0], t.shape[1], heads, hd).transpose(0, 2, 1, 3)
    qh, kh, vh = sep(q), sep(k), sep(v)
    a = softmax(((qh @ kh.transpose(0, 1, 3, 2)) / F32(math.sqrt(hd))).astype(F32), axis=-1)
    o = (a @ vh).transpose(0, 2, 1, 3).reshape(B, Nq, C)
    return linear(o, sd[f"{p}.out_proj.weight"], sd[f"{p}.out_proj.bias"])


def two_way_transformer(sd, keys, key_pe, tokens, heads=8, depth=2):
    """TwoWayTransformer.forward (modeling/transformer.py:62-106,151-182).
    keys: [B, HW, C]; key_pe: [HW, C]; tokens: [B, T, C]."""
    p = "mask_decoder.transformer"
    ln = lambda x, n: layer_norm(x, sd[f"{n}.weight"], sd[f"{n}.bias"], 1e-5)
    queries, point_pe = tokens, tokens
    for i in range(depth):
        l = f"{p}.layers.{i}"
        if i == 0:
            queries = dec_attention(sd, f"{l}.self_attn", queries, queries, queries, heads)
        else:
            q = queries + point_pe
            queries = queries + dec_attention(sd, f"{l}.self_attn", q, q, queries, heads)
        queries = ln(queries, f"{l}.norm1")
        q, k = queries + point_pe, keys + key_pe
        queries = ln(queries + dec_attention(sd, f"{l}.cross_attn_token_to_image", q, k, keys, heads), f"{l}.norm2")
        h = np.maximum(linear(queries, sd[f"{l}.mlp.lin1.weight"], sd[f"{l}.mlp.lin1.bias"]), 0)
        queries = ln(queries + linear(h, sd[f"{l}.mlp.lin2.weight"], sd[f"{l}.mlp.lin2.bias"]), f"{l}.norm3")
        q, k = queries + point_pe, keys + key_pe
        keys = ln(keys + dec_attention(sd, f"{l}.cross_attn_image_to_token", k, q, queries, heads), f"{l}.norm4")
    q, k = queries + point_pe, keys + key_pe
    queries = queries + dec_attention(sd, f"{p}.final_attn_token_to_image", q, k, keys, heads)
    return ln(queries, f"{p}.norm_final_attn"), keys


def conv_transpose_2x2(x, w, b):
    """nn.ConvTranspose2d(k=2, s=2) on NHWC [B,H,W,Cin] with weight [Cin,Cout,2,2] -> [B,2H,2W,Cout]."""
    B, H, W, Ci = x.shape
    Co = w.shape[1]
    y = (x.reshape(-1, Ci) @ w.reshape(Ci, Co * 4)).reshape(B, H, W, Co, 2, 2) + b[None, None, None, :, None, None]
    return y.transpose(0, 1, 4, 2, 5, 3).reshape(B, 2 * H, 2 * W, Co).astype(F32)


def mlp3(sd, p, x, n=3):
    """modeling/mask_decoder.py:155-176 MLP (ReLU between layers)."""
    for i in range(n):
        x = linear(x, sd[f"{p}.layers.{i}.weight"], sd[f"{p}.layers.{i}.bias"])
        if i < n - 1:
            x = np.maximum(x, 0)
    return x.astype(F32)


def embed_masks(sd, mask_input):
    """PromptEncoder._embed_masks = mask_downscaling (modeling/prompt_encoder.py:57-66, :103-106).
    mask_input [B,1,4h,4w] -> dense rows [B, h*w, C] (NHWC)."""
    p = "prompt_encoder.mask_downscaling"
    x = mask_input[:, 0].astype(F32)                                     # [B, 4h, 4w]

    def conv2(x_nhwc, w, b):                                              # Conv2d(k=2, s=2): w [co, ci, 2, 2]
        B, H, W, ci = x_nhwc.shape
        blk = x_nhwc.reshape(B, H // 2, 2, W // 2, 2, ci).transpose(0, 1, 3, 5, 2, 4).reshape(B, H // 2, W // 2, ci * 4)
        return (blk @ w.reshape(w.shape[0], -1).T + b).astype(F32)

    x = conv2(x[..., None], sd[f"{p}.0.weight"], sd[f"{p}.0.bias"])
    x = gelu(layer_norm_2d(x, sd[f"{p}.1.weight"], sd[f"{p}.1.bias"]))
    x = conv2(x, sd[f"{p}.3.weight"], sd[f"{p}.3.bias"])
    x = gelu(layer_norm_2d(x, sd[f"{p}.4.weight"], sd[f"{p}.4.bias"]))
    w3 = sd[f"{p}.6.weight"]
    x = (x @ w3.reshape(w3.shape[0], -1).T + sd[f"{p}.6.bias"]).astype(F32)
    return x.reshape(x.shape[0], -1, x.shape[-1])


def mask_decoder(sd, image_emb_nhwc, sparse, multimask=True, dense=None):
    """MaskDecoder.forward/predict_masks (modeling/mask_decoder.py:71-149).
    image_emb_nhwc: [h,w,C]; sparse: [B,P,C]; dense: None (no_mask_embed) or [B,h*w,C] (embed_masks)
    -> (low-res logits [B,3,4h,4w], iou [B,3])."""
    h, w, C = image_emb_nhwc.shape
    B = sparse.shape[0]
    out_tok = np.concatenate([sd["mask_decoder.iou_token.weight"], sd["mask_decoder.mask_tokens.weight"]], 0)
    tokens = np.concatenate([np.broadcast_to(out_tok, (B,) + out_tok.shape), sparse], axis=1).astype(F32)
    if dense is None:
        src = image_emb_nhwc.reshape(1, h * w, C) + sd["prompt_encoder.no_mask_embed.weight"].reshape(1, 1, C)
        src = np.broadcast_to(src, (B, h * w, C)).astype(F32)
    else:
        src = (image_emb_nhwc.reshape(1, h * w, C) + dense).astype(F32)
    pos = dense_pe(sd, h, w)
    hs, src = two_way_transformer(sd, src, pos, tokens)
    iou_tok, mask_tok = hs[:, 0, :], hs[:, 1:5, :]
    p = "mask_decoder.output_upscaling"
    x = conv_transpose_2x2(src.reshape(B, h, w, C), sd[f"{p}.0.weight"], sd[f"{p}.0.bias"])
    x = gelu(layer_norm_2d(x, sd[f"{p}.1.weight"], sd[f"{p}.1.bias"]))
    x = gelu(conv_transpose_2x2(x, sd[f"{p}.3.weight"], sd[f"{p}.3.bias"]))        # [B,4h,4w,C/8]
    hyper = np.stack([mlp3(sd, f"mask_decoder.output_hypernetworks_mlps.{i}", mask_tok[:, i, :]) for i in range(4)], 1)
    masks = np.einsum("btc,bhwc->bthw", hyper, x).astype(F32)
    iou = mlp3(sd, "mask_decoder.iou_prediction_head", iou_tok)
    if multimask:
        return masks[:, 1:], iou[:, 1:]
    return masks[:, :1], iou[:, :1]


# ----------------------------------------------------------------------------- post-processing / AMG
def postprocess_masks(low_res, input_size, original_size, img_size=1024):
    """Sam.postprocess_masks (modeling/sam.py:133-162)."""
    m = bilinear_resize(low_res, img_size, img_size)
    m = m[..., :input_size[0], :input_size[1]]
    return bilinear_resize(m, original_size[0], original_size[1])


def stability_score(logits, thr=0.0, off=1.0):
    """utils/amg.py:156-176."""
    inter = (logits > (thr + off)).sum((-1, -2)).astype(np.int64)
    union = (logits > (thr - off)).sum((-1, -2)).astype(np.int64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / union).astype(F32), inter, union


def mask_to_box(masks):
    """batched_mask_to_box (utils/amg.py:303-346): XYXY inclusive, [0,0,0,0] for empty."""
    out = np.zeros((len(masks), 4), dtype=np.int64)
    for i, m in enumerate(masks):
        ys, xs = np.nonzero(m)
        if len(ys):
            out[i] = [xs.min(), ys.min(), xs.max(), ys.max()]
    return out


def box_iou(a, b):
    """torchvision.ops.box_iou on float XYXY boxes."""
    area = lambda x: (x[:, 2] - x[:, 0]) * (x[:, 3] - x[:, 1])
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area(a)[:, None] + area(b)[None, :] - inter)


def nms(boxes, scores, thr):
    """torchvision.ops.nms: greedy, descending score (stable order), suppress IoU > thr.  A NaN score sorts FIRST, as
    torch.sort(descending=True) places it (the order nms takes: scores.sort(0, descending=True))."""
    boxes = boxes.astype(F32)
    sc = scores.astype(np.float64)
    nan = np.isnan(sc)
    order = np.lexsort((-np.where(nan, 0.0, sc), ~nan))      # stable; last key first: NaNs, then descending score
    keep = []
    alive = np.ones(len(boxes), bool)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = box_iou(boxes, boxes)
    for i in order:
        if not alive[i]:
            continue
        keep.append(i)
        alive &= ~(iou[i] > thr)
        alive[i] = False
    return np.array(keep, dtype=np.int64)


def point_grid(n):
    """build_point_grid (utils/amg.py:179-186)."""
    off = 1 / (2 * n)
    p = np.linspace(off, 1 - off, n)
    return np.stack([np.tile(p[None, :], (n, 1)), np.tile(p[:, None], (1, n))], axis=-1).reshape(-1, 2)


def preprocess_shape(oldh, oldw, long_side=1024):
    """ResizeLongestSide.get_preprocess_shape (utils/transforms.py:93-102)."""
    scale = long_side * 1.0 / max(oldh, oldw)
    return int(oldh * scale + 0.5), int(oldw * scale + 0.5)


_PIL_PB = 32 - 8 - 2   # PRECISION_BITS of Pillow's 8-bit resampler


def pil_bilinear_coeffs(in_size, out_size):
    """Pillow ImagingResample precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter
    (support 1.0, anti-aliased when shrinking): int32 weights [out, ksize] and (first, count) bounds."""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 1.0 * fs
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [max(0.0, 1.0 - abs((x + xmin - center + 0.5) / fs)) for x in range(xmax)]
        ww = sum(w)
        for x in range(xmax):
            v = (w[x] / ww if ww != 0.0 else w[x]) * (1 << _PIL_PB)
            kk[xx, x] = int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def pil_bilinear_resize(img, oh, ow):
    """ResizeLongestSide.apply_image (utils/transforms.py:26-31) = PIL Image.resize(BILINEAR) on uint8
    HWC: horizontal pass then vertical pass, 22-bit fixed-point weights, uint8 intermediate.
    Bit-exact against Pillow (tests/test_oracle_sam_golden.py)."""
    H, W, C = img.shape
    kx, bx = pil_bilinear_coeffs(W, ow)
    ky, by = pil_bilinear_coeffs(H, oh)
    tmp = np.zeros((H, ow, C), dtype=np.uint8)
    for xx in range(ow):
        xmin, n = bx[xx]
        acc = np.full((H, C), 1 << (_PIL_PB - 1), dtype=np.int64)
        for x in range(n):
            acc += img[:, xmin + x, :].astype(np.int64) * int(kx[xx, x])
        tmp[:, xx, :] = np.clip(acc >> _PIL_PB, 0, 255)
    out = np.zeros((oh, ow, C), dtype=np.uint8)
    for yy in range(oh):
        ymin, n = by[yy]
        acc = np.full((ow, C), 1 << (_PIL_PB - 1), dtype=np.int64)
        for y in range(n):
            acc += tmp[ymin + y].astype(np.int64) * int(ky[yy, y])
        out[yy] = np.clip(acc >> _PIL_PB, 0, 255)
    return out


def remove_small_regions(mask, area_thresh, mode):
    """utils/amg.py:267-291 with scipy.ndimage.label (8-connectivity) in place of
    cv2.connectedComponentsWithStats (parity unpinned: OpenCV absent offline)."""
    from scipy import ndimage
    correct_holes = mode == "holes"
    working = (correct_holes ^ mask).astype(np.uint8)
    regions, n = ndimage.label(working, structure=np.ones((3, 3), int))
    sizes = np.bincount(regions.ravel(), minlength=n + 1)[1:]
    small = [i + 1 for i, s in enumerate(sizes) if s < area_thresh]
    if not small:
        return mask, False
    fill = [0] + small
    if not correct_holes:
        fill = [i for i in range(n + 1) if i not in fill]
        if not fill:
            fill = [int(np.argmax(sizes)) + 1]
    return np.isin(regions, fill), True


def amg_filter(logits_full, iou_pred, pred_iou_thresh=0.7, stab_thresh=0.7, nms_thresh=0.7,
               min_area=0):
    """_process_batch filters + NMS + postprocess_small_regions for the single full-image crop
    (automatic_mask_generator.py:251-257,287-372).  logits_full: [K,H,W] at the original size,
    iou_pred: [K].  Returns (kept candidate indices in output order, masks, boxes XYXY, stability)."""
    K = len(iou_pred)
    idx = np.arange(K)
    if pred_iou_thresh > 0.0:        # :287-290: the filters exist only for positive thresholds
        idx = idx[iou_pred > pred_iou_thresh]
    stab, _, _ = stability_score(logits_full[idx])
    if stab_thresh > 0.0:            # :293-298
        keep = stab >= stab_thresh
        idx, stab = idx[keep], stab[keep]
    masks = logits_full[idx] > 0
    boxes = mask_to_box(masks)
    k = nms(boxes, iou_pred[idx], nms_thresh)
    idx, stab, masks, boxes = idx[k], stab[k], masks[k], boxes[k]
    if min_area > 0 and len(idx):
        new, scores = [], []
        for m in masks:
            m1, c1 = remove_small_regions(m, min_area, "holes")
            m2, c2 = remove_small_regions(m1, min_area, "islands")
            new.append(m2)
            scores.append(float(not (c1 or c2)))
        new = np.stack(new)
        nb = mask_to_box(new)
        k = nms(nb, np.array(scores), nms_thresh)
        for i in k:
            if scores[i] == 0.0:
                masks[i], boxes[i] = new[i], nb[i]
        idx, stab, masks, boxes = idx[k], stab[k], masks[k], boxes[k]
    return idx, masks, boxes, stab
