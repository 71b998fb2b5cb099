"""ORACLE (test infrastructure, NOT product code): numpy fp32 restatement of the reference's
CLIP hybrid encoder, text encoder and scoring helpers.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The shipped path (hybridgl_amd/) never does: it calls libhybridgl.so and has no CPU fallback.

Parity pin: tests/test_oracle_golden.py checks every function here against fixtures in
tests/golden/ that were produced by importing the reference itself (oracle/gen_golden.py).

Each function cites the reference lines it restates (paths relative to the reference tree).
"""
import math

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------- basic layers
def layer_norm(x, w, b, eps=1e-5):
    """clip/model.py:189-195 (nn.LayerNorm evaluated in fp32)."""
    x = x.astype(F32, copy=False)
    mean = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = x - mean
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps))) * w + b


def quick_gelu(x):
    """clip/model.py:198-200: x * sigmoid(1.702 x)."""
    return x / (F32(1.0) + np.exp(F32(-1.702) * x))


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, F32(0))
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True, dtype=F32)


def linear(x, w, b=None):
    y = x @ w.T
    if b is not None:
        y = y + b
    return y.astype(F32, copy=False)


def multi_head_attention(x, in_w, in_b, out_w, out_b, heads, add_mask=None):
    """nn.MultiheadAttention(x,x,x, attn_mask) (clip/model.py:209,220-229). x: [B,S,D].

    add_mask: additive fp32 mask broadcastable to [B, heads, S, S] (0 / -inf)."""
    B, S, D = x.shape
    hd = D // heads
    qkv = linear(x, in_w, in_b)  # [B,S,3D]
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    q = q.reshape(B, S, heads, hd).transpose(0, 2, 1, 3) * F32(hd ** -0.5)
    k = k.reshape(B, S, heads, hd).transpose(0, 2, 1, 3)
    v = v.reshape(B, S, heads, hd).transpose(0, 2, 1, 3)
    att = q @ k.transpose(0, 1, 3, 2)  # [B,h,S,S]
    if add_mask is not None:
        att = att + add_mask
    att = softmax(att.astype(F32), axis=-1)
    o = (att @ v).transpose(0, 2, 1, 3).reshape(B, S, D)
    return linear(o, out_w, out_b)


def resblock(x, sd, prefix, heads, add_mask=None):
    """ResidualAttentionBlock.forward (clip/model.py:244-257)."""
    g = lambda k: sd[f"{prefix}.{k}"]
    h = layer_norm(x, g("ln_1.weight"), g("ln_1.bias"))
    x = x + multi_head_attention(h, g("attn.in_proj_weight"), g("attn.in_proj_bias"),
                                 g("attn.out_proj.weight"), g("attn.out_proj.bias"), heads, add_mask)
    h = layer_norm(x, g("ln_2.weight"), g("ln_2.bias"))
    h = quick_gelu(linear(h, g("mlp.c_fc.weight"), g("mlp.c_fc.bias")))
    return x + linear(h, g("mlp.c_proj.weight"), g("mlp.c_proj.bias"))


# ----------------------------------------------------------------------------- resampling
def _src_index(out_size, in_size):
    """ATen area_pixel_compute_source_index, align_corners=False (UpSample.h)."""
    scale = F32(in_size) / F32(out_size)
    # the compiled ATen kernels contract scale*(dst+0.5)-0.5 into ONE fma (checked against
    # F.interpolate in tests/golden/resize.npz); emulate it exactly through float64
    d = (np.arange(out_size, dtype=F32) + F32(0.5)).astype(np.float64)
    src = (np.float64(scale) * d - 0.5).astype(F32)
    src = np.maximum(src, F32(0))
    i0 = src.astype(np.int64)
    i1 = i0 + (i0 < in_size - 1)
    l1 = (src - i0.astype(F32)).astype(F32)
    return i0, i1, (F32(1) - l1).astype(F32), l1


def bilinear_resize(x, oh, ow):
    """F.interpolate(x, (oh,ow), mode='bilinear', align_corners=False, antialias=False) on the
    last two dims == torchvision 0.15 TF.resize on a tensor (model/backbone.py:160,
    Hybridgl_main.py:116,121)."""
    x = x.astype(F32, copy=False)
    H, W = x.shape[-2:]
    y0, y1, ly0, ly1 = _src_index(oh, H)
    x0, x1, lx0, lx1 = _src_index(ow, W)
    top = x[..., y0, :]
    bot = x[..., y1, :]
    t = top[..., x0] * lx0 + top[..., x1] * lx1
    b = bot[..., x0] * lx0 + bot[..., x1] * lx1
    return (t * ly0[:, None] + b * ly1[:, None]).astype(F32)


# ----------------------------------------------------------------------------- CLIP vision
def vit_embed(sd, imgs):
    """conv1 -> flatten -> cat(cls) -> +pos -> ln_pre (model/backbone.py:130-139)."""
    w = sd["visual.conv1.weight"]
    D, _, p, _ = w.shape
    N, C, R, _ = imgs.shape
    g = R // p
    cols = imgs.reshape(N, C, g, p, g, p).transpose(0, 2, 4, 1, 3, 5).reshape(N, g * g, C * p * p)
    tok = cols.astype(F32) @ w.reshape(D, -1).T
    cls = np.broadcast_to(sd["visual.class_embedding"], (N, 1, D))
    x = np.concatenate([cls, tok], axis=1) + sd["visual.positional_embedding"]
    return layer_norm(x.astype(F32), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])


def make_attn_mask(pm, heads):
    """CLIPViTFM.make_attn_mask (model/backbone.py:108-115) as an additive mask
    [N,1,S,S]: only the CLS query row is restricted, to patches with pm != 0."""
    N, P = pm.shape
    S = P + 1
    m = np.zeros((N, 1, S, S), dtype=F32)
    m[:, 0, 0, 1:] = np.where(pm != 0, F32(0), F32(-np.inf))
    return m


def token_mask(x, pm):
    """cat(cls, x[1:] * pm) (model/backbone.py:235-247)."""
    y = x.copy()
    y[:, 1:, :] = x[:, 1:, :] * pm[:, :, None]
    return y


def vit_head(sd, x):
    """ln_post(x[:,0]) @ proj (model/backbone.py:254-260)."""
    h = layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    return (h @ sd["visual.proj"]).astype(F32)


def clip_hybrid_forward(sd, local_imgs, global_imgs, pred_masks, masking_block=None,
                        fusion_mode="G2L", last_layer=10, heads=None):
    """CLIPViTFM.forward (model/backbone.py:117-309), all six fusion modes. Returns [N, embed]."""
    if masking_block is None:
        masking_block = last_layer
    D = sd["visual.conv1.weight"].shape[0]
    if heads is None:
        heads = D // 64  # clip/model.py:333
    layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    blk = lambda i, x, m=None: resblock(x, sd, f"visual.transformer.resblocks.{i}", heads, m)

    x = vit_embed(sd, local_imgs)
    if fusion_mode == "crop":  # :126-128
        for i in range(layers):
            x = blk(i, x)
        return vit_head(sd, x)
    x2 = vit_embed(sd, global_imgs) if global_imgs is not None else None
    N = x.shape[0]
    g = int(round(math.sqrt(x.shape[1] - 1)))
    pm = bilinear_resize(pred_masks.astype(F32), g, g).reshape(N, g * g)  # :160
    amask = make_attn_mask(pm, heads)

    if fusion_mode == "token_masking":  # :161-184
        for i in range(layers):
            if i >= masking_block:
                x = blk(i, token_mask(x, pm))
                if i == last_layer + 1:
                    return vit_head(sd, x)
            else:
                x = blk(i, x)
        return x
    if fusion_mode == "attn_masking":  # :186-204
        for i in range(layers):
            if i >= masking_block:
                x = blk(i, x, amask)
                if i == last_layer:
                    return vit_head(sd, x)
            else:
                x = blk(i, x)
        return x
    if fusion_mode == "L2G":  # :206-225
        for i in range(layers):
            if i >= masking_block:
                x_ori_local = x.copy()
                x = blk(i, x)
                x2 = blk(i, x_ori_local + x2 * F32(2), amask)
            else:
                x, x2 = blk(i, x), blk(i, x2)
            if i == last_layer + 1:
                return vit_head(sd, x2)
        return x
    if fusion_mode == "G2L":  # :227-260
        for i in range(layers):
            if i >= masking_block:
                xg = token_mask(x2, pm)
                x = blk(i, xg * F32(2) + x)
                x2 = blk(i, x2, amask)
            else:
                x, x2 = blk(i, x), blk(i, x2)
            if i == last_layer + 1:
                return vit_head(sd, x)
        return x
    if fusion_mode == "G2L&L2G":  # :262-306
        hl = hg = None
        for i in range(layers):
            if i >= masking_block:
                if i == masking_block:
                    hl, hg = x.copy(), x2.copy()
                x_ori_local = x.copy()
                xg = token_mask(x2, pm)
                x = blk(i, x)
                x2 = blk(i, x2, amask)
                hl = blk(i, hl + F32(2) * xg)
                hg = blk(i, x_ori_local + F32(2) * hg, amask)
            else:
                x, x2 = blk(i, x), blk(i, x2)
            if i == last_layer + 1:
                return vit_head(sd, hl) + vit_head(sd, hg)
        return x
    raise ValueError(fusion_mode)


# ----------------------------------------------------------------------------- CLIP text
def encode_text(sd, tokens, heads=None, target_noun_index=None, masking_index=(), masking_block=None):
    """CLIP.encode_text (clip/model.py:414-431); causal mask from build_attention_mask (:396-402).
    target_noun_index: the pooled position is target_noun_index + 1 instead of the EOT (:426-428; None and 0 are
    falsy there and fall through to the EOT).  masking_index / masking_block: CLIPViTFM.text_masking_feature
    (model/backbone.py:34-56): positions masking_index + 1 of every sequence are zeroed before each block >= masking_block."""
    tokens = np.asarray(tokens)
    B, S = tokens.shape
    D = sd["ln_final.weight"].shape[0]
    if heads is None:
        heads = D // 64
    layers = len(set(k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")))
    x = (sd["token_embedding.weight"][tokens] + sd["positional_embedding"]).astype(F32)
    causal = np.triu(np.full((S, S), -np.inf, dtype=F32), 1)[None, None]
    zero = [int(i) + 1 for i in masking_index]
    for i in range(layers):
        if zero and masking_block is not None and i >= masking_block:
            x = x.copy()
            x[:, zero] = 0
        x = resblock(x, sd, f"transformer.resblocks.{i}", heads, causal)
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    eot = tokens.argmax(axis=-1)
    if target_noun_index:
        eot = np.full(B, int(target_noun_index) + 1)
    return (x[np.arange(B), eot] @ sd["text_projection"]).astype(F32)


# ----------------------------------------------------------------------------- scoring
def calculate_score(img, txt, logit_scale):
    """CLIPViTFM.calculate_score (model/backbone.py:74-87)."""
    img = img / np.linalg.norm(img, axis=1, keepdims=True).astype(F32)
    txt = txt / np.linalg.norm(txt, axis=1, keepdims=True).astype(F32)
    return ((F32(logit_scale) * img) @ txt.T).astype(F32)


def torch_linspace(start, end, steps):
    """torch.linspace fp32 (ATen RangeFactories.cpp: symmetric fill)."""
    if steps == 1:
        return np.array([start], dtype=F32)
    start, end = F32(start), F32(end)
    step = (end - start) / F32(steps - 1)
    i = np.arange(steps)
    half = steps // 2
    # ATen evaluates start + step*i / end - step*(n-1-i) as single fmas (checked bit-exactly
    # against torch.linspace through gen_dir_mask in tests/golden/scoring.npz)
    lo = (np.float64(start) + np.float64(step) * i).astype(F32)
    hi = (np.float64(end) - np.float64(step) * (steps - i - 1)).astype(F32)
    return np.where(i < half, lo, hi).astype(F32)


def gen_dir_mask(dirflag, height, width):
    """utils.py:135-161 (up/down are commented out there -> ones)."""
    if dirflag == "left":
        row = torch_linspace(1, 0, width)
    elif dirflag == "right":
        row = torch_linspace(0, 1, width)
    elif dirflag == "middle":
        row = np.concatenate([torch_linspace(0, 1, width // 2), torch_linspace(1, 0, width - width // 2)])
    else:
        row = np.ones(width, dtype=F32)
    return np.broadcast_to(row, (height, width)).astype(F32)


def coherence_scores(imgattn, masks, dirflag="none", black=1.8):
    """Hybridgl_main.py:203-223 -- min-max, direction mask, /mean, then the per-mask score.
    Sums are taken in float64 (the reference's fp32 torch.sum differs by ~1e-6 relative)."""
    a = imgattn.astype(F32)
    a = (a - a.min()) / (a.max() - a.min())
    a = a * gen_dir_mask(dirflag, a.shape[0], a.shape[1])
    a = (a / a.mean(dtype=F32)).astype(F32)
    out = []
    for m in masks:
        m = (m != 0)
        cnt = m.sum()
        s_in = (a.astype(np.float64) * (2 - F32(black)))[m].sum() / cnt if cnt else np.nan
        n_out = m.size - cnt
        s_out = (a.astype(np.float64) * F32(black))[~m].sum() / n_out if n_out else np.nan
        out.append(s_in - s_out)
    return np.asarray(out, dtype=F32)


def compute_iou(pred, target):
    """Compute_IoU (utils.py:365-384): returns (I, U) integer counts."""
    p, t = pred != 0, target != 0
    return int(np.logical_and(p, t).sum()), int(np.logical_or(p, t).sum())


def relation_boxes(bi, bj, si, sj, rela):
    """utils.py:240-268; boxes are integer XYWH; integer/2 is fp32 true division."""
    f = F32
    si, sj = f(si), f(sj)
    if rela == "left":
        return si * sj * f((f(bi[0]) + f(bi[2]) / f(2)) < (f(bj[0]) + f(bj[2]) / f(2)))
    if rela == "right":
        return si * sj * f((f(bi[0]) + f(bi[2]) / f(2)) > (f(bj[0]) + f(bj[2]) / f(2)))
    if rela == "up":
        return si * sj * f((f(bi[1]) + f(bi[3]) / f(2)) < (f(bj[1]) + f(bj[3]) / f(2)))
    if rela == "down":
        return si * sj * f((f(bi[1]) + f(bi[3]) / f(2)) > (f(bj[1]) + f(bj[3]) / f(2)))
    if rela == "big":
        return si * sj * f(int(bi[2]) * int(bi[3]) > int(bj[2]) * int(bj[3]))
    if rela == "small":
        return si * sj * f(int(bi[2]) * int(bi[3]) < int(bj[2]) * int(bj[3]))
    if rela == "within":
        x1 = max(int(bi[0]), int(bj[0]))
        x2 = max(x1, min(int(bi[0] + bi[2]), int(bj[0] + bj[2])))
        y1 = max(int(bi[1]), int(bj[1]))
        y2 = max(y1, min(int(bi[1] + bi[3]), int(bj[1] + bj[3])))
        return si * sj * f(x2 - x1) * f(y2 - y1) / f(int(bi[2]) * int(bi[3]))
    return si  # "none" and anything else


def topk_desc(v, k):
    """torch.topk(v, k) indices: descending, lowest index first among equal values."""
    order = np.lexsort((np.arange(len(v)), -v.astype(np.float64)))
    return order[:k]


def score_sentence(hybrid, text_ensemble, neg_text, boxes, gem_score, logit_scale=100.0, k1=3, k2=6,
                   alpha=0.6, rela="none", has_other_nouns=False):
    """Hybridgl_main.py:153-196,225-228.  Returns (idx_pure, idx_final, score_clip, score_neg)."""
    sc = calculate_score(hybrid, text_ensemble.reshape(1, -1), logit_scale)[:, 0]
    sn = calculate_score(hybrid, neg_text.reshape(1, -1), logit_scale)[:, 0]
    idx_pure = int(np.argmax(sc))
    p, pn = softmax(sc, 0), softmax(sn, 0)
    k1, k2 = min(k1, len(p)), min(k2, len(pn))
    top, topn = topk_desc(p, k1), topk_desc(pn, k2)
    # `np.float64 + torch.Tensor` defers to Tensor.__radd__: every partial sum is an fp32 tensor
    # (Hybridgl_main.py:183-193), verified against the reference in tests/golden/scoring.npz
    ts = np.zeros(k1, dtype=F32)
    for i in range(k1):
        if not has_other_nouns:
            for j in top:
                ts[i] = F32(ts[i] + relation_boxes(boxes[top[i]], boxes[j], p[top[i]], p[j], rela))
        else:
            for j in topn:
                ts[i] = F32(ts[i] + relation_boxes(boxes[top[i]], boxes[j], p[top[i]], pn[j], rela))
    tsf = softmax(ts, 0)
    blend = tsf * F32(1 - alpha) + F32(alpha) * gem_score[top]
    return idx_pure, int(top[int(np.argmax(blend))]), sc, sn


# ----------------------------------------------------------------------------- image synthesis
CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], dtype=F32)
IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=F32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=F32)


def synthesize_views(sam_img, blurred, image_norm, masks, res=224):
    """Hybridgl_main.py:93-125: per mask the mean-filled local view and the
    blurred-background global view, both bilinear (no antialias) to res x res.
    sam_img/blurred: [H,W,3] uint8; image_norm: [3,H,W] fp32; masks: [N,H,W]."""
    loc, glo = [], []
    for m in masks:
        m = (m != 0)
        comp = np.where(m[:, :, None], sam_img, blurred)  # cv2.add(sharp_region, blurred_region)
        g = (comp.astype(F32) / F32(255)).transpose(2, 0, 1)  # T.ToTensor
        g = bilinear_resize(g, res, res)
        g = (g - IMAGENET_MEAN[:, None, None]) / IMAGENET_STD[:, None, None]
        mf = m.astype(F32)[None]
        l = image_norm * mf + (F32(1) - mf) * CLIP_MEAN[:, None, None]
        loc.append(bilinear_resize(l, res, res))
        glo.append(g.astype(F32))
    return np.stack(loc), np.stack(glo)
