"""ORACLE (test infrastructure, NOT product code): the heavy stages of the reference path restated in plain PyTorch on the
CPU -- the operators the reference itself runs when `device = "cpu"` (Hybridgl_main.py:30-34): nn.Linear / LayerNorm /
softmax / matmul in fp32 on torch's intra-op thread pool.  It exists for ONE purpose: bench.py's `cpu_baseline` leg times it on
the GPU box's host cores (the reference cannot travel there), unsampled where the work is -- the CLIP hybrid encoder on all
64 masks, all 32 blocks of the SAM ViT-H encoder, the mask decoder on all 64 prompts, the text encoder.

Only tests/ and bench.py's cpu_baseline leg may import this file.  Pinned by tests/test_torch_cpu_baseline.py against the
numpy oracle (oracle/clip_oracle.py, oracle/sam_oracle.py), which is pinned against outputs of the imported reference.
Each function cites the reference lines it restates.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import sam_oracle as S


def to_torch(sd):
    """numpy state_dict -> torch tensors sharing the memory (fp32, CPU)."""
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


# ----------------------------------------------------------------------------- CLIP (clip/model.py)
def _mha(x, sd, p, heads, add_mask=None):
    """nn.MultiheadAttention(x, x, x, attn_mask) (clip/model.py:209, 220-229); x [B,S,D], add_mask broadcastable to [B,h,S,S]."""
    B, S_, D = x.shape
    hd = D // heads
    qkv = F.linear(x, sd[f"{p}.in_proj_weight"], sd[f"{p}.in_proj_bias"])
    q, k, v = (t.reshape(B, S_, heads, hd).transpose(1, 2) for t in qkv.split(D, dim=-1))
    att = (q * hd ** -0.5) @ k.transpose(-1, -2)
    if add_mask is not None:
        att = att + add_mask
    o = (torch.softmax(att, dim=-1) @ v).transpose(1, 2).reshape(B, S_, D)
    return F.linear(o, sd[f"{p}.out_proj.weight"], sd[f"{p}.out_proj.bias"])


def resblock(x, sd, p, heads, add_mask=None):
    """ResidualAttentionBlock.forward (clip/model.py:244-257) with QuickGELU (:198-200)."""
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), sd[f"{p}.ln_1.weight"], sd[f"{p}.ln_1.bias"], 1e-5)
    x = x + _mha(h, sd, f"{p}.attn", heads, add_mask)
    h = F.layer_norm(x, (D,), sd[f"{p}.ln_2.weight"], sd[f"{p}.ln_2.bias"], 1e-5)
    h = F.linear(h, sd[f"{p}.mlp.c_fc.weight"], sd[f"{p}.mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h)
    return x + F.linear(h, sd[f"{p}.mlp.c_proj.weight"], sd[f"{p}.mlp.c_proj.bias"])


def vit_embed(sd, imgs):
    """conv1 -> flatten -> cat(cls) -> + pos -> ln_pre (model/backbone.py:130-139)."""
    w = sd["visual.conv1.weight"]
    D, p = w.shape[0], w.shape[2]
    tok = F.conv2d(imgs, w, stride=p).flatten(2).transpose(1, 2)
    cls = sd["visual.class_embedding"].expand(tok.shape[0], 1, D)
    x = torch.cat([cls, tok], 1) + sd["visual.positional_embedding"]
    return F.layer_norm(x, (D,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5)


def vit_head(sd, x):
    """ln_post(x[:, 0]) @ proj (model/backbone.py:254-260)."""
    D = x.shape[-1]
    return F.layer_norm(x[:, 0], (D,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5) @ sd["visual.proj"]


def clip_hybrid_forward(sd, local_imgs, global_imgs, pred_masks, masking_block=None, fusion_mode="G2L", last_layer=10):
    """CLIPViTFM.forward (model/backbone.py:117-309), modes G2L (:227-260), L2G (:206-225), G2L&L2G (:262-306).
    local_imgs / global_imgs [N,3,R,R] fp32, pred_masks [N,H,W] bool -> [N, embed]."""
    if masking_block is None:
        masking_block = last_layer
    D = sd["visual.conv1.weight"].shape[0]
    heads = D // 64
    layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    blk = lambda i, x, m=None: resblock(x, sd, f"visual.transformer.resblocks.{i}", heads, m)
    x, x2 = vit_embed(sd, local_imgs), vit_embed(sd, global_imgs)
    N, g = x.shape[0], int(round(math.sqrt(x.shape[1] - 1)))
    # TF.resize on a tensor = bilinear, align_corners=False, no antialias (torchvision 0.15; model/backbone.py:160)
    pm = F.interpolate(pred_masks.float()[:, None], (g, g), mode="bilinear", align_corners=False)[:, 0].reshape(N, g * g)
    amask = torch.zeros(N, 1, g * g + 1, g * g + 1)
    amask[:, 0, 0, 1:] = torch.where(pm != 0, 0.0, float("-inf"))                  # make_attn_mask (:108-115)
    tokm = lambda t: torch.cat([t[:, :1], t[:, 1:] * pm[:, :, None]], 1)             # :235-247
    hl = hg = None
    for i in range(layers):
        if i < masking_block:
            x, x2 = blk(i, x), blk(i, x2)
        elif fusion_mode == "G2L":
            x, x2 = blk(i, tokm(x2) * 2 + x), blk(i, x2, amask)
        elif fusion_mode == "L2G":
            x, x2 = blk(i, x), blk(i, x + x2 * 2, amask)
        elif fusion_mode == "G2L&L2G":
            if i == masking_block:
                hl, hg = x.clone(), x2.clone()
            xo, xg = x, tokm(x2)
            x, x2 = blk(i, x), blk(i, x2, amask)
            hl, hg = blk(i, hl + 2 * xg), blk(i, xo + 2 * hg, amask)
        else:
            raise ValueError(fusion_mode)
        if i == last_layer + 1:
            if fusion_mode == "G2L":
                return vit_head(sd, x)
            if fusion_mode == "L2G":
                return vit_head(sd, x2)
            return vit_head(sd, hl) + vit_head(sd, hg)
    return x


# ----------------------------------------------------------------------------- GEM (gem_torch 1.0.1, restated in gem_oracle.py)
def _ss_attention(x, sd, p, heads):
    """gem SelfSelfAttention.forward on x = ln_1(tokens) [B, S, D] -> (x_gem, x_ori) after the out-projection
    (oracle/gem_oracle.py self_self_attention: ss_attn_iter = 1, temperature from the mean token norm)."""
    B, S_, D = x.shape
    hd = D // heads
    scale = hd ** -0.5
    qkv = F.linear(x, sd[f"{p}.attn.in_proj_weight"], sd[f"{p}.attn.in_proj_bias"])
    q, k, v = (t.reshape(B, S_, heads, hd).transpose(1, 2) for t in qkv.split(D, dim=-1))
    out = lambda t: F.linear(t.transpose(1, 2).reshape(B, S_, D), sd[f"{p}.attn.out_proj.weight"], sd[f"{p}.attn.out_proj.bias"])
    x_ori = out(torch.softmax((q @ k.transpose(-1, -2)) * scale, dim=-1) @ v)
    inv_temp = (x.norm(dim=-1).mean(dim=-1) * scale).reshape(B, 1, 1, 1)
    acc = 0
    for xs in (v, k, q):
        xs = F.normalize(xs, dim=-1)
        xs = torch.softmax((xs @ xs.transpose(-1, -2)) * inv_temp, dim=-1) @ xs
        xs = F.normalize(xs, dim=-1)
        acc = acc + torch.softmax((xs @ xs.transpose(-1, -2)) * inv_temp, dim=-1) @ v
    return out(acc / 3), x_ori


def gem_vit_forward(sd, imgs, pos, gem_depth=7, heads=None):
    """GEMViT.forward (gem_oracle.gem_vit_forward): imgs [B, 3, R, R] torch, pos = the position embedding already interpolated to
    the (R / patch)^2 grid (gem_oracle.interpolate_pos_encoding: a table computed once per resolution) -> feat_gem [B, S, E]."""
    w = sd["visual.conv1.weight"]
    D = w.shape[0]
    heads = heads or D // 64
    layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    x = vit_embed(dict(sd, **{"visual.positional_embedding": pos}), imgs)
    n_gem = max(0, min(layers, gem_depth - 1))
    x_gem = None
    for i in range(layers):
        p = f"visual.transformer.resblocks.{i}"
        if i < layers - n_gem:
            x = resblock(x, sd, p, heads)
            continue
        if x_gem is None:
            x_gem = x
        r_gem, r_ori = _ss_attention(F.layer_norm(x, (D,), sd[f"{p}.ln_1.weight"], sd[f"{p}.ln_1.bias"], 1e-5), sd, p, heads)
        x = x + r_ori
        h = F.layer_norm(x, (D,), sd[f"{p}.ln_2.weight"], sd[f"{p}.ln_2.bias"], 1e-5)
        h = F.linear(h, sd[f"{p}.mlp.c_fc.weight"], sd[f"{p}.mlp.c_fc.bias"])
        x = x + F.linear(h * torch.sigmoid(1.702 * h), sd[f"{p}.mlp.c_proj.weight"], sd[f"{p}.mlp.c_proj.bias"])
        x_gem = x_gem + r_gem
    if x_gem is None:
        x_gem = x
    return F.layer_norm(x_gem, (D,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5) @ sd["visual.proj"]


def gem_heatmap(feat, text, res):
    """GEMWrapper.forward after the encoders (gem_oracle.gem_heatmap): feat [S, E] (row 0 = CLS), text [T, E] -> [T, res, res]."""
    f = F.normalize(feat[1:], dim=-1)
    t = F.normalize(text, dim=-1)
    g = int(round(math.sqrt(f.shape[0])))
    m = (100.0 * (f @ t.T)).T.reshape(1, -1, g, g)
    up = F.interpolate(m, size=(res, res), mode="bilinear", align_corners=False)[0]
    mn = up.flatten(1).min(dim=1)[0][:, None, None]
    mx = up.flatten(1).max(dim=1)[0][:, None, None]
    return (up - mn) / (mx - mn)


def encode_text(sd, tokens):
    """CLIP.encode_text (clip/model.py:414-431), causal mask of :396-402.  tokens [B,77] int -> [B, embed]."""
    tokens = torch.as_tensor(np.asarray(tokens)).long()
    B, S_ = tokens.shape
    D = sd["ln_final.weight"].shape[0]
    layers = len(set(k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")))
    x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"]
    causal = torch.full((S_, S_), float("-inf")).triu_(1)[None, None]
    for i in range(layers):
        x = resblock(x, sd, f"transformer.resblocks.{i}", D // 64, causal)
    x = F.layer_norm(x, (D,), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
    return x[torch.arange(B), tokens.argmax(-1)] @ sd["text_projection"]


# ----------------------------------------------------------------------------- SAM (segment_anything/modeling)
def _enc_attention(x, sd, p, heads):
    """Attention.forward + add_decomposed_rel_pos (image_encoder.py:224-240, 325-361); x [B,H,W,D]."""
    B, H, W, D = x.shape
    hd = D // heads
    qkv = F.linear(x.reshape(B, H * W, D), sd[f"{p}.qkv.weight"], sd[f"{p}.qkv.bias"])
    q, k, v = qkv.reshape(B, H * W, 3, heads, hd).permute(2, 0, 3, 1, 4).reshape(3, B * heads, H * W, hd).unbind(0)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    idx = torch.arange(H)[:, None] - torch.arange(H)[None, :] + (H - 1)          # get_rel_pos (:292-322), q_size == k_size
    Rh, Rw = sd[f"{p}.rel_pos_h"][idx], sd[f"{p}.rel_pos_w"][idx]
    rq = q.reshape(B * heads, H, W, hd)
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
    attn = (attn.view(B * heads, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(B * heads, H * W, H * W)
    o = (torch.softmax(attn, dim=-1) @ v).view(B, heads, H, W, hd).permute(0, 2, 3, 1, 4).reshape(B, H, W, D)
    return F.linear(o, sd[f"{p}.proj.weight"], sd[f"{p}.proj.bias"])


def encoder_block(x, sd, p, heads, window):
    """Block.forward (image_encoder.py:166-182) with window_partition / window_unpartition (:243-290); eps 1e-6."""
    D = x.shape[-1]
    short = x
    x = F.layer_norm(x, (D,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], 1e-6)
    if window > 0:
        B, H, W, _ = x.shape
        ph, pw = (window - H % window) % window, (window - W % window) % window
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
        Hp, Wp = H + ph, W + pw
        x = x.view(B, Hp // window, window, Wp // window, window, D).permute(0, 1, 3, 2, 4, 5).reshape(-1, window, window, D)
    x = _enc_attention(x, sd, f"{p}.attn", heads)
    if window > 0:
        x = x.view(B, Hp // window, Wp // window, window, window, D).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, D)[:, :H, :W]
    x = short + x
    h = F.layer_norm(x, (D,), sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], 1e-6)
    h = F.gelu(F.linear(h, sd[f"{p}.mlp.lin1.weight"], sd[f"{p}.mlp.lin1.bias"]))
    return x + F.linear(h, sd[f"{p}.mlp.lin2.weight"], sd[f"{p}.mlp.lin2.bias"])


def image_encoder(sd, img, cfg):
    """ImageEncoderViT.forward (image_encoder.py:106-116); img [3,S,S] fp32 -> [g, g, out_chans] (NHWC)."""
    p = "image_encoder"
    w = sd[f"{p}.patch_embed.proj.weight"]
    D, ps = w.shape[0], w.shape[2]
    x = F.conv2d(img[None], w, sd[f"{p}.patch_embed.proj.bias"], stride=ps).permute(0, 2, 3, 1) + sd[f"{p}.pos_embed"]
    for i in range(cfg["depth"]):
        x = encoder_block(x, sd, f"{p}.blocks.{i}", cfg["num_heads"], 0 if i in cfg["global_attn_indexes"] else cfg["window_size"])
    x = x.permute(0, 3, 1, 2)
    ln2d = lambda t, n: F.layer_norm(t.permute(0, 2, 3, 1), (t.shape[1],), sd[f"{n}.weight"], sd[f"{n}.bias"], 1e-6).permute(0, 3, 1, 2)
    x = ln2d(F.conv2d(x, sd[f"{p}.neck.0.weight"]), f"{p}.neck.1")
    x = ln2d(F.conv2d(x, sd[f"{p}.neck.2.weight"], padding=1), f"{p}.neck.3")
    return x[0].permute(1, 2, 0)


def _dec_attention(sd, p, q, k, v, heads):
    """transformer.py:185-240 Attention (internal dim = C / downsample_rate)."""
    q = F.linear(q, sd[f"{p}.q_proj.weight"], sd[f"{p}.q_proj.bias"])
    k = F.linear(k, sd[f"{p}.k_proj.weight"], sd[f"{p}.k_proj.bias"])
    v = F.linear(v, sd[f"{p}.v_proj.weight"], sd[f"{p}.v_proj.bias"])
    B, Nq, C = q.shape
    hd = C // heads
    sep = lambda t: t.view(t.shape[0], t.shape[1], heads, hd).transpose(1, 2)
    a = torch.softmax(sep(q) @ sep(k).transpose(-1, -2) / math.sqrt(hd), dim=-1)
    return F.linear((a @ sep(v)).transpose(1, 2).reshape(B, Nq, C), sd[f"{p}.out_proj.weight"], sd[f"{p}.out_proj.bias"])


def mask_decoder(sd, emb_nhwc, sparse):
    """MaskDecoder.forward / predict_masks with the TwoWayTransformer (mask_decoder.py:71-149, transformer.py:62-182);
    emb_nhwc [h,w,C], sparse [B,P,C] -> (low-res logits [B,3,4h,4w], iou [B,3])."""
    h, w, C = emb_nhwc.shape
    B = sparse.shape[0]
    ln = lambda x, n: F.layer_norm(x, (x.shape[-1],), sd[f"{n}.weight"], sd[f"{n}.bias"], 1e-5)
    out_tok = torch.cat([sd["mask_decoder.iou_token.weight"], sd["mask_decoder.mask_tokens.weight"]], 0)
    tokens = torch.cat([out_tok.expand(B, -1, -1), sparse], 1)
    keys = (emb_nhwc.reshape(1, h * w, C) + sd["prompt_encoder.no_mask_embed.weight"].reshape(1, 1, C)).expand(B, -1, -1)
    key_pe = torch.from_numpy(S.dense_pe({k: v.numpy() for k, v in sd.items() if k.startswith("prompt_encoder.pe_layer")}, h, w))
    p = "mask_decoder.transformer"
    queries = point_pe = tokens
    for i in range(2):
        l = f"{p}.layers.{i}"
        if i == 0:
            queries = _dec_attention(sd, f"{l}.self_attn", queries, queries, queries, 8)
        else:
            q = queries + point_pe
            queries = queries + _dec_attention(sd, f"{l}.self_attn", q, q, queries, 8)
        queries = ln(queries, f"{l}.norm1")
        queries = ln(queries + _dec_attention(sd, f"{l}.cross_attn_token_to_image", queries + point_pe, keys + key_pe, keys, 8), f"{l}.norm2")
        hmid = F.relu(F.linear(queries, sd[f"{l}.mlp.lin1.weight"], sd[f"{l}.mlp.lin1.bias"]))
        queries = ln(queries + F.linear(hmid, sd[f"{l}.mlp.lin2.weight"], sd[f"{l}.mlp.lin2.bias"]), f"{l}.norm3")
        keys = ln(keys + _dec_attention(sd, f"{l}.cross_attn_image_to_token", keys + key_pe, queries + point_pe, queries, 8), f"{l}.norm4")
    queries = ln(queries + _dec_attention(sd, f"{p}.final_attn_token_to_image", queries + point_pe, keys + key_pe, keys, 8),
                 f"{p}.norm_final_attn")
    iou_tok, mask_tok = queries[:, 0], queries[:, 1:5]
    u = "mask_decoder.output_upscaling"
    x = keys.transpose(1, 2).reshape(B, C, h, w)
    x = F.conv_transpose2d(x, sd[f"{u}.0.weight"], sd[f"{u}.0.bias"], stride=2)
    x = F.gelu(F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), sd[f"{u}.1.weight"], sd[f"{u}.1.bias"], 1e-6).permute(0, 3, 1, 2))
    x = F.gelu(F.conv_transpose2d(x, sd[f"{u}.3.weight"], sd[f"{u}.3.bias"], stride=2))

    def mlp3(pp, t):
        for j in range(3):
            t = F.linear(t, sd[f"{pp}.layers.{j}.weight"], sd[f"{pp}.layers.{j}.bias"])
            if j < 2:
                t = F.relu(t)
        return t
    hyper = torch.stack([mlp3(f"mask_decoder.output_hypernetworks_mlps.{i}", mask_tok[:, i]) for i in range(4)], 1)
    masks = (hyper @ x.flatten(2)).view(B, 4, 4 * h, 4 * w)
    iou = mlp3("mask_decoder.iou_prediction_head", iou_tok)
    return masks[:, 1:], iou[:, 1:]


def postprocess_and_stats(low_res, input_size, original_size, img_size=1024, thr=0.0, off=1.0):
    """Sam.postprocess_masks (modeling/sam.py:133-162) + calculate_stability_score (utils/amg.py:156-176) + the binarised masks'
    boxes (utils/amg.py:303-346) over all candidates; low_res [B,3,h,w] -> (masks bool [B*3,H,W], stability [B*3], boxes [B*3,4])."""
    m = F.interpolate(low_res, (img_size, img_size), mode="bilinear", align_corners=False)
    m = m[..., :input_size[0], :input_size[1]]
    m = F.interpolate(m, tuple(original_size), mode="bilinear", align_corners=False).flatten(0, 1)
    inter = (m > (thr + off)).sum((-1, -2))
    union = (m > (thr - off)).sum((-1, -2))
    stab = inter / union
    b = m > thr
    rows, cols = b.any(-1), b.any(-2)
    hh, ww = b.shape[-2:]
    ar_h, ar_w = torch.arange(hh), torch.arange(ww)
    top = torch.where(rows, ar_h, hh).min(-1).values
    bot = torch.where(rows, ar_h, -1).max(-1).values
    left = torch.where(cols, ar_w, ww).min(-1).values
    right = torch.where(cols, ar_w, -1).max(-1).values
    empty = (right < left) | (bot < top)
    boxes = torch.stack([left, top, right, bot], -1) * (~empty)[:, None]
    return b, stab, boxes
