"""ORACLE (test infrastructure, NOT product code): numpy fp32 restatement of the GEM heat-map stage that
Hybridgl_main.py:36-39,200-201 calls through the external package `gem_torch==1.0.1` over
`open_clip_torch==2.24.0` (environment.yaml:206,227).

PARITY UNPINNED for the GEM arithmetic: neither package is vendored in the reference tree nor installed in this
image, and no reference test holds a vector of its output.  What is restated here is the PUBLISHED algorithm
(Bousselham et al., "Grounding Everything: Emerging Localization Properties in Vision-Language Transformers",
CVPR 2024, and the gem_torch 1.0.1 sources as published on PyPI): self-self attention over the q, k and v
projections of the last `gem_depth - 1` blocks with the adaptive temperature mean(||x||) / sqrt(head_dim), a
second residual stream without MLPs, ln_post + proj on every token, cosine matching against the text embedding of
"a photo of a {phrase}.", bilinear up-sampling to the input size and min-max normalisation.  The anchors are the
reference's call sites: `gem.create_gem_model(model_name='ViT-B/16', pretrained='openai')`,
`gem.get_gem_img_transform()` (448 x 448) and `gem_model(tensor_img, [noun_phrase])[0]` followed by
`T.Resize((h, w), antialias=True)` (Hybridgl_main.py:200-201).

PINNED here (torch itself is the arithmetic the reference calls, and it is installed): the three resampling
operators -- F.interpolate bilinear (gem), bilinear antialias (T.Resize on a tensor) and bicubic with a scale
factor (position-embedding interpolation) -- are checked against torch in tests/test_gem_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import math

import numpy as np

from . import clip_oracle as O

F32 = np.float32


# ----------------------------------------------------------------------------- resampling (pinned against torch)
def _cubic_coeffs(t, A=-0.75):
    """ATen get_cubic_upsample_coefficients (UpSampleKernel / UpSample.h), A = -0.75."""
    t = t.astype(F32)
    A = F32(A)
    c1 = lambda x: ((A + F32(2)) * x - (A + F32(3))) * x * x + F32(1)
    c2 = lambda x: ((A * x - F32(5) * A) * x + F32(8) * A) * x - F32(4) * A
    return np.stack([c2(t + F32(1)), c1(t), c1(F32(1) - t), c2(F32(2) - t)], axis=-1).astype(F32)


def _bicubic_axis(x, axis, out_size, scale_factor):
    """one axis of F.interpolate(mode='bicubic', align_corners=False, scale_factor=s): the kernel maps with 1/s
    (recompute_scale_factor unset), border taps are clamped."""
    n = x.shape[axis]
    scale = F32(1.0 / scale_factor) if scale_factor else F32(n) / F32(out_size)
    src = (scale * (np.arange(out_size, dtype=F32) + F32(0.5)) - F32(0.5)).astype(F32)
    i = np.floor(src).astype(np.int64)
    co = _cubic_coeffs(src - i.astype(F32))
    out = 0
    for k in range(4):
        idx = np.clip(i - 1 + k, 0, n - 1)
        shape = [1] * x.ndim
        shape[axis] = out_size
        out = out + np.take(x, idx, axis=axis) * co[:, k].reshape(shape)
    return out.astype(F32)


def bicubic_scale(x, out_h, out_w, sf_h, sf_w):
    """F.interpolate(x[..., h, w], scale_factor=(sf_h, sf_w), mode='bicubic'): separable, width first then
    height as ATen accumulates (sum over rows of row-coefficient x (sum over columns))."""
    y = _bicubic_axis(x.astype(F32), x.ndim - 1, out_w, sf_w)
    return _bicubic_axis(y, x.ndim - 2, out_h, sf_h)


def interpolate_pos_encoding(pos, grid_h, grid_w):
    """GEM's (DINO-style) position-embedding interpolation: the patch part of positional_embedding [1+n*n, D] is
    resampled bicubically to grid_h x grid_w with scale factors (grid + 0.1) / n."""
    N = pos.shape[0] - 1
    n = int(round(math.sqrt(N)))
    if n == grid_h and n == grid_w:
        return pos.astype(F32)
    D = pos.shape[1]
    patch = pos[1:].reshape(n, n, D).transpose(2, 0, 1)                    # [D, n, n]
    sf_h, sf_w = (grid_h + 0.1) / n, (grid_w + 0.1) / n
    assert int(n * sf_h) == grid_h and int(n * sf_w) == grid_w
    up = bicubic_scale(patch, grid_h, grid_w, sf_h, sf_w)                  # [D, gh, gw]
    return np.concatenate([pos[:1], up.transpose(1, 2, 0).reshape(grid_h * grid_w, D)], axis=0).astype(F32)


def _aa_weights(in_size, out_size):
    """ATen _compute_indices_min_size_weights_aa for the triangle (bilinear) filter, align_corners=False.
    Returns per output index (xmin, weights)."""
    scale = F32(in_size) / F32(out_size)
    support = F32(scale) if scale >= 1 else F32(1.0)
    invscale = F32(1.0) / scale if scale >= 1 else F32(1.0)
    res = []
    for i in range(out_size):
        center = scale * F32(i + 0.5)
        xmin = max(int(center - support + F32(0.5)), 0)
        xsize = min(int(center + support + F32(0.5)), in_size) - xmin
        j = np.arange(xsize, dtype=F32)
        w = np.maximum(F32(0), F32(1) - np.abs((j + F32(xmin) - center + F32(0.5)) * invscale)).astype(F32)
        tot = w.sum(dtype=F32)
        if tot != 0:
            w = (w / tot).astype(F32)
        res.append((xmin, w))
    return res


def resize_bilinear_aa(x, oh, ow):
    """T.Resize((oh, ow), antialias=True) on a float tensor [..., h, w] == F.interpolate(bilinear,
    antialias=True, align_corners=False) (Hybridgl_main.py:201)."""
    x = x.astype(F32)
    h, w = x.shape[-2:]
    wx = _aa_weights(w, ow)
    wy = _aa_weights(h, oh)
    tmp = np.empty(x.shape[:-1] + (ow,), dtype=F32)
    for X, (x0, ww) in enumerate(wx):
        tmp[..., X] = (x[..., x0:x0 + len(ww)] * ww).sum(axis=-1, dtype=F32)
    out = np.empty(x.shape[:-2] + (oh, ow), dtype=F32)
    for Y, (y0, ww) in enumerate(wy):
        out[..., Y, :] = (tmp[..., y0:y0 + len(ww), :] * ww[:, None]).sum(axis=-2, dtype=F32)
    return out


# ----------------------------------------------------------------------------- GEM ViT (unpinned, see header)
def _normalize(x, eps=1e-12):
    """F.normalize(x, dim=-1)."""
    n = np.sqrt((x * x).sum(axis=-1, keepdims=True, dtype=F32))
    return (x / np.maximum(n, F32(eps))).astype(F32)


def _heads(x, heads):
    B, S, D = x.shape
    return x.reshape(B, S, heads, D // heads).transpose(0, 2, 1, 3)


def _merge(x):
    B, h, S, hd = x.shape
    return x.transpose(0, 2, 1, 3).reshape(B, S, h * hd)


def self_self_attention(x, sd, prefix, heads, ss_attn_iter=1, ss_attn_temp=None):
    """gem SelfSelfAttention.forward on x = ln_1(tokens) [B, S, D]: returns (x_gem, x_ori), both after the
    block's out-projection."""
    g = lambda k: sd[f"{prefix}.{k}"]
    B, S, D = x.shape
    hd = D // heads
    scale = F32(hd ** -0.5)
    qkv = O.linear(x, g("attn.in_proj_weight"), g("attn.in_proj_bias"))
    q, k, v = (_heads(qkv[..., i * D:(i + 1) * D], heads) for i in range(3))
    att = O.softmax(((q @ k.transpose(0, 1, 3, 2)) * scale).astype(F32))
    x_ori = O.linear(_merge(att @ v), g("attn.out_proj.weight"), g("attn.out_proj.bias"))
    if ss_attn_temp is None:
        pre_norm = np.sqrt((x * x).sum(axis=-1, dtype=F32)).mean(axis=-1, dtype=F32)      # [B]
        inv_temp = (pre_norm * scale).reshape(B, 1, 1, 1).astype(F32)
    else:
        inv_temp = F32(ss_attn_temp)
    outs = []
    for xs in (v, k, q):
        for _ in range(ss_attn_iter):
            xs = _normalize(xs)
            a = O.softmax(((xs @ xs.transpose(0, 1, 3, 2)) * inv_temp).astype(F32))
            xs = (a @ xs).astype(F32)
        xs = _normalize(xs)
        a = O.softmax(((xs @ xs.transpose(0, 1, 3, 2)) * inv_temp).astype(F32))
        outs.append((a @ v).astype(F32))
    xs = ((outs[0] + outs[1] + outs[2]) / F32(3)).astype(F32)
    x_gem = O.linear(_merge(xs), g("attn.out_proj.weight"), g("attn.out_proj.bias"))
    return x_gem, x_ori


def gem_vit_forward(sd, imgs, gem_depth=7, ss_attn_iter=1, ss_attn_temp=None, heads=None):
    """GEMViT.forward: imgs [B, 3, R, R] (R a multiple of the patch size) -> (feat_gem, feat_ori), each
    [B, 1 + g*g, embed]: ln_post + proj applied to every token of the two residual streams.  The last
    gem_depth - 1 blocks are GEM blocks (gem_wrapper: `for i in range(1, depth)` swaps resblocks[-i])."""
    w = sd["visual.conv1.weight"]
    D, _, p, _ = w.shape
    if heads is None:
        heads = D // 64
    layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    g = imgs.shape[-1] // p
    sd2 = dict(sd)
    sd2["visual.positional_embedding"] = interpolate_pos_encoding(sd["visual.positional_embedding"], g, g)
    x = O.vit_embed(sd2, imgs)
    n_gem = max(0, min(layers, gem_depth - 1))
    x_gem = None
    for i in range(layers):
        prefix = f"visual.transformer.resblocks.{i}"
        if i < layers - n_gem:
            x = O.resblock(x, sd, prefix, heads)
            continue
        gg = lambda k: sd[f"{prefix}.{k}"]
        if x_gem is None:
            x_gem = x
        r_gem, r_ori = self_self_attention(O.layer_norm(x, gg("ln_1.weight"), gg("ln_1.bias")), sd, prefix, heads,
                                           ss_attn_iter, ss_attn_temp)
        x = x + r_ori
        h = O.layer_norm(x, gg("ln_2.weight"), gg("ln_2.bias"))
        h = O.quick_gelu(O.linear(h, gg("mlp.c_fc.weight"), gg("mlp.c_fc.bias")))
        x = x + O.linear(h, gg("mlp.c_proj.weight"), gg("mlp.c_proj.bias"))
        x_gem = x_gem + r_gem
    if x_gem is None:
        x_gem = x
    head = lambda t: (O.layer_norm(t, sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]) @ sd["visual.proj"]).astype(F32)
    return head(x_gem), head(x)


def gem_heatmap(feat, text, res, normalize=True):
    """GEMWrapper.forward after the encoders: feat [S, E] (row 0 = CLS), text [T, E] -> [T, res, res]:
    100 * cos(patch, text), tokens laid out row-major on the g x g grid, F.interpolate(bilinear) to (res, res),
    min-max per prompt."""
    f = _normalize(feat[1:].astype(F32))
    t = _normalize(text.astype(F32))
    g = int(round(math.sqrt(f.shape[0])))
    m = (F32(100.0) * (f @ t.T)).astype(F32).T.reshape(-1, g, g)
    up = O.bilinear_resize(m, res, res)
    if normalize:
        mn = up.reshape(up.shape[0], -1).min(axis=1)[:, None, None]
        mx = up.reshape(up.shape[0], -1).max(axis=1)[:, None, None]
        up = ((up - mn) / (mx - mn)).astype(F32)
    return up


def gem_prompts(phrases):
    """GEMWrapper.encode_text prompt template."""
    return [f"a photo of a {p}." for p in phrases]
