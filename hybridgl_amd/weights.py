"""Seeded synthetic weights with the reference's state_dict key names.

No checkpoints exist offline, so parity fixtures and the benchmark use weights drawn here.
Each tensor is drawn from its own PCG64 stream keyed by (seed, crc32(name)), so any subset
of tensors is reproducible anywhere (container, GPU box) without shipping the values.

Key names and shapes follow clip/model.py:272-287,340-365 (OpenAI CLIP state_dict) and
segment_anything/build_sam.py:55-101 (SAM state_dict); stds follow
CLIP.initialize_parameters (clip/model.py:367-394).  LayerNorm weights/biases and linear
biases are perturbed (not 1/0) so that every parameter influences the output.
"""
import zlib
from collections import OrderedDict

import numpy as np


def _draw(seed, name, shape, std, mean=0.0):
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
    a = rng.standard_normal(size=shape, dtype=np.float32)
    if std != 1.0:
        a *= np.float32(std)
    if mean != 0.0:
        a += np.float32(mean)
    return a


def _resblocks(sd, prefix, width, layers, seed):
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    attn_std = width ** -0.5
    fc_std = (2 * width) ** -0.5
    for i in range(layers):
        p = f"{prefix}.resblocks.{i}"
        sd[f"{p}.ln_1.weight"] = _draw(seed, f"{p}.ln_1.weight", (width,), 0.1, 1.0)
        sd[f"{p}.ln_1.bias"] = _draw(seed, f"{p}.ln_1.bias", (width,), 0.02)
        sd[f"{p}.attn.in_proj_weight"] = _draw(seed, f"{p}.attn.in_proj_weight", (3 * width, width), attn_std)
        sd[f"{p}.attn.in_proj_bias"] = _draw(seed, f"{p}.attn.in_proj_bias", (3 * width,), 0.02)
        sd[f"{p}.attn.out_proj.weight"] = _draw(seed, f"{p}.attn.out_proj.weight", (width, width), proj_std)
        sd[f"{p}.attn.out_proj.bias"] = _draw(seed, f"{p}.attn.out_proj.bias", (width,), 0.02)
        sd[f"{p}.ln_2.weight"] = _draw(seed, f"{p}.ln_2.weight", (width,), 0.1, 1.0)
        sd[f"{p}.ln_2.bias"] = _draw(seed, f"{p}.ln_2.bias", (width,), 0.02)
        sd[f"{p}.mlp.c_fc.weight"] = _draw(seed, f"{p}.mlp.c_fc.weight", (4 * width, width), fc_std)
        sd[f"{p}.mlp.c_fc.bias"] = _draw(seed, f"{p}.mlp.c_fc.bias", (4 * width,), 0.02)
        sd[f"{p}.mlp.c_proj.weight"] = _draw(seed, f"{p}.mlp.c_proj.weight", (width, 4 * width), proj_std)
        sd[f"{p}.mlp.c_proj.bias"] = _draw(seed, f"{p}.mlp.c_proj.bias", (width,), 0.02)


# geometry presets: (embed_dim, image_resolution, vision_layers, vision_width, vision_patch,
#                    context_length, vocab_size, text_width, text_heads, text_layers)
CLIP_CONFIGS = {
    "ViT-B/16": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768,
                     vision_patch_size=16, context_length=77, vocab_size=49408,
                     transformer_width=512, transformer_heads=8, transformer_layers=12),
    "ViT-B/32": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768,
                     vision_patch_size=32, context_length=77, vocab_size=49408,
                     transformer_width=512, transformer_heads=8, transformer_layers=12),
    # extension named by BASELINE.json (no oracle in the reference: model/backbone.py:16-21)
    "ViT-L/14": dict(embed_dim=768, image_resolution=224, vision_layers=24, vision_width=1024,
                     vision_patch_size=14, context_length=77, vocab_size=49408,
                     transformer_width=768, transformer_heads=12, transformer_layers=12),
    # tiny geometry for fast parity tests (heads = width/64 as in clip/model.py:333,486)
    "tiny": dict(embed_dim=32, image_resolution=64, vision_layers=12, vision_width=128,
                 vision_patch_size=16, context_length=16, vocab_size=512,
                 transformer_width=64, transformer_heads=1, transformer_layers=3),
}


def clip_state_dict(name="ViT-B/16", seed=0):
    """Seeded CLIP state_dict (numpy fp32) with OpenAI key names."""
    cfg = CLIP_CONFIGS[name]
    w, L, p = cfg["vision_width"], cfg["vision_layers"], cfg["vision_patch_size"]
    g = cfg["image_resolution"] // p
    E = cfg["embed_dim"]
    sd = OrderedDict()
    scale = w ** -0.5
    sd["visual.conv1.weight"] = _draw(seed, "visual.conv1.weight", (w, 3, p, p), (3 * p * p) ** -0.5)
    sd["visual.class_embedding"] = _draw(seed, "visual.class_embedding", (w,), scale)
    sd["visual.positional_embedding"] = _draw(seed, "visual.positional_embedding", (g * g + 1, w), scale)
    sd["visual.ln_pre.weight"] = _draw(seed, "visual.ln_pre.weight", (w,), 0.1, 1.0)
    sd["visual.ln_pre.bias"] = _draw(seed, "visual.ln_pre.bias", (w,), 0.02)
    _resblocks(sd, "visual.transformer", w, L, seed)
    sd["visual.ln_post.weight"] = _draw(seed, "visual.ln_post.weight", (w,), 0.1, 1.0)
    sd["visual.ln_post.bias"] = _draw(seed, "visual.ln_post.bias", (w,), 0.02)
    sd["visual.proj"] = _draw(seed, "visual.proj", (w, E), scale)
    tw, tl = cfg["transformer_width"], cfg["transformer_layers"]
    _resblocks(sd, "transformer", tw, tl, seed)
    sd["token_embedding.weight"] = _draw(seed, "token_embedding.weight", (cfg["vocab_size"], tw), 0.02)
    sd["positional_embedding"] = _draw(seed, "positional_embedding", (cfg["context_length"], tw), 0.01)
    sd["ln_final.weight"] = _draw(seed, "ln_final.weight", (tw,), 0.1, 1.0)
    sd["ln_final.bias"] = _draw(seed, "ln_final.bias", (tw,), 0.02)
    sd["text_projection"] = _draw(seed, "text_projection", (tw, E), tw ** -0.5)
    sd["logit_scale"] = np.array(np.log(1 / 0.07), dtype=np.float32)
    return sd


def clip_vision_heads(cfg):
    return cfg["vision_width"] // 64  # clip/model.py:333
