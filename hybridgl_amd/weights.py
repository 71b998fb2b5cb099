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


def clip_shapes(name="ViT-B/16"):
    """{key: shape} of the CLIP state_dict of a geometry, without drawing a single value (the key set and shapes are what
    build_model's geometry inference reads, clip/model.py:474-503)"""
    global _draw
    keep = _draw
    _draw = lambda seed, nm, shape, std, mean=0.0: np.broadcast_to(np.float32(0), shape)
    try:
        return {k: tuple(v.shape) for k, v in clip_state_dict(name, 0).items()}
    finally:
        _draw = keep


def clip_vision_heads(cfg):
    return cfg["vision_width"] // 64  # clip/model.py:333


# ----------------------------------------------------------------------------- SAM
SAM_CONFIGS = {
    # segment_anything/build_sam.py:14-21 (vit_h) + :55-101
    "vit_h": dict(embed_dim=1280, depth=32, num_heads=16, global_attn_indexes=(7, 15, 23, 31),
                  img_size=1024, patch_size=16, window_size=14, out_chans=256),
    # full ViT-H width/grid but two blocks (one windowed, one global): full-size parity test
    "vit_h_d2": dict(embed_dim=1280, depth=2, num_heads=16, global_attn_indexes=(1,),
                     img_size=1024, patch_size=16, window_size=14, out_chans=256),
    # segment_anything/build_sam.py:24-44: the smaller published encoders (head dim 64)
    "vit_l": dict(embed_dim=1024, depth=24, num_heads=16, global_attn_indexes=(5, 11, 17, 23),
                  img_size=1024, patch_size=16, window_size=14, out_chans=256),
    "vit_b": dict(embed_dim=768, depth=12, num_heads=12, global_attn_indexes=(2, 5, 8, 11),
                  img_size=1024, patch_size=16, window_size=14, out_chans=256),
    "vit_b_d2": dict(embed_dim=768, depth=2, num_heads=12, global_attn_indexes=(1,),
                     img_size=1024, patch_size=16, window_size=14, out_chans=256),
    # tiny geometry for parity tests: 16x16 tokens, 2 heads of 80, windowed + global blocks
    "tiny": dict(embed_dim=160, depth=4, num_heads=2, global_attn_indexes=(1, 3),
                 img_size=256, patch_size=16, window_size=14, out_chans=256),
    # 24x24 tokens (576: not a multiple of the 256-key chunks of the decoder's token -> image attention, nor of a whole
    # number of 64-token tiles per grid row): the ragged paths of the decoder kernels
    "tiny24": dict(embed_dim=160, depth=2, num_heads=2, global_attn_indexes=(1,),
                   img_size=384, patch_size=16, window_size=14, out_chans=256),
}


def sam_state_dict(name="vit_h", seed=0):
    """Seeded SAM state_dict (numpy fp32) with the reference's key names
    (segment_anything/modeling/*.py).  Linear/conv weights use std = fan_in^-0.5 so that
    activations and mask logits stay O(1) with random weights."""
    cfg = SAM_CONFIGS[name]
    D, L, H = cfg["embed_dim"], cfg["depth"], cfg["num_heads"]
    ps, g = cfg["patch_size"], cfg["img_size"] // cfg["patch_size"]
    hd, C = D // H, cfg["out_chans"]
    sd = OrderedDict()

    def lin(key, out_f, in_f, bias=True):
        sd[f"{key}.weight"] = _draw(seed, f"{key}.weight", (out_f, in_f), in_f ** -0.5)
        if bias:
            sd[f"{key}.bias"] = _draw(seed, f"{key}.bias", (out_f,), 0.02)

    def norm(key, n):
        sd[f"{key}.weight"] = _draw(seed, f"{key}.weight", (n,), 0.1, 1.0)
        sd[f"{key}.bias"] = _draw(seed, f"{key}.bias", (n,), 0.02)

    e = "image_encoder"
    sd[f"{e}.pos_embed"] = _draw(seed, f"{e}.pos_embed", (1, g, g, D), 0.02)
    sd[f"{e}.patch_embed.proj.weight"] = _draw(seed, f"{e}.patch_embed.proj.weight", (D, 3, ps, ps), (3 * ps * ps) ** -0.5)
    sd[f"{e}.patch_embed.proj.bias"] = _draw(seed, f"{e}.patch_embed.proj.bias", (D,), 0.02)
    for i in range(L):
        b = f"{e}.blocks.{i}"
        s = g if i in cfg["global_attn_indexes"] else cfg["window_size"]
        norm(f"{b}.norm1", D)
        lin(f"{b}.attn.qkv", 3 * D, D)
        lin(f"{b}.attn.proj", D, D)
        sd[f"{b}.attn.rel_pos_h"] = _draw(seed, f"{b}.attn.rel_pos_h", (2 * s - 1, hd), 0.1)
        sd[f"{b}.attn.rel_pos_w"] = _draw(seed, f"{b}.attn.rel_pos_w", (2 * s - 1, hd), 0.1)
        norm(f"{b}.norm2", D)
        lin(f"{b}.mlp.lin1", 4 * D, D)
        lin(f"{b}.mlp.lin2", D, 4 * D)
    sd[f"{e}.neck.0.weight"] = _draw(seed, f"{e}.neck.0.weight", (C, D, 1, 1), D ** -0.5)
    norm(f"{e}.neck.1", C)
    sd[f"{e}.neck.2.weight"] = _draw(seed, f"{e}.neck.2.weight", (C, C, 3, 3), (9 * C) ** -0.5)
    norm(f"{e}.neck.3", C)

    p = "prompt_encoder"
    sd[f"{p}.pe_layer.positional_encoding_gaussian_matrix"] = _draw(seed, f"{p}.pe_gauss", (2, C // 2), 1.0)
    for i in range(4):
        sd[f"{p}.point_embeddings.{i}.weight"] = _draw(seed, f"{p}.point_embeddings.{i}.weight", (1, C), 1.0)
    sd[f"{p}.not_a_point_embed.weight"] = _draw(seed, f"{p}.not_a_point_embed.weight", (1, C), 1.0)
    sd[f"{p}.no_mask_embed.weight"] = _draw(seed, f"{p}.no_mask_embed.weight", (1, C), 1.0)
    # mask_downscaling (dense mask prompts) is never used by the automatic generator but is part
    # of the reference state_dict (modeling/prompt_encoder.py:50-58, mask_in_chans=16)
    sd[f"{p}.mask_downscaling.0.weight"] = _draw(seed, f"{p}.md0.w", (4, 1, 2, 2), 0.5)
    sd[f"{p}.mask_downscaling.0.bias"] = _draw(seed, f"{p}.md0.b", (4,), 0.02)
    norm(f"{p}.mask_downscaling.1", 4)
    sd[f"{p}.mask_downscaling.3.weight"] = _draw(seed, f"{p}.md3.w", (16, 4, 2, 2), 0.25)
    sd[f"{p}.mask_downscaling.3.bias"] = _draw(seed, f"{p}.md3.b", (16,), 0.02)
    norm(f"{p}.mask_downscaling.4", 16)
    sd[f"{p}.mask_downscaling.6.weight"] = _draw(seed, f"{p}.md6.w", (C, 16, 1, 1), 0.25)
    sd[f"{p}.mask_downscaling.6.bias"] = _draw(seed, f"{p}.md6.b", (C,), 0.02)

    m = "mask_decoder"
    sd[f"{m}.iou_token.weight"] = _draw(seed, f"{m}.iou_token.weight", (1, C), 1.0)
    sd[f"{m}.mask_tokens.weight"] = _draw(seed, f"{m}.mask_tokens.weight", (4, C), 1.0)

    def attn(key, internal):
        lin(f"{key}.q_proj", internal, C)
        lin(f"{key}.k_proj", internal, C)
        lin(f"{key}.v_proj", internal, C)
        lin(f"{key}.out_proj", C, internal)

    for i in range(2):
        l = f"{m}.transformer.layers.{i}"
        attn(f"{l}.self_attn", C)
        norm(f"{l}.norm1", C)
        attn(f"{l}.cross_attn_token_to_image", C // 2)
        norm(f"{l}.norm2", C)
        lin(f"{l}.mlp.lin1", 2048, C)
        lin(f"{l}.mlp.lin2", C, 2048)
        norm(f"{l}.norm3", C)
        norm(f"{l}.norm4", C)
        attn(f"{l}.cross_attn_image_to_token", C // 2)
    attn(f"{m}.transformer.final_attn_token_to_image", C // 2)
    norm(f"{m}.transformer.norm_final_attn", C)
    sd[f"{m}.output_upscaling.0.weight"] = _draw(seed, f"{m}.up0.w", (C, C // 4, 2, 2), C ** -0.5)
    sd[f"{m}.output_upscaling.0.bias"] = _draw(seed, f"{m}.up0.b", (C // 4,), 0.02)
    norm(f"{m}.output_upscaling.1", C // 4)
    sd[f"{m}.output_upscaling.3.weight"] = _draw(seed, f"{m}.up3.w", (C // 4, C // 8, 2, 2), (C // 4) ** -0.5)
    sd[f"{m}.output_upscaling.3.bias"] = _draw(seed, f"{m}.up3.b", (C // 8,), 0.02)
    for i in range(4):
        h = f"{m}.output_hypernetworks_mlps.{i}"
        lin(f"{h}.layers.0", C, C)
        lin(f"{h}.layers.1", C, C)
        lin(f"{h}.layers.2", C // 8, C)
    h = f"{m}.iou_prediction_head"
    lin(f"{h}.layers.0", 256, C)
    lin(f"{h}.layers.1", 256, 256)
    lin(f"{h}.layers.2", 4, 256)
    return sd
