"""Weight statistics of TRAINED checkpoints imposed on the seeded random weights (test support; no checkpoints exist
offline).  Seeded Gaussian weights keep every activation O(1); trained CLIP / SAM transformers do not: LayerNorm gains
reach ~10, a handful of residual channels carry "massive activations" hundreds of times the median, and some MLP
units see pre-activations of several tens.  These transforms reproduce those three features so that the split-fp16
(f16x3) arithmetic is exercised where its hi / lo halves span many binades."""
import numpy as np


def _blocks(sd, prefix):
    ids = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix)})
    return [f"{prefix}{i}" for i in ids]


def stress_clip_state_dict(sd, seed=1, massive=300.0, gain_max=10.0, fc_scale=8.0):
    """CLIP (OpenAI key names): both towers."""
    rng = np.random.default_rng(seed)
    out = {k: np.array(v, copy=True) for k, v in sd.items()}
    for tower in ("visual.transformer.resblocks.", "transformer.resblocks."):
        blocks = _blocks(out, tower)
        for bi, b in enumerate(blocks):
            D = out[f"{b}.ln_1.weight"].shape[0]
            for ln in ("ln_1", "ln_2"):
                out[f"{b}.{ln}.weight"] = (out[f"{b}.{ln}.weight"] * rng.uniform(0.5, gain_max, D)).astype(np.float32)
                out[f"{b}.{ln}.bias"] = (out[f"{b}.{ln}.bias"] + rng.normal(0, 0.5, D)).astype(np.float32)
            hid = out[f"{b}.mlp.c_fc.weight"].shape[0]
            rows = rng.choice(hid, size=max(4, hid // 64), replace=False)
            out[f"{b}.mlp.c_fc.weight"][rows] *= np.float32(fc_scale)            # pre-activations of several tens
            if bi in (1, len(blocks) // 2):                                       # massive residual channels appear early / mid-depth
                ch = rng.choice(D, size=3, replace=False)
                out[f"{b}.mlp.c_proj.weight"][ch] *= np.float32(massive)
                out[f"{b}.attn.out_proj.weight"][ch[:1]] *= np.float32(massive / 3)
    return out


def stress_sam_state_dict(sd, cfg, seed=1, massive=300.0, gain_max=10.0, fc_scale=6.0):
    """SAM image encoder (segment_anything key names)."""
    rng = np.random.default_rng(seed)
    out = {k: np.array(v, copy=True) for k, v in sd.items()}
    blocks = _blocks(out, "image_encoder.blocks.")
    for bi, b in enumerate(blocks):
        D = out[f"{b}.norm1.weight"].shape[0]
        for ln in ("norm1", "norm2"):
            out[f"{b}.{ln}.weight"] = (out[f"{b}.{ln}.weight"] * rng.uniform(0.5, gain_max, D)).astype(np.float32)
            out[f"{b}.{ln}.bias"] = (out[f"{b}.{ln}.bias"] + rng.normal(0, 0.5, D)).astype(np.float32)
        hid = out[f"{b}.mlp.lin1.weight"].shape[0]
        rows = rng.choice(hid, size=max(4, hid // 64), replace=False)
        out[f"{b}.mlp.lin1.weight"][rows] *= np.float32(fc_scale)
        if bi == 0:
            ch = rng.choice(D, size=3, replace=False)
            out[f"{b}.mlp.lin2.weight"][ch] *= np.float32(massive)
            out[f"{b}.attn.proj.weight"][ch[:1]] *= np.float32(massive / 3)
    return out
