"""The checkpoint loaders (clip/clip.py:119-142 + build_model clip/model.py:474-511; build_sam.py:103-106) on files in the
reference's formats -- host side: the rebuilt files ARE the reference's (digests), every format loads to the same fp32
values, and the geometry inferred from the key set is the reference's.  The GPU side is tests/test_gpu_checkpoints.py."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ckpt_files as CK   # noqa: E402


def test_rebuilt_files_are_the_references(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "ckpt.npz"))
    sd = CK.clip_openai_state_dict(golden_dir)           # asserts the digest of the reference's converted state_dict
    assert sorted(sd) == [str(k) for k in g["clip_keys"]]
    n16 = sum(v.dtype == torch.float16 for v in sd.values())
    assert n16 == len(g["clip_fp16_keys"]) and 0 < n16 < len(sd)      # LayerNorms, embeddings, logit_scale stay fp32
    ssd = CK.sam_state_dict_tensors(golden_dir)
    assert sorted(ssd) == [str(k) for k in g["sam_keys"]]


def test_clip_loader_reads_every_archive_format(golden_dir, tmp_path):
    """plain state_dict, {'state_dict': ...} wrapper and TorchScript archive -> the same fp32 numpy dict, fp16 values
    up-cast exactly, the three scalar entries dropped (clip/model.py:505-507)"""
    from hybridgl_amd import weights
    from hybridgl_amd.backbone import _infer_config, load_clip_state_dict
    files = CK.write_clip_files(golden_dir, str(tmp_path))
    want = CK.clip_openai_state_dict(golden_dir)
    loaded = [load_clip_state_dict(p) for p in files]
    for sd in loaded:
        assert not ({"input_resolution", "context_length", "vocab_size"} & set(sd))
        assert sorted(sd) == sorted(k for k in want if k not in ("input_resolution", "context_length", "vocab_size"))
        for k, v in sd.items():
            assert v.dtype == np.float32
            assert np.array_equal(v, want[k].float().numpy()), k
        # build_model's geometry inference on the real key set (clip/model.py:474-503)
        cfg = _infer_config(sd)
        ref = weights.CLIP_CONFIGS["tiny"]
        assert {k: cfg[k] for k in ref} == ref
    # fp16 storage really rounds: the loaded weights differ from the seeded fp32 ones, by at most half an fp16 ulp
    seeded = weights.clip_state_dict("tiny", 0)
    k = "visual.transformer.resblocks.0.attn.in_proj_weight"
    d = np.abs(loaded[0][k] - seeded[k])
    assert 0 < d.max() <= np.abs(seeded[k]).max() * 2.0 ** -11


def test_infer_config_on_the_full_size_key_sets():
    """ViT-B/16 and ViT-L/14 key sets (shapes only) -> the published geometries (clip/model.py:474-503)"""
    from hybridgl_amd import weights
    from hybridgl_amd.backbone import _infer_config

    class Shape:
        def __init__(self, shape):
            self.shape = tuple(shape)
    for name in ("ViT-B/16", "ViT-L/14"):
        cfg = weights.CLIP_CONFIGS[name]
        shapes = {k: Shape(v) for k, v in weights.clip_shapes(name).items()}
        got = _infer_config(shapes)
        assert {k: got[k] for k in cfg} == cfg, name


def test_sam_loader_reads_a_released_style_file(golden_dir, tmp_path):
    from hybridgl_amd import weights
    from hybridgl_amd.sam import load_sam_state_dict
    p = CK.write_sam_file(golden_dir, str(tmp_path))
    sd = load_sam_state_dict(p)
    seeded = weights.sam_state_dict("tiny", 0)
    assert sorted(sd) == sorted(seeded)
    assert all(v.dtype == np.float32 and np.array_equal(v, seeded[k]) for k, v in sd.items())
