"""Checkpoint FILES in the formats the reference's loaders read (clip/clip.py:119-142, build_sam.py:103-106), rebuilt
from the seeded weights of hybridgl_amd/weights.py.  tests/golden/ckpt.npz (oracle/gen_golden.py: gen_ckpt) holds the
digests of the reference's own state_dicts -- `CLIP.state_dict()` after the reference's convert_weights, `Sam.state_dict()`
-- and the list of tensors convert_weights stores as fp16; a file written here is accepted only when its digest equals the
reference's, i.e. it is byte for byte what `torch.save(reference_module.state_dict())` holds (the files are 30 MB: not stored)."""
import hashlib
import os

import numpy as np
import torch


def state_dict_digest(sd):
    h = hashlib.sha256()
    for k in sorted(sd):
        v = sd[k].detach().cpu().contiguous()
        h.update(k.encode())
        h.update(str(v.dtype).encode())
        h.update(str(tuple(v.shape)).encode())
        h.update(v.numpy().tobytes())
    return h.hexdigest()


def clip_openai_state_dict(golden_dir, name="tiny", seed=0):
    """state_dict of an OpenAI-style archive: fp16 where the reference's convert_weights makes it fp16, the three scalar
    entries build_model deletes (clip/model.py:505-507) included"""
    from hybridgl_amd import weights
    g = np.load(os.path.join(golden_dir, "ckpt.npz"))
    half = set(str(k) for k in g["clip_fp16_keys"])
    cfg = weights.CLIP_CONFIGS[name]
    sd = {k: (torch.from_numpy(v.copy()).half() if k in half else torch.from_numpy(v.copy()))
          for k, v in weights.clip_state_dict(name, seed).items()}
    sd["input_resolution"] = torch.tensor(cfg["image_resolution"])
    sd["context_length"] = torch.tensor(cfg["context_length"])
    sd["vocab_size"] = torch.tensor(cfg["vocab_size"])
    assert state_dict_digest(sd) == str(g["clip_digest"]), "rebuilt CLIP state_dict differs from the reference's"
    return sd


def sam_state_dict_tensors(golden_dir, name="tiny", seed=0):
    from hybridgl_amd import weights
    g = np.load(os.path.join(golden_dir, "ckpt.npz"))
    sd = {k: torch.from_numpy(v.copy()) for k, v in weights.sam_state_dict(name, seed).items()}
    assert state_dict_digest(sd) == str(g["sam_digest"]), "rebuilt SAM state_dict differs from the reference's"
    return sd


class _Holder(torch.nn.Module):
    """a module tree that carries a state_dict's tensors under their dotted names (what survives in a TorchScript
    archive as far as `torch.jit.load(path).state_dict()` is concerned)"""

    def __init__(self, entries):
        super().__init__()
        kids = {}
        for k, v in entries.items():
            if "." in k:
                head, rest = k.split(".", 1)
                kids.setdefault(head, {})[rest] = v
            else:
                self.register_buffer(k, v.clone())
        for head, sub in kids.items():
            self.add_module(head, _Holder(sub))

    def forward(self, x: torch.Tensor):
        return x


def write_clip_files(golden_dir, out_dir):
    """-> (plain state_dict file, {'state_dict': ...} wrapper file, TorchScript archive)"""
    sd = clip_openai_state_dict(golden_dir)
    p1, p2, p3 = (os.path.join(out_dir, n) for n in ("clip_tiny_fp16.pt", "clip_tiny_wrapped.pt", "clip_tiny_jit.pt"))
    torch.save(sd, p1)
    torch.save({"state_dict": sd}, p2)
    torch.jit.save(torch.jit.script(_Holder(sd)), p3)
    return p1, p2, p3


def write_sam_file(golden_dir, out_dir):
    p = os.path.join(out_dir, "sam_tiny.pth")
    torch.save(sam_state_dict_tensors(golden_dir), p)
    return p
