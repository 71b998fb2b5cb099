"""The reference's free helper functions (utils.py) under their own names, evaluated on the device.

`Hybridgl_main.py` imports `Compute_IoU, gen_dir_mask, relation_boxes` from `utils` (Hybridgl_main.py:14); the fused
tail of this package (ops.coherence_scores / ops.score_sentence / ops.iou_select) does not need them, but code written
against the reference keeps working with `from hybridgl_amd.utils import ...`.  The spaCy-based extractors of utils.py
(`extract_noun_phrase`, `extract_dir_phrase`, ...) belong to the external parser and are inputs here.
"""
import torch

from . import _lib, ops
from ._lib import check


def gen_dir_mask(dirflag, height, width, device="cuda"):
    """utils.py:135-161 -> [height, width] fp32 on `device` (a GPU)."""
    lib = _lib.load()
    out = torch.empty((height, width), dtype=torch.float32, device=device if device else "cuda")
    check(lib.hgl_gen_dir_mask(ops.DIRFLAG.get(dirflag, 0), int(height), int(width), out.data_ptr(), ops._stream()),
          "hgl_gen_dir_mask")
    return out


def relation_boxes(boxi, boxj, scorei, scorej, relaword):
    """utils.py:240-268.  boxi / boxj: XYWH int64 tensors [4] (or [n,4] for n pairs), scores: 0-d (or [n]) fp32
    tensors, all on the GPU -> tensor of the same leading shape as the scores."""
    lib = _lib.load()
    bi = boxi.reshape(-1, 4).to(torch.int64).contiguous()
    bj = boxj.reshape(-1, 4).to(torch.int64).contiguous()
    si = torch.as_tensor(scorei, dtype=torch.float32, device=bi.device).reshape(-1).contiguous()
    sj = torch.as_tensor(scorej, dtype=torch.float32, device=bi.device).reshape(-1).contiguous()
    n = bi.shape[0]
    assert bj.shape[0] == n and si.numel() == n and sj.numel() == n
    out = torch.empty((n,), dtype=torch.float32, device=bi.device)
    check(lib.hgl_relation_boxes(ops._dev(bi, torch.int64, "boxi"), ops._dev(bj, torch.int64, "boxj"), si.data_ptr(),
                                 sj.data_ptr(), n, ops.RELAWORD.get(relaword, 0), out.data_ptr(), ops._stream()),
          "hgl_relation_boxes")
    return out.reshape(()) if boxi.dim() == 1 else out


def Compute_IoU(pred, target, cum_I, cum_U, mean_IoU=[]):   # noqa: B006 -- the reference's signature, shared default list included
    """utils.py:365-384: popcounts on the device; cum_I / cum_U grow in place semantics as in the reference (returned)."""
    iu = ops.iou_counts(pred, target.squeeze(0) if target.dim() == 3 else target)
    I, U = iu[0], iu[1]
    this_iou = 0.0 if int(U) == 0 else I * 1.0 / U      # `if U == 0` reads the count back, as the reference does
    cum_I += I
    cum_U += U
    mean_IoU.append(this_iou)
    return this_iou, mean_IoU, cum_I, cum_U
