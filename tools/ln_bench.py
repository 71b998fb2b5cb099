#!/usr/bin/env python3
"""The streaming passes between the GEMMs, timed alone (HIP events): LayerNorm at the row counts of a group of 8 refs
(CLIP 201728 x 768: writes the fp16 hi + lo pair; SAM 32768 x 1280), as GB/s of the bytes they move (read 4 B + write 4 B per
element)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
ops.set_precision("f16x3")


def timed(fn, nbytes, name, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    print(f"{name:40s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:7.1f} GB/s")


for M, D, name in [(201728, 768, "clip group of 8"), (25216, 768, "clip one ref"), (32768, 1280, "sam group of 8"), (4096, 1280, "sam one image")]:
    x = torch.randn(M, D, device=dev)
    w, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
    timed(lambda: ops.layernorm(x, w, b), M * D * 8.0, f"layernorm fp32 {M} x {D} ({name})")
