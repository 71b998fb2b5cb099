#!/usr/bin/env python3
"""Kernel timing of the pre-split attention kernels (csrc/attention_ps.hip) against the fp32-input kernels on the pipeline's
shapes: CLIP 197 x 64 (plain / CLS keep), GEM 785 x 64.  The pre-split time INCLUDES the split pass of this entry point
(hgl_attention_presplit_f32 splits its fp32 input first; in the pipeline the GEMM write-out does that), reported apart."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
ops.set_precision("f16x3")


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for (B, H, S, hd, mask) in [(1024, 12, 197, 64, "none"), (1024, 12, 197, 64, "cls_keep"), (24, 12, 785, 64, "none")]:
    qkv = torch.randn(B, S, 3 * H * hd, device=dev)
    D = H * hd
    q, k, v = (qkv[..., i * D:(i + 1) * D].contiguous() for i in range(3))
    kw = {}
    if mask == "cls_keep":
        kw = dict(keep=(torch.rand(B, S - 1, device=dev) < 0.3).to(torch.uint8), keep_b0=0, keep_n=B)
    fl = 4.0 * B * H * S * S * hd
    t0 = timed(lambda: ops.attention(q, k, v, H, mask=mask, **kw))
    t1 = timed(lambda: ops.attention_presplit(qkv, H, mask=mask, **kw))
    print(f"B {B} S {S} hd {hd} {mask:8s}: fp32-input {t0:8.1f} us ({fl / t0 / 1e6:6.1f} TF/s) | pre-split incl. split pass {t1:8.1f} us")
