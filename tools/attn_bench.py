#!/usr/bin/env python3
"""Kernel timing of the attention shapes of the pipeline (groups of 8 refs): CLIP 197 x 64, SAM window 196 x 80 (rel-pos
tables in the kernel), SAM global 4096 x 80, GEM 785 x 64, text (causal).  HGL_ATTN_WIDE=0 (diagnostic library only) selects the 4-wave kernels."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
ops.set_precision("f16x3")


def timed(fn, flops, name, iters=10):
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
    print(f"{name:28s} {ms * 1e3:9.1f} us  {flops / ms / 1e9:7.1f} TF/s algorithmic")


def plain(B, H, S, hd, name, mask="none"):
    q, k, v = (torch.randn(B, S, H * hd, device=dev) for _ in range(3))
    timed(lambda: ops.attention(q, k, v, H, mask=mask), 4.0 * B * H * S * S * hd, name)


plain(1024, 12, 197, 64, "clip 1024x12 S197 hd64")
plain(128, 12, 197, 64, "clip 128x12 S197 hd64")


def cls_keep(B, H, S, hd, name):
    q, k, v = (torch.randn(B, S, H * hd, device=dev) for _ in range(3))
    keep = (torch.rand(B, S - 1, device=dev) < 0.3).to(torch.uint8)
    timed(lambda: ops.attention(q, k, v, H, mask="cls_keep", keep=keep, keep_b0=0, keep_n=B), 4.0 * B * H * S * S * hd, name)


cls_keep(1024, 12, 197, 64, "clip 1024x12 S197 CLS-keep")

plain(8, 12, 785, 64, "gem 8x12 S785 hd64")
plain(24, 12, 785, 64, "gem self-self 24x12 S785")      # q-q / k-k / v-v of eight images stacked along the batch
plain(8, 16, 4096, 80, "sam global 8x16 S4096 hd80")
plain(96, 8, 77, 64, "text 96x8 S77 hd64 causal", mask="causal")
# SAM windows through the fused-table entry point (sam_api path): time via the encoder instead
from hybridgl_amd import sam as hsam, weights
from hybridgl_amd.synth import synth_image
cfg = weights.SAM_CONFIGS["vit_h_d2"]
m = hsam.Sam(weights.sam_state_dict("vit_h_d2", 0), cfg, dev)
imgs = [torch.from_numpy(synth_image(1024, 1024, 20 + i)).to(dev) for i in range(8)]
lib.hgl_prof_enable(1)
m.encode_batch(imgs)
torch.cuda.synchronize()
lib.hgl_prof_enable(0)
n, ms, fl, by = C.c_longlong(), C.c_double(), C.c_double(), C.c_double()
lib.hgl_prof_read(1, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by))
print(f"sam vit_h_d2 x8 attention class: {n.value} launches {ms.value * 1e3:.1f} us total, {fl.value / ms.value / 1e9:.1f} TF/s (1 windowed 200 windows + 1 global)")
# layout probe: the same 12288 items with every (batch, head) slice CONTIGUOUS (rows of 256 B instead of 256-B pieces of
# 9 KB rows): what the access pattern costs
plain(1024 * 12, 1, 197, 64, "12288x1 S197 fp32, head-major")
