#!/usr/bin/env python3
"""Kernel-only timing of the f16x3 GEMM variants (X3_KERNEL=v1|P|auto) on the hot-path shapes.

Uses the library's HIP-event profiler (class 3 = the f16x3 GEMM launch alone, without the A split) and checks
every variant against the fp32-MFMA GEMM.  Run one process per variant: the choice is read once.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd import _lib, ops
from _diag import use_lib_from_env

use_lib_from_env()      # HGL_LIB_NAME=libhybridgl_diag.so: the experiment switches (HGL_X3_STAGGER, HGL_X3_GM, ...) are read

SHAPES = [  # (name, M, N, K, act, residual)
    ("clip qkv", 25216, 2304, 768, "none", False),
    ("clip out", 25216, 768, 768, "none", True),
    ("clip fc1", 25216, 3072, 768, "quickgelu", False),
    ("clip fc2", 25216, 768, 3072, "none", True),
    ("clip qkv N", 12608, 2304, 768, "none", False),
    ("sam qkv win", 4900, 3840, 1280, "none", False),
    ("sam proj win", 4900, 1280, 1280, "none", False),
    ("sam lin1", 4096, 5120, 1280, "gelu", False),
    ("sam lin2", 4096, 1280, 5120, "none", True),
    ("sam qkv glob", 4096, 3840, 1280, "none", False),
    ("sam proj glob", 4096, 1280, 1280, "none", False),
    ("k2560", 4096, 3840, 2560, "none", False),
    ("k5120", 4096, 3840, 5120, "none", False),
    ("h1280", 4096, 1920, 1280, "none", False),
    ("h5120", 4096, 1920, 5120, "none", False),
    ("text qkv", 693, 1536, 512, "none", False),
    ("ragged", 1000, 1000, 192, "relu", True),
]


GEM_SHAPES = [  # the GEM ViT-B/16 at 448x448 (785 tokens) and the text encoder with the GEM prompts (12 x 77)
    ("gem qkv", 785, 2304, 768, "none", False),
    ("gem out", 785, 768, 768, "none", True),
    ("gem fc1", 785, 3072, 768, "quickgelu", False),
    ("gem fc2", 785, 768, 3072, "none", True),
    ("text qkv", 924, 1536, 512, "none", False),
    ("text out", 924, 512, 512, "none", True),
    ("text fc1", 924, 2048, 512, "quickgelu", False),
    ("text fc2", 924, 512, 2048, "none", True),
]
SAM2_SHAPES = [  # the SAM ViT-H encoder GEMMs for one image and for two images stacked along M
    ("qkv win x1", 4900, 3840, 1280, "none", False), ("qkv win x2", 9800, 3840, 1280, "none", False),
    ("proj win x1", 4900, 1280, 1280, "none", False), ("proj win x2", 9800, 1280, 1280, "none", False),
    ("lin1 x1", 4096, 5120, 1280, "gelu", False), ("lin1 x2", 8192, 5120, 1280, "gelu", False),
    ("lin2 x1", 4096, 1280, 5120, "none", True), ("lin2 x2", 8192, 1280, 5120, "none", True),
    ("qkv glob x1", 4096, 3840, 1280, "none", False), ("qkv glob x2", 8192, 3840, 1280, "none", False),
    ("proj glob x1", 4096, 1280, 1280, "none", False), ("proj glob x2", 8192, 1280, 1280, "none", False),
]
GROUP8_SHAPES = [  # the GEMMs of the grouped pipeline (8 refs per group)
    ("clip qkv", 201728, 2304, 768, "none", False), ("clip out", 201728, 768, 768, "none", True),
    ("clip fc1", 201728, 3072, 768, "quickgelu", False), ("clip fc2", 201728, 768, 3072, "none", True),
    ("sam qkv", 32768, 3840, 1280, "none", False), ("sam proj", 32768, 1280, 1280, "none", True),
    ("sam lin1", 32768, 5120, 1280, "gelu", False), ("sam lin2", 32768, 1280, 5120, "none", True),
    ("text qkv", 7392, 1536, 512, "none", False), ("text fc1", 7392, 2048, 512, "quickgelu", False),
    ("gem qkv", 6280, 2304, 768, "none", False), ("gem fc2", 6280, 768, 3072, "none", True),
]
GROUP16_SHAPES = [  # the residual GEMMs of a group of 16 refs next to their twins WITHOUT the residual (same M, N, K)
    ("clip out", 201728, 768, 768, "none", True), ("clip out -R", 201728, 768, 768, "none", False),
    ("clip fc2", 201728, 768, 3072, "none", True), ("clip fc2 -R", 201728, 768, 3072, "none", False),
    ("sam proj", 65536, 1280, 1280, "none", True), ("sam proj -R", 65536, 1280, 1280, "none", False),
    ("sam lin2", 65536, 1280, 5120, "none", True), ("sam lin2 -R", 65536, 1280, 5120, "none", False),
    ("clip qkv", 201728, 2304, 768, "none", False), ("clip fc1", 201728, 3072, 768, "quickgelu", False),
    ("sam qkv", 65536, 3840, 1280, "none", False), ("sam lin1", 65536, 5120, 1280, "gelu", False),
]
GROUP10_SHAPES = [  # the residual GEMMs of a group of 10 refs (the driver's 20 timed steps = 10 + 10)
    ("clip out", 126080, 768, 768, "none", True), ("clip fc2", 126080, 768, 3072, "none", True),
    ("sam proj", 40960, 1280, 1280, "none", True), ("sam lin2", 40960, 1280, 5120, "none", True),
    ("sam qkv", 40960, 3840, 1280, "none", False), ("sam lin1", 40960, 5120, 1280, "gelu", False),
]
if os.environ.get("X3_SHAPES") == "group10":
    SHAPES = GROUP10_SHAPES
if os.environ.get("X3_SHAPES") == "group16":
    SHAPES = GROUP16_SHAPES
if os.environ.get("X3_SHAPES") == "group8":
    SHAPES = GROUP8_SHAPES
if os.environ.get("X3_SHAPES") == "gem":
    SHAPES = GEM_SHAPES
if os.environ.get("X3_SHAPES") == "sam2":
    SHAPES = SAM2_SHAPES


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    kind = os.environ.get("X3_KERNEL", "auto")      # v1 | P | auto, through hgl_gemm_f16x3_select
    ops.select_x3_kernel(kind)
    tot_ms = tot_fl = 0.0
    for name, M, N, K, act, res in SHAPES:
        torch.manual_seed(0)
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev) if res else None
        ref = ops.gemm(A, W, b, R, act)
        bal = os.environ.get("X3_BALANCED") == "1"      # the row-balanced launch (whole rounds + split-K tail) of the model code
        out = ops.gemm_f16x3(A, W, b, R, act, balanced=bal)
        err = float((out - ref).abs().max() / ref.abs().max())
        for _ in range(3):
            ops.gemm_f16x3(A, W, b, R, act, out=out, balanced=bal)
        torch.cuda.synchronize()
        # stream time of 10 back-to-back GEMMs (events around the loop: launch gaps, split-K tail and its reduce pass included)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm_f16x3(A, W, b, R, act, out=out, balanced=bal)
        e1.record()
        torch.cuda.synchronize()
        wall_us = e0.elapsed_time(e1) * 100.0
        lib.hgl_prof_enable(1)
        flush = torch.empty(1 << 28, device=dev) if os.environ.get("X3_COLD") else None   # 1 GiB: evicts L2 + MALL
        for _ in range(10):
            if flush is not None:
                flush.fill_(1.0)
            ops.gemm_f16x3(A, W, b, R, act, out=out, balanced=bal)
        torch.cuda.synchronize()
        lib.hgl_prof_enable(0)
        # classes 3 (register-staged), 4 (LDS-DMA), 5 (< 256 tiles: the split-K tail of a balanced launch lands here) summed:
        # kernel time of ONE GEMM = all its launches (the reduce pass of a split-K tail is not a GEMM launch: its ~10 us are
        # not in this figure -- the pipeline numbers are the judge of the balanced launch)
        class V:
            value = 0.0
        n, ms, fl = V(), V(), V()
        for cls in (3, 4, 5):
            n_, ms_, fl_, by_ = C.c_longlong(), C.c_double(), C.c_double(), C.c_double()
            lib.hgl_prof_read(cls, C.byref(n_), C.byref(ms_), C.byref(fl_), C.byref(by_))
            ms.value += ms_.value
            fl.value += fl_.value
        n.value = 10
        tf = fl.value / ms.value / 1e9
        if not name.startswith(("text", "ragged", "k", "h")) and not name.endswith("-R"):
            tot_ms += ms.value / n.value
            tot_fl += fl.value / n.value
        print(f"x3[{kind}] {name:14s} M={M:6d} N={N:5d} K={K:5d} {ms.value / n.value * 1e3:8.1f} us {tf:7.1f} TF/s  stream {wall_us:8.1f} us {2.0 * M * N * K / wall_us / 1e6:7.1f} TF/s  relerr {err:.1e}"
              + ("  MISMATCH" if not err < 2e-6 else ""))
    print(f"x3[{kind}] total {tot_ms * 1e3:8.1f} us  {tot_fl / tot_ms / 1e9:7.1f} TF/s")


if __name__ == "__main__":
    main()
