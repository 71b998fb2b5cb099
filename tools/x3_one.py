#!/usr/bin/env python3
"""Run one f16x3 GEMM shape a few times (for rocprofv3 counter passes).  usage: x3_one.py M N K [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd import ops
from _diag import use_lib_from_env

use_lib_from_env()      # HGL_LIB_NAME=libhybridgl_diag.so: HGL_X3_GM and friends are read
ops.select_x3_kernel(os.environ.get("X3_KERNEL", "auto"))

M, N, K = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) / K ** 0.5
if os.environ.get("X3_FP16_VALUED_W", "0") == "1":     # what an OpenAI CLIP archive holds: W_lo == 0 -> the NT = 2 instantiation
    W = W.half().float()
b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev)
for _ in range(iters):
    ops.gemm_f16x3(A, W, b, None, "none", out=out)
torch.cuda.synchronize()
