#!/usr/bin/env python3
"""Runs one GEMM shape N times (for rocprofv3 --pmc passes): python tools/gemm_one.py M N K [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd import ops

M, N, K = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev) / K ** 0.5
out = torch.empty(M, N, device=dev)
x3 = os.environ.get("X3") == "1"
for _ in range(iters):
    (ops.gemm_f16x3 if x3 else ops.gemm)(A, W, None, None, "none", out=out)
torch.cuda.synchronize()
