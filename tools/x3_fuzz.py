#!/usr/bin/env python3
"""Fuzz of the two f16x3 GEMM tilings against each other: random (M, N, K), bias / residual / activation, many workgroup
tile counts (one tile per workgroup up to many tiles per persistent workgroup).  The ping-pong kernel's waits are counted by
hand (LDS-DMA units, write-out stores, bias / row-map loads on one in-order counter), so an under-count would show up as
a result that differs from the register-staged kernel, whose waits the compiler counts; both accumulate in the same
order and must agree BIT FOR BIT.  Every case is repeated to catch timing-dependent differences.
X3_FUZZ_FP16_W=1: every second case uses an fp16-valued weight (the NT = 2 instantiation: two products per step, W_lo plane not
staged, its own wait counts).
usage: x3_fuzz.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hybridgl_amd import ops

dev = torch.device("cuda:0")
ops.set_precision("f16x3")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(n_cases):
    K = 64 * int(rng.integers(1, 41))
    N = 4 * int(rng.integers(16, 900))
    M = int(rng.integers(1, 60000)) if rng.random() < 0.8 else int(rng.integers(60000, 250000))
    if M * (N + K) > 400e6:
        M = int(400e6 // (N + K))
    act = ("none", "quickgelu", "gelu", "relu")[int(rng.integers(0, 4))]
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    if os.environ.get("X3_FUZZ_FP16_W", "0") == "1" and c % 2 == 0:
        W = W.half().float()
    b = torch.randn(N, device=dev) if rng.random() < 0.7 else None
    R = torch.randn(M, N, device=dev) if rng.random() < 0.5 else None
    ops.select_x3_kernel("v1")
    ref = ops.gemm_f16x3(A, W, b, R, act)
    ops.select_x3_kernel("P")
    for rep in range(3):
        out = ops.gemm_f16x3(A, W, b, R, act)
        if not torch.equal(out, ref):
            bad += 1
            d = (out - ref).abs()
            print(f"MISMATCH case {c} rep {rep}: M={M} N={N} K={K} act={act} bias={b is not None} R={R is not None} max|d|={float(d.max()):.3e} "
                  f"at {int(d.argmax()) // N},{int(d.argmax()) % N}")
            break
    torch.cuda.synchronize()
    ops.release_split_weights(list(ops._split_cache))
ops.select_x3_kernel("auto")
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
