#!/usr/bin/env python3
"""Micro-benchmark of hgl_gemm_f32 / hgl_attention_f32 on the shapes of the hot path (GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd import ops

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
    ("dec kproj", 262144, 128, 256, "none", False),
    ("dec outproj", 262144, 256, 128, "none", True),
    ("dec up0", 262144, 256, 256, "none", False),
    ("dec up3", 1048576, 128, 64, "gelu", False),
    ("text qkv", 693, 1536, 512, "none", False),
]


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda:0")
    tot_f = tot_t = 0.0
    for name, M, N, K, act, res in SHAPES:
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev) if res else None
        out = torch.empty(M, N, device=dev)
        ms = bench(lambda: ops.gemm(A, W, b, R, act, out=out))
        tf = 2.0 * M * N * K / ms / 1e9
        line = f"gemm {name:14s} M={M:7d} N={N:5d} K={K:5d} f32 {ms:8.3f} ms {tf:7.1f} TF/s"
        if K % 64 == 0:
            ms3 = bench(lambda: ops.gemm_f16x3(A, W, b, R, act, out=out))
            ref = ops.gemm(A, W, b, R, act)
            err = float((ops.gemm_f16x3(A, W, b, R, act) - ref).abs().max() / ref.abs().max())
            line += f" | f16x3 {ms3:8.3f} ms {2.0 * M * N * K / ms3 / 1e9:7.1f} TF/s (incl. A split) relerr {err:.1e}"
        print(line)
    for name, B, H, S, hd in [("clip", 128, 12, 197, 64), ("sam win", 25, 16, 196, 80), ("sam glob", 1, 16, 4096, 80),
                              ("text", 9, 8, 77, 64)]:
        q, k, v = (torch.randn(B, S, H * hd, device=dev) for _ in range(3))
        ms = bench(lambda: ops.attention(q, k, v, H))
        print(f"attn {name:10s} B={B} H={H} S={S} hd={hd} {ms:8.3f} ms {4.0 * B * H * S * S * hd / ms / 1e9:7.1f} TF/s")


if __name__ == "__main__":
    main()
