#!/usr/bin/env python3
"""How long the SAME attention launch (CLIP: 2048 images x 12 heads x 197 tokens x 64, attn_x3q_kernel, the pipeline's
interleaved q | k | v layout) takes depending on what ran on the GPU just before it: nothing (idle clocks), four large
split-fp16 GEMMs, the in-projection GEMM that produces its input, both.  The launch is timed alone (events around it).
Round 5: 1190 us after hipBLAS fp32 GEMMs, 1380-1540 us after the split-fp16 GEMMs, 1730-1780 us inside the pipeline
(profiles/r05c_attn_after_gemm.txt): the step is bound by the power the matrix-core GEMMs draw, and the kernels between them
inherit their clock."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hybridgl_amd import _lib, ops
from hybridgl_amd.ops import _dev, _stream, check, MASK
dev = torch.device("cuda:0")
lib = _lib.load()
ops.set_precision("f16x3")
B, H, S, hd = 2048, 12, 197, 64
D = H * hd
qkv = torch.randn(B, S, 3 * D, device=dev)
q, k, v = (qkv[..., i * D:(i + 1) * D].contiguous() for i in range(3))
out = torch.empty(B, S, D, device=dev)
A = torch.randn(65536, 1280, device=dev); W = torch.randn(3840, 1280, device=dev) * 0.02; Co = torch.empty(65536, 3840, device=dev)
X = torch.randn(B * S, D, device=dev); Wq = torch.randn(3 * D, D, device=dev) * 0.02
ops.gemm_f16x3(A, W, out=Co); ops.gemm_f16x3(X, Wq, out=qkv.view(B * S, 3 * D))
def att_contig():
    check(lib.hgl_attention_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, S, S, hd, D, D, D, D, S*D, S*D, S*D, S*D, hd ** -0.5, MASK["none"], None, 0, 0, None, None, 0, 0, _stream()), "a")
def att_inter():
    p = qkv.data_ptr()
    check(lib.hgl_attention_f32(p, p + 4*D, p + 8*D, out.data_ptr(), B, H, S, S, hd, 3*D, 3*D, 3*D, D, S*3*D, S*3*D, S*3*D, S*D, hd ** -0.5, MASK["none"], None, 0, 0, None, None, 0, 0, _stream()), "a")
def timed(fn, heat, iters=8):
    tot = 0.0
    for it in range(iters + 2):
        if heat == 1:
            for _ in range(4): ops.gemm_f16x3(A, W, out=Co)
        if heat == 2:
            ops.gemm_f16x3(X, Wq, out=qkv.view(B * S, 3 * D))
        if heat == 3:
            for _ in range(4): ops.gemm_f16x3(A, W, out=Co)
            ops.gemm_f16x3(X, Wq, out=qkv.view(B * S, 3 * D))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        if it >= 2: tot += a.elapsed_time(b)
    return tot / iters * 1e3
for name, fn in (("interleaved qkv ld 2304", att_inter),):
    for heat in (0, 1, 2, 3):
        print(f"x3q B {B} {name:26s} heat {heat}: {timed(fn, heat):8.1f} us", flush=True)
