#!/usr/bin/env python3
"""Run one attention shape a few times (for rocprofv3 counter passes): attn_one.py B H S hd [relpos_size] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd import ops

B, H, S, hd = (int(v) for v in sys.argv[1:5])
size = int(sys.argv[5]) if len(sys.argv) > 5 else 0
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
ops.set_precision(ops.default_precision())
q, k, v = (torch.randn(B, S, H * hd, device=dev) for _ in range(3))
kw = {}
if size:
    kw = dict(rel_h=torch.randn(B * H, S, size, device=dev), rel_w=torch.randn(B * H, S, size, device=dev))
for _ in range(iters):
    ops.attention(q, k, v, H, **kw)
torch.cuda.synchronize()
