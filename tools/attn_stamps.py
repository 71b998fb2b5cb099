#!/usr/bin/env python3
"""In-kernel cycle stamps of the persistent wide attention kernel.  Needs attention.hip compiled with the stamps in:
    touch hybridgl_amd/csrc/attention.hip && make -C hybridgl_amd/csrc EXTRA=-DHGL_ATTN_STAMPS      (and again without EXTRA afterwards)

where the cycles of an item go, for workgroup 0, wave argv[1] (default 0).  CLIP shape 1024 x 12 x 197 x 64."""
import ctypes as C
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd import _lib, ops

if os.environ.get('HGL_LIB_NAME'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.__file__), os.environ['HGL_LIB_NAME'])

dev = torch.device("cuda:0")
lib = _lib.load()
ops.set_precision("f16x3")
wave = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, H, S, hd = 1024, 12, 197, 64
q, k, v = (torch.randn(B, S, H * hd, device=dev) for _ in range(3))
buf = (C.c_ulonglong * 2048)()
run = lambda: ops.attention(q, k, v, H)
run()
torch.cuda.synchronize()
lib.hgl_debug_attn_stamps(buf, 2048, wave)   # clears, selects the wave
run()
torch.cuda.synchronize()
lib.hgl_debug_attn_stamps(buf, 2048, wave)
st = [(x >> 48, x & ((1 << 48) - 1)) for x in buf if x]
names = {1: "item start", 2: "Q split done", 3: "chunk 0 stored, loads issued", 4: "barrier", 5: "next chunk stored / loads issued", 6: "QK^T of a tile",
         7: "softmax of the tile", 12: "PV of the chunk's last tile", 13: "chunk barrier", 14: "output stored"}
acc = defaultdict(lambda: [0, 0])
for (i0, t0), (i1, t1) in zip(st, st[1:]):
    acc[(i0, i1)][0] += t1 - t0
    acc[(i0, i1)][1] += 1
items = sum(1 for i, _ in st if i == 14) or 48
tot = st[-1][1] - st[0][1]
print(f"wave {wave}: {len(st)} stamps, {items} items, {tot / max(items, 1):.0f} clock64 ticks per item")
for (i0, i1), (d, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"  {names.get(i0, i0):34s} -> {names.get(i1, i1):34s} {d / max(items, 1):9.0f} per item ({n / max(items, 1):.1f} x {d / n:7.0f})")
