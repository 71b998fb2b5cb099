#!/usr/bin/env python3
"""In-kernel cycle stamps of the pre-split attention kernel (csrc/attention_ps.hip).  Needs a library with the stamps in:
    make -C hybridgl_amd/csrc stamps      (-> hybridgl_amd/libhybridgl_stamps.so; the product library is untouched)
    HGL_LIB_NAME=libhybridgl_stamps.so python tools/attn_ps_stamps.py [wave] [workgroup] [images]
Where the cycles of one wave of one workgroup go (SAM windows, vit_h_d2, the last attn_ps launch of the encoder call)."""
import ctypes as C
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd import _lib, ops, weights

if os.environ.get('HGL_LIB_NAME'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.__file__), os.environ['HGL_LIB_NAME'])
from hybridgl_amd import sam as hsam
from hybridgl_amd.synth import synth_image

dev = torch.device("cuda:0")
lib = _lib.load()
ops.set_precision("f16x3")
wave = int(sys.argv[1]) if len(sys.argv) > 1 else 0
block = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 16
name = sys.argv[4] if len(sys.argv) > 4 else "vit_h_w1"
cfg = dict(weights.SAM_CONFIGS["vit_h_d2"])
if name == "vit_h_w1":      # one windowed block only: the stamps are those of the window kernel
    cfg["depth"], cfg["global_attn_indexes"] = 1, ()
    weights.SAM_CONFIGS["vit_h_w1"] = cfg
m = hsam.Sam(weights.sam_state_dict(name, 0), weights.SAM_CONFIGS[name], dev)
imgs = [torch.from_numpy(synth_image(1024, 1024, 20 + i)).to(dev) for i in range(nb)]
buf = (C.c_ulonglong * 1024)()
lib.hgl_debug_ps_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
m.encode_batch(imgs)
torch.cuda.synchronize()
for blk, wv in ((block, wave), (block + 8, wave), (block, 3), (8, 0)):
    lib.hgl_debug_ps_stamps(buf, 1024, blk, wv)   # clears, selects
    m.encode_batch(imgs)
    torch.cuda.synchronize()
    lib.hgl_debug_ps_stamps(buf, 1024, blk, wv)
    st = [(x >> 48, x & ((1 << 48) - 1)) for x in buf if x]
    names = {1: "start", 2: "chunk 0 issued", 3: "Q loaded", 4: "rel-pos ready", 10: "iteration top", 11: "vmcnt(0) passed", 12: "barrier passed",
             13: "next chunk issued", 14: "QK^T issued", 15: "softmax done", 16: "PV issued", 20: "loop done", 21: "stored"}
    acc = defaultdict(lambda: [0, 0])
    for (i0, t0), (i1, t1) in zip(st, st[1:]):
        acc[(i0, i1)][0] += t1 - t0
        acc[(i0, i1)][1] += 1
    if not st:
        print(f"workgroup {blk} wave {wv}: no stamps")
        continue
    tot = st[-1][1] - st[0][1]
    print(f"workgroup {blk} wave {wv}: {len(st)} stamps, {tot} clock64 ticks start to end")
    for (i0, i1), (d, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print(f"  {names.get(i0, i0):18s} -> {names.get(i1, i1):18s} {d:8d} total ({n} x {d / n:7.0f})")
