#!/usr/bin/env python3
"""A/B of the pre-split attention (csrc/attention_ps.hip) against the kernels that split the fp32 qkv themselves, through the
SAM encoder (vit_h_d2: one windowed + one global block, 16 images) with the per-class HIP-event timers of the library:
attention class and the f16x3 GEMM classes (the in-projection's write-out now splits)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd import _lib, ops, weights
from hybridgl_amd import sam as hsam
from hybridgl_amd.synth import synth_image

dev = torch.device("cuda:0")
lib = _lib.load()
ops.set_precision("f16x3")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
name = sys.argv[2] if len(sys.argv) > 2 else "vit_h_d2"
cfg = weights.SAM_CONFIGS[name]
m = hsam.Sam(weights.sam_state_dict(name, 0), cfg, dev)
imgs = [torch.from_numpy(synth_image(1024, 1024, 20 + i)).to(dev) for i in range(nb)]


def read(cls):
    n, ms, fl, by = C.c_longlong(), C.c_double(), C.c_double(), C.c_double()
    lib.hgl_prof_read(cls, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by))
    return n.value, ms.value, fl.value


outs = {}
for on in (0, 1, 0, 1):
    lib.hgl_attention_presplit(on)
    for _ in range(2):
        m.encode_batch(imgs)
    torch.cuda.synchronize()
    lib.hgl_prof_enable(1)
    e = m.encode_batch(imgs)
    torch.cuda.synchronize()
    lib.hgl_prof_enable(0)
    outs[on] = e.clone()
    a = read(1)
    g = [read(c) for c in (3, 4, 5)]
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    for _ in range(3):
        m.encode_batch(imgs)
    a1.record()
    torch.cuda.synchronize()
    print(f"presplit={on}: attention {a[0]} launches {a[1] * 1e3:9.1f} us ({a[2] / max(a[1], 1e-9) / 1e9:6.1f} TF/s) | "
          f"x3 GEMM classes {sum(x[1] for x in g) * 1e3:9.1f} us ({sum(x[2] for x in g) / max(sum(x[1] for x in g), 1e-9) / 1e9:6.1f} TF/s) | "
          f"encoder {a0.elapsed_time(a1) / 3:8.3f} ms")
print("bit-identical:", bool(torch.equal(outs[0], outs[1])), "max |diff|", float((outs[0] - outs[1]).abs().max()))
