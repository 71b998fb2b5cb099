#!/usr/bin/env python3
"""SAM mask decoder at small prompt batches (64 = one RefCOCO image, 256): the raw-token path (fusion bit 5: key ranges cut into
eight workgroups per prompt at <= 128 prompts) beside the projected path (bit 5 off).  usage: decoder_small_batch.py"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from hybridgl_amd import sam as hsam, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
m = hsam.sam_model_registry["vit_h"](device=dev)
emb = torch.randn(4096, 256, device=dev)
for side in (8, 16):
    p01 = torch.from_numpy(((hsam.build_point_grid(side) * 1024 + 0.5) / 1024).astype(np.float32)).to(dev)
    for mask in (0x7fffffff, 0x7fffffff & ~32):
        lib.hgl_sam_decoder_fusion(mask)
        for _ in range(5): m.decode_points(emb, p01)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(50): m.decode_points(emb, p01)
        torch.cuda.synchronize()
        print(f"{side*side} prompts, fusion {'raw (bit 5 on)' if mask & 32 else 'projected (bit 5 off)'}: {(time.time()-t0)/50*1e3:.3f} ms per call")
lib.hgl_sam_decoder_fusion(0x7fffffff)
