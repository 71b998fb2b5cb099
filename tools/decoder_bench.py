#!/usr/bin/env python3
"""SAM mask decoder alone: 64 point prompts against one ViT-H image embedding (the unit the PhraseCut
configuration runs 128 times per image).  usage: decoder_bench.py [iters] [points per side: 8 -> 64 prompts, 23 -> 529]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hybridgl_amd import sam as hsam


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    side = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda:0")
    m = hsam.sam_model_registry["vit_h"](device=dev)
    emb = torch.randn(4096, 256, device=dev)
    p01 = torch.from_numpy(((hsam.build_point_grid(side) * 1024 + 0.5) / 1024).astype(np.float32)).to(dev)
    for _ in range(3):
        m.decode_points(emb, p01)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        m.decode_points(emb, p01)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / iters * 1e3
    print(f"decoder, {side * side} prompts: {dt:.3f} ms per batch ({dt * 64 / (side * side):.3f} ms per 64 prompts)")


if __name__ == "__main__":
    main()
