#!/usr/bin/env python3
"""Time the GEM heat-map stage alone (ViT-B/16 at 448x448, 3 prompts, 640x640 output) on cuda:0."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hybridgl_amd import gem as G
from hybridgl_amd.backbone import CLIPViTFM

dev = torch.device("cuda:0")
clip = CLIPViTFM("ViT-B/16", seed=0, device=dev)
gm = G.create_gem_model("ViT-B/16", clip=clip)
img = torch.from_numpy(np.random.default_rng(0).standard_normal((3, 448, 448)).astype(np.float32)).to(dev)
txt = torch.from_numpy(np.random.default_rng(1).standard_normal((3, 512)).astype(np.float32)).to(dev)
n = int(os.environ.get("N", "20"))


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


feat = gm.image_features(img)
print(f"image_features {timed(lambda: gm.image_features(img)):.3f} ms")
print(f"heatmap x3     {timed(lambda: gm.heatmap(feat, txt, 448)):.3f} ms")
m = gm.heatmap(feat, txt, 448)
print(f"resize_aa x3   {timed(lambda: G.resize_antialias(m, (640, 640))):.3f} ms")
