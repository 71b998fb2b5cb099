#!/usr/bin/env python3
"""PhraseCut-shaped items through HybridGLPipeline.run, stages back to back on one stream (serial=True), for a per-kernel
breakdown under `rocprofv3 --kernel-trace --stats`: 480x640 image, heavy AMG (64x64 points + one crop layer, filters open, 512
prompts per decoder launch), <= 256 of SAM's masks into CLIP G2L&L2G, 8 phrases.  usage: phrasecut_profile.py [images]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd.backbone import CLIPViTFM
from hybridgl_amd.gem import create_gem_model
from hybridgl_amd.pipeline import HybridGLPipeline, synthetic_ref
from hybridgl_amd.sam import SamAutomaticMaskGenerator, sam_model_registry

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
model = CLIPViTFM("ViT-B/16", seed=0, device=dev)
sam = sam_model_registry["default"](seed=0, device=dev)
gen = SamAutomaticMaskGenerator(sam, points_per_side=64, points_per_batch=int(os.environ.get("PC_PPB", "1024")), pred_iou_thresh=-1e30, stability_score_thresh=0.0,
                                crop_n_layers=1, crop_n_points_downscale_factor=2, min_mask_region_area=100)
pipe = HybridGLPipeline(model, "G2L&L2G", 9, mask_generator=gen, use_sam_masks=True, gem_model=create_gem_model("ViT-B/16", clip=model))
refs = [synthetic_ref(100 + j, dev, N=64, H=480, W=640, n_sent=8, sam_img_size=1024, gem=True, device_blur=True)[0] for j in range(2)]
for it in range(2):
    if it == 1:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    pipe.run((refs[i % 2] for i in range(n)), group=2, proposal_cap=256, serial=True)
torch.cuda.synchronize()
print(f"serial PhraseCut items: {(time.perf_counter() - t0) / n * 1e3:.1f} ms per image")
