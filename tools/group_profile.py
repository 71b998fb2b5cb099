#!/usr/bin/env python3
"""One group of 8 refs through the pipeline with its stages back to back on ONE stream (HybridGLPipeline.run(serial=True):
SAM's own masks into clean-up + CLIP, counts read back), a few
times: under `rocprofv3 --kernel-trace --stats` this gives the isolated per-kernel times of the grouped pipeline
(no overlap between streams except the text / GEM side stream).  usage: group_profile.py [groups] [group_size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd.backbone import CLIPViTFM
from hybridgl_amd.gem import create_gem_model
from hybridgl_amd.pipeline import HybridGLPipeline, synthetic_ref
from hybridgl_amd.sam import SamAutomaticMaskGenerator, sam_model_registry

n_groups = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
model = CLIPViTFM("ViT-B/16", seed=0, device=dev)
sam = sam_model_registry["default"](seed=0, device=dev)
gen = SamAutomaticMaskGenerator(sam, points_per_side=8, pred_iou_thresh=-1e30, stability_score_thresh=0.0, box_nms_thresh=2.0,
                                crop_n_layers=0, crop_n_points_downscale_factor=1, min_mask_region_area=800)
seeded = os.environ.get("HGL_PROPOSALS_FROM", "sam") == "seeded"
pipe = HybridGLPipeline(model, "G2L", 9, mask_generator=gen, use_sam_masks=not seeded, cleanup_given_masks=seeded,
                        gem_model=create_gem_model("ViT-B/16", clip=model))
refs = [synthetic_ref(j, dev, N=64, sam_img_size=1024, gem=True, device_blur=True)[0] for j in range(8)]
group = [refs[j % 8] for j in range(g)]
import time
for it in range(n_groups + 1):
    if it == 1:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    pipe.run(iter(group), group=g, proposal_cap=64 if not seeded else None, serial=True)
torch.cuda.synchronize()
print(f"serial group of {g}: {(time.perf_counter() - t0) / n_groups / g * 1e3:.2f} ms per ref")
