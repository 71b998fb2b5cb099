#!/usr/bin/env python3
"""SAM proposal stage in the PhraseCut configuration (Hybridgl_main_PhraseCut.py:56-62: 64x64 points, one crop
layer, downscale 2, min area 100) on a synthetic image with seeded ViT-H weights: 5 encoder passes, 128 decoder
batches of 64 prompts, per-crop NMS over up to 12288 candidates, cross-crop NMS.  Prints the stage times.
With random weights the predicted IoUs / stability scores are noise, so the two thresholds are opened (every
candidate reaches the NMS: the worst case for the device path)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd import sam as hsam
from hybridgl_amd.synth import synth_image


def main():
    pps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    ppb = int(sys.argv[2]) if len(sys.argv) > 2 else 64          # prompts per decoder launch (points_per_batch)
    dev = torch.device("cuda:0")
    m = hsam.sam_model_registry["vit_h"](device=dev)
    gen = hsam.SamAutomaticMaskGenerator(m, points_per_side=pps, points_per_batch=ppb, pred_iou_thresh=0.0, stability_score_thresh=0.0,
                                         crop_n_layers=1, crop_n_points_downscale_factor=2, min_mask_region_area=100)
    img = synth_image(480, 640, 9)
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        out = gen.generate_device_crops(img)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"run {it} ({ppb} prompts per launch): {pps}x{pps} points + 4 crops of {pps // 2}x{pps // 2}: {dt * 1e3:.1f} ms, {out[0].shape[0]} masks out, "
              f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
