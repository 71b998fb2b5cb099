"""Where does the first-use cost of the evaluation loop go?  The headline workload of bench.py (scope B, G2L, ViT-B/16,
SAM's own masks, GEM on the device) as successive HybridGLPipeline.run calls of `--warmup` then `--steps` refs (x reps),
each bracketed by a synchronize, with the caching allocator's counters (device mallocs, reserved bytes) and the host-side
time stamp of every group boundary.   usage: python3 tools/first_use.py [--warmup 5] [--steps 20] [--reps 3] [--prepare]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--group", type=int, default=16)
    ap.add_argument("--prepare", action="store_true", help="call HybridGLPipeline.prepare before the warm-up")
    args = ap.parse_args()
    from hybridgl_amd.backbone import CLIPViTFM
    from hybridgl_amd.gem import create_gem_model
    from hybridgl_amd.pipeline import HybridGLPipeline, synthetic_ref
    from hybridgl_amd.sam import SamAutomaticMaskGenerator, sam_model_registry
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model = CLIPViTFM("ViT-B/16", seed=0, device=dev)
    sam = sam_model_registry["default"](seed=0, device=dev)
    gen = SamAutomaticMaskGenerator(sam, points_per_side=8, pred_iou_thresh=-1e30, stability_score_thresh=0.0, box_nms_thresh=2.0,
                                    crop_n_layers=0, crop_n_points_downscale_factor=1, min_mask_region_area=800)
    gem = create_gem_model("ViT-B/16", clip=model)
    pipe = HybridGLPipeline(model, fusion_mode="G2L", masking_block=9, mask_generator=gen, use_sam_masks=True, gem_model=gem)
    refs = [synthetic_ref(j, dev, N=64, sam_img_size=1024, gem=True, device_blur=True)[0] for j in range(16)]
    torch.cuda.synchronize()

    def stats():
        s = torch.cuda.memory_stats(dev)
        return s.get("num_device_alloc", 0), s.get("reserved_bytes.all.current", 0) / 2**30

    out = []

    def leg(name, k):
        a0, r0 = stats()
        pipe.group_marks = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = pipe.run((refs[i % len(refs)] for i in range(k)), group=args.group, proposal_cap=64)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        a1, r1 = stats()
        marks = [round((m - t0) * 1e3, 1) for m in getattr(pipe, "group_marks", [])]
        out.append({"leg": name, "refs": n, "ms": round(dt * 1e3, 1), "ms_per_ref": round(dt * 1e3 / max(n, 1), 2),
                    "host_ms": round(t_host * 1e3, 1), "device_mallocs": a1 - a0, "reserved_GiB": [round(r0, 2), round(r1, 2)],
                    "group_marks_ms": marks})
        print(json.dumps(out[-1]), flush=True)

    if args.prepare:
        t0 = time.perf_counter()
        pipe.prepare(group=args.group, H=640, W=640, proposals=64, n_sent=3)
        torch.cuda.synchronize()
        print(json.dumps({"leg": "prepare", "ms": round((time.perf_counter() - t0) * 1e3, 1), "stats": stats()}), flush=True)
    leg("warmup", args.warmup)
    for r in range(args.reps):
        leg(f"timed{r}", args.steps)
    leg("steady48", 48)
    leg("steady48b", 48)


if __name__ == "__main__":
    main()
