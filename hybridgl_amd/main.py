"""Evaluation driver with the reference's command line and report format (Hybridgl_main.py:23-261,
flags of utils.py:397-471) for the pieces this package owns.

Datasets (REFER/COCO), spaCy parsing and the GEM heat-map are host/external inputs of the reference
 and are not re-implemented: `--synthetic N` evaluates N seeded RefCOCO-shaped refs
(hybridgl_amd/synth.py) through the full device pipeline and prints/appends the same two result lines.

    python -m hybridgl_amd.main --dataset refcocog --split val --fusion_mode G2L --synthetic 8
"""
import argparse
import os

import torch


def default_argument_parser():
    """the live flags of utils.py:397-471"""
    p = argparse.ArgumentParser(description="HybridGL evaluation (MI355X-native hot path)")
    p.add_argument("--dataset", default="refcoco", choices=["refcoco", "refcoco+", "refcocog"])
    p.add_argument("--split", default="val")
    p.add_argument("--fusion_mode", default="G2L", choices=["G2L", "L2G", "G2L&L2G"])
    p.add_argument("--refer_data_root", default="./refer/data")
    p.add_argument("--synthetic", type=int, default=4, help="number of seeded synthetic refs to evaluate")
    p.add_argument("--proposals", type=int, default=64)
    p.add_argument("--sam", action="store_true", help="also run the SAM ViT-H proposal stage on every ref")
    p.add_argument("--result_dir", default="./result_log")
    return p


def main(args):
    from .backbone import CLIPViTFM
    from .pipeline import HybridGLPipeline, synthetic_ref
    assert torch.cuda.is_available(), "hybridgl_amd has no CPU path"
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    splitBy = "umd" if args.dataset == "refcocog" else "unc"          # Hybridgl_main.py:26-29
    model = CLIPViTFM(model_name="ViT-B/16", device=dev).eval()
    gen = None
    if args.sam:
        from .sam import SamAutomaticMaskGenerator, sam_model_registry
        sam = sam_model_registry["default"](device=dev)
        gen = SamAutomaticMaskGenerator(sam, points_per_side=8, pred_iou_thresh=0.7, stability_score_thresh=0.7,
                                        crop_n_layers=0, crop_n_points_downscale_factor=1, min_mask_region_area=800)
    pipe = HybridGLPipeline(model, fusion_mode=args.fusion_mode, masking_block=9, mask_generator=gen)
    print(f"fusion mode={args.fusion_mode}")
    for i in range(args.synthetic):
        ref, _ = synthetic_ref(i, dev, N=args.proposals, sam_img_size=1024 if gen else 0)
        pipe.step(ref)
    m = pipe.metrics()
    text = (f"\n\n fusion_mode={args.fusion_mode} "
            f"\nDataset: {args.dataset} / {args.split} / {splitBy}"
            f"\nOverall IoU / mean IoU"
            f"\npure hybridgl: {m['oIoU']:.2f} / {m['mIoU']:.2f}"
            f"\nhybridgl w/ spatial guidance: {m['oIoU_final']:.2f} / {m['mIoU_final']:.2f}")
    os.makedirs(args.result_dir, exist_ok=True)                         # Hybridgl_main.py:233-248
    with open(os.path.join(args.result_dir, f"result_log_{args.dataset}_{args.split}.txt"), "a") as f:
        f.write(text)
    print(text)
    return m


if __name__ == "__main__":
    main(default_argument_parser().parse_args())
