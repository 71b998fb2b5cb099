"""Evaluation driver with the reference's command line and report format (Hybridgl_main.py:23-261,
flags of utils.py:397-471) for the pieces this package owns.

`--synthetic N` evaluates N seeded RefCOCO-shaped refs (hybridgl_amd/synth.py) through the full device pipeline
and prints/appends the reference's two result lines.

`--real` walks the REFER annotations under --refer_data_root (hybridgl_amd/refer_io.py: refs(<splitBy>).p,
instances.json, COCO images; ground truth from the native polygon/RLE codec) with real checkpoints
(HYBRIDGL_CLIP_CHECKPOINT / HYBRIDGL_SAM_CHECKPOINT / HYBRIDGL_BPE_VOCAB).  The two external models of the
spaCy parse of the reference stays an input (--parse_json: {sent_id: {"noun_phrase", "other_nouns", "dirflag",
"relaflag"}}, default = whole sentence, no relation words).  The GEM heat-map of the noun phrase is computed on the
device (hybridgl_amd/gem.py, Hybridgl_main.py:200-201) unless --heatmap_dir/<sent_id>.npy supplies it; the blurred
background is OpenCV's fixed-point GaussianBlur restated on the device (Hybridgl_main.py:99).

    python -m hybridgl_amd.main --dataset refcocog --split val --fusion_mode G2L --synthetic 8
    python -m hybridgl_amd.main --dataset refcoco --split testA --real --refer_data_root ./refer/data
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m hybridgl_amd.main --real ...

Under a launcher (WORLD_SIZE > 1) rank r evaluates the items i = r (mod R) of the loader's order and the metric rows
are gathered once at the end (hybridgl_amd/dist.py); rank 0 writes the report.
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
    p.add_argument("--real", action="store_true", help="evaluate the REFER refs under --refer_data_root")
    p.add_argument("--parse_json", default="", help="pre-computed parse records keyed by sent_id")
    p.add_argument("--heatmap_dir", default="", help="pre-computed heat-maps <sent_id>.npy ([H,W] or any size, fp32)")
    p.add_argument("--heatmap", default="device", choices=["device", "given"],
                   help="device: GEM heat-maps computed here; given: --heatmap_dir files (real) / seeded maps (synthetic)")
    p.add_argument("--max_refs", type=int, default=0, help="stop after this many refs (0 = all)")
    p.add_argument("--clip_model", default="ViT-B/16")
    p.add_argument("--sam_model", default="default")
    p.add_argument("--bpe_vocab", default="", help="bpe_simple_vocab_16e6.txt.gz (or HYBRIDGL_BPE_VOCAB)")
    # proposal thresholds of Hybridgl_main.py:67-73
    p.add_argument("--points_per_side", type=int, default=8)
    p.add_argument("--pred_iou_thresh", type=float, default=0.7)
    p.add_argument("--stability_score_thresh", type=float, default=0.7)
    p.add_argument("--min_mask_region_area", type=int, default=800)
    p.add_argument("--group", type=int, default=16,
                   help="images taken at a time by the two-stream loop (HybridGLPipeline.run); 1 = ref by ref on one stream")
    p.add_argument("--workers", type=int, default=4, help="loader threads (Hybridgl_main.py:45 num_workers)")
    p.add_argument("--k_clamp", default="auto", choices=["auto", "persistent", "per_ref"],
                   help="the k1 / k2 clamp of Hybridgl_main.py:178-181: persistent = the reference's quirk (once an image yields "
                        "fewer than 3 / 6 proposals the clamp stays for every later item OF THE PROCESS: depends on the number of "
                        "ranks); per_ref = that item only (order- and sharding-independent); auto = persistent on one rank, "
                        "per_ref under sharding")
    return p


OTHER_NOUN_PREFIX = "a photo of "   # Hybridgl_main.py:160: clip.tokenize('a photo of ' + other_noun)


def sentence_strings(raw, rec):
    """The strings the reference tokenises for one sentence (Hybridgl_main.py:146-161), in the row order of
    pipeline.Sentence: [sentence, noun phrase, 'a photo of ' + other noun ...].  `rec` is the sentence's parse record
    ({"noun_phrase", "other_nouns": bare phrases as extract_nouns returns them, ...}); missing -> whole sentence, no nouns."""
    others = list(rec.get("other_nouns", []))
    return [raw, rec.get("noun_phrase", raw)] + [OTHER_NOUN_PREFIX + o for o in others]


class RealRefs:
    """The dataset side of the loop (Hybridgl_main.py:40-45,79-146): `jobs(rank, world)` = the dataset positions of this
    rank in the loader's order, `load(i)` = one RefBatch (image decode, GEM transform, strings, tokens, ground truth on the
    host; uploads on the calling thread's current stream).  `load` is what hybridgl_amd.loader.Prefetcher runs on its
    background threads."""

    def __init__(self, args, dev, splitBy, context_length):
        import json
        from .refer_io import ReferDataset
        from .tokenizer import SimpleTokenizer
        from .gem import get_gem_img_transform
        self.args, self.dev, self.context_length = args, dev, context_length
        self.ds = ReferDataset(args.refer_data_root, args.dataset, splitBy, args.split)
        self.tk = SimpleTokenizer(args.bpe_vocab or None)
        self.preprocess = get_gem_img_transform()                                       # Hybridgl_main.py:39
        self.parse = json.load(open(args.parse_json)) if args.parse_json else {}
        self.n = len(self.ds) if args.max_refs <= 0 else min(len(self.ds), args.max_refs)

    def jobs(self, rank=0, world=1):
        from .dist import shard_by_groups
        image_ids = [self.ds.refer.Refs[r]["image_id"] for r in self.ds.ref_ids[:self.n]]
        return shard_by_groups(image_ids, rank, world)   # refs of one image stay on one rank (per-image cache)

    def load(self, i):
        import numpy as np
        from . import synth
        from .gem import GEMWrapper
        from .loader import pin_upload
        from .pipeline import RefBatch, Sentence
        from .tokenizer import tokenize
        args, dev = self.args, self.dev
        data, annot, sentences = self.ds[i]
        img = data["sam_img"]
        H, W = img.shape[:2]
        strings, sents = [], []
        t = lambda a: pin_upload(a, dev)
        for sent_id, raw in zip(data["sent_ids"], sentences):
            rec = self.parse.get(str(sent_id), {})
            row = len(strings)
            others = list(rec.get("other_nouns", []))   # extract_nouns' phrases, bare (utils.py:82-98)
            strings += sentence_strings(raw, rec)
            attn = None
            if args.heatmap_dir and os.path.exists(os.path.join(args.heatmap_dir, f"{sent_id}.npy")):
                a = torch.from_numpy(np.load(os.path.join(args.heatmap_dir, f"{sent_id}.npy")).astype(np.float32))
                if tuple(a.shape) != (H, W):     # Hybridgl_main.py:201-202: bilinear to the image size
                    a = torch.nn.functional.interpolate(a[None, None], size=(H, W), mode="bilinear", align_corners=False)[0, 0]
                attn = t(a)
            gem_row = None
            if attn is None and args.heatmap == "device":
                gem_row = len(strings)                                         # Hybridgl_main.py:200 gem_model(tensor_img, [noun_phrase])
                strings += GEMWrapper.prompts([rec.get("noun_phrase", raw)])
            elif attn is None:
                attn = torch.ones((H, W), dtype=torch.float32, device=dev)     # uniform: no spatial guidance
            sents.append(Sentence(row, row + 1, list(range(row + 2, row + 2 + len(others))), rec.get("dirflag", "none"),
                                  rec.get("relaflag", "none"), len(others), attn, gem_row=gem_row))
        tokens = tokenize(strings, context_length=self.context_length, tokenizer=self.tk)   # raises on over-long text, as clip.tokenize
        placeholder = torch.zeros((1, H, W), dtype=torch.bool, device=dev)
        return RefBatch(t(img), None, t(synth.imagenet_normalize(img)), placeholder,
                        torch.zeros((1, 4), dtype=torch.int64, device=dev), t(tokens), t(annot), sents, None,
                        int(data["img_id"][0]), tensor_img=t(self.preprocess(img)) if args.heatmap == "device" else None,
                        token_len=int(tokens.argmax(axis=1).max()) + 1, index=i)


def real_refs(args, dev, splitBy, context_length, rank=0, world=1):
    """RefBatch per dataset item of this rank, in the loader's order, prepared on the calling thread (no prefetching)."""
    rr = RealRefs(args, dev, splitBy, context_length)
    for i in rr.jobs(rank, world):
        yield rr.load(i)


def main(args):
    from .backbone import CLIPViTFM
    from .pipeline import EmptyProposals, HybridGLPipeline, synthetic_ref
    from . import dist as D
    assert torch.cuda.is_available(), "hybridgl_amd has no CPU path"
    rank, local_rank, world = D.env_rank()
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    cores = D.pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))   # launch + loader threads of a rank on its own cores
    torch.cuda.set_device(dev)
    dist = D.init_process_group(os.environ.get("HYBRIDGL_DIST_BACKEND", "nccl"), dev) if world > 1 else None
    splitBy = "umd" if args.dataset == "refcocog" else "unc"          # Hybridgl_main.py:26-29
    model = CLIPViTFM(model_name=args.clip_model, device=dev).eval()
    gen = None
    if args.sam or args.real:
        from .sam import SamAutomaticMaskGenerator, sam_model_registry
        sam = sam_model_registry[args.sam_model](device=dev)
        # Hybridgl_main.py:67-73
        gen = SamAutomaticMaskGenerator(sam, points_per_side=args.points_per_side, pred_iou_thresh=args.pred_iou_thresh,
                                        stability_score_thresh=args.stability_score_thresh, crop_n_layers=0,
                                        crop_n_points_downscale_factor=1, min_mask_region_area=args.min_mask_region_area)
    gem_model = None
    if args.heatmap == "device":
        from .gem import create_gem_model
        gem_model = create_gem_model(args.clip_model, clip=model)          # Hybridgl_main.py:36-38 (same checkpoint: shared weights)
    k_clamp = args.k_clamp if args.k_clamp != "auto" else ("persistent" if world == 1 else "per_ref")
    pipe = HybridGLPipeline(model, fusion_mode=args.fusion_mode, masking_block=9, mask_generator=gen,
                            use_sam_masks=args.real, gem_model=gem_model, k_clamp=k_clamp)
    if rank == 0:
        print(f"fusion mode={args.fusion_mode}")
        if world > 1:
            print(f"ranks: {dist.get_world_size()} ({dist.get_backend()}), k_clamp={k_clamp}, host cores per rank: {len(cores) or 'unpinned'}")
    from .loader import Prefetcher
    if args.real:
        from .weights import CLIP_CONFIGS
        rr = RealRefs(args, dev, splitBy, CLIP_CONFIGS[args.clip_model]["context_length"])
        jobs, make = rr.jobs(rank, world), rr.load
    else:
        jobs = D.shard_indices(args.synthetic, rank, world)
        make = lambda i: synthetic_ref(i, dev, N=args.proposals, sam_img_size=1024 if gen else 0, gem=gem_model is not None,
                                       device_blur=True)[0]
    # Hybridgl_main.py:45,79: DataLoader(num_workers=4) feeding the loop; here loader threads feed the grouped loop
    loader = Prefetcher(jobs, make, workers=args.workers, depth=2 * args.group + 2, device=dev)
    if args.group <= 1:      # ref by ref on one stream (Hybridgl_main.py:79-230 as written)
        for ref in loader:
            try:
                pipe.step(ref)
            except EmptyProposals:
                pipe.skipped = getattr(pipe, "skipped", 0) + 1
    else:
        pipe.run(loader, group=args.group)
    if getattr(pipe, "skipped", 0):
        # the reference would fail on an image without proposals; count and go on
        print(f"{pipe.skipped} refs skipped: the proposal stage returned no mask")
    m = pipe.metrics(dist)      # one all-gather of the metric rows; identical on every rank
    if dist is not None:
        dist.destroy_process_group()
    if rank != 0:
        return m
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
