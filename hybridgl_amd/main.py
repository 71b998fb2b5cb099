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
    p.add_argument("--dataset", default="refcoco", choices=["refcoco", "refcoco+", "refcocog", "phrasecut"],
                   help="phrasecut = the Hybridgl_main_PhraseCut.py path: one item per IMAGE with all its phrases, heavy AMG")
    p.add_argument("--phrasecut_root", default="./PhraseCutDataset/data/VGPhraseCut_v0",
                   help="image_data_split.json, refer_<split>.json, images/ (data/dataset_phrasecut.py:40)")
    p.add_argument("--unseen_mode", action="store_true", help="PhraseCut: skip phrases of COCO categories (dataset_phrasecut.py:63)")
    p.add_argument("--seen_mode", action="store_true", help="PhraseCut: only phrases of COCO categories (:65)")
    p.add_argument("--split", default=None,
                   help="default: val (Hybridgl_main.py), test for --dataset phrasecut (Hybridgl_main_PhraseCut.py:42)")
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
    # proposal thresholds: Hybridgl_main.py:67-73 (8 / 0.7 / 0.7 / 800, no crops); --dataset phrasecut:
    # Hybridgl_main_PhraseCut.py:56-62 (64 / 0.86 / 0.92 / 100, one crop layer with 32 x 32 points per crop)
    p.add_argument("--points_per_side", type=int, default=None)
    p.add_argument("--pred_iou_thresh", type=float, default=None)
    p.add_argument("--stability_score_thresh", type=float, default=None)
    p.add_argument("--min_mask_region_area", type=int, default=None)
    p.add_argument("--crop_n_layers", type=int, default=None)
    p.add_argument("--crop_n_points_downscale_factor", type=int, default=None)
    p.add_argument("--points_per_batch", type=int, default=None,
                   help="prompts per decoder launch (a memory knob: 64 in the reference; default 64, PhraseCut 512)")
    p.add_argument("--box_nms_thresh", type=float, default=0.7, help="SamAutomaticMaskGenerator's default (Hybridgl_main.py:67-73 does not set it)")
    p.add_argument("--group", type=int, default=None,
                   help="images taken at a time by the two-stream loop (HybridGLPipeline.run); 1 = ref by ref on one stream; "
                        "default 16 (PhraseCut: 4 -- a heavy-AMG image holds ~5 GB of candidate masks until its counts are read)")
    p.add_argument("--workers", type=int, default=4, help="loader threads (Hybridgl_main.py:45 num_workers)")
    p.add_argument("--prepare", action="store_true",
                   help="size every workspace and touch every kernel of a full group before the loop starts (HybridGLPipeline.prepare)")
    p.add_argument("--proposal_cap", type=int, default=0,
                   help="keep at most this many proposals of the first NMS order per image (0 = all; the benchmark's fixed 64)")
    p.add_argument("--host_transforms", action="store_true",
                   help="ToTensor/Normalize and the GEM transform on the loader threads' CPU (numpy / PIL) as the reference's "
                        "DataLoader workers do, instead of bit-identical device kernels (hybridgl_amd/transforms.py)")
    p.add_argument("--stats_json", default="", help="rank 0 writes {stats, metrics} of the run here (throughput, loader wait)")
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


class _Slot:
    """one image of RealRefs' look-aside: filled by the loader thread that met the image first, awaited by the others"""
    __slots__ = ("done", "value", "error")

    def __init__(self):
        import threading
        self.done, self.value, self.error = threading.Event(), None, None


class RealRefs:
    """The dataset side of the loop (Hybridgl_main.py:40-45,79-146): `jobs(rank, world)` = the dataset positions of this
    rank in the loader's order, `load(i)` = one RefBatch.  `load` is what hybridgl_amd.loader.Prefetcher runs on its
    background threads, each with its own copy stream current:

        host    JPEG decode (PIL), BPE, ground-truth rasterisation (native codec), pinned uploads of the uint8 image,
                the tokens and the target
        device  image['image'] = ToTensor + Normalize and image['tensor_img'] = the GEM transform (bicubic resize to
                448 x 448 + Normalize), bit-identical to the host transforms (hybridgl_amd/transforms.py) -- 24 of the
                28 ms of host work an item costs the reference's workers; `device_transforms=False` restores the host
                versions (numpy / PIL)

    The dataset yields one item per REF and re-decodes the image for each (data/dataset_refer_bert.py:103-110); here
    the decoded image and its two device tensors are shared by the refs of an image that the loader meets within its
    look-ahead (`image_lru` images)."""

    def __init__(self, args, dev, splitBy, context_length, device_transforms=True, image_lru=48):
        import collections
        import json
        import threading
        from .refer_io import ReferDataset
        from .tokenizer import SimpleTokenizer
        from .gem import get_gem_img_transform
        self.args, self.dev, self.context_length = args, dev, context_length
        self.ds = ReferDataset(args.refer_data_root, args.dataset, splitBy, args.split)
        self.tk = SimpleTokenizer(args.bpe_vocab or None)
        self._tk_lock = threading.Lock()        # the BPE cache is a plain dict shared by the loader threads
        self.preprocess = get_gem_img_transform()                                       # Hybridgl_main.py:39
        self.parse = json.load(open(args.parse_json)) if args.parse_json else {}
        self.n = len(self.ds) if args.max_refs <= 0 else min(len(self.ds), args.max_refs)
        self.device_transforms = device_transforms
        self.image_lru = int(image_lru)
        self._imgs = collections.OrderedDict()
        self._lock = threading.Lock()
        self.decoded = 0        # images decoded / uploaded (<= items loaded)

    def jobs(self, rank=0, world=1):
        from .dist import shard_by_groups
        image_ids = [self.ds.image_id(i) for i in range(self.n)]
        return shard_by_groups(image_ids, rank, world)   # refs of one image stay on one rank (per-image cache)

    def _image(self, i):
        """(sam_img u8 [H,W,3], image_norm f32 [3,H,W], tensor_img f32 [3,448,448] | None) of item i on the device, valid
        on the calling thread's current stream."""
        iid = self.ds.image_id(i)
        with self._lock:
            slot = self._imgs.get(iid) if self.image_lru > 0 else None
            owner = slot is None
            if owner:
                slot = _Slot()
                if self.image_lru > 0:
                    self._imgs[iid] = slot
                    while len(self._imgs) > self.image_lru:
                        self._imgs.popitem(last=False)
            elif self.image_lru > 0:
                self._imgs.move_to_end(iid)
        if not owner:
            slot.done.wait()
            if slot.error is not None:
                raise slot.error
            ev, val = slot.value
            torch.cuda.current_stream().wait_event(ev)     # the uploads ran on another loader thread's stream
            return val
        try:
            from . import synth, transforms as T
            from .loader import pin_upload
            img = self.ds.image(i)
            sam_img = pin_upload(img, self.dev)
            want_gem = self.args.heatmap == "device"
            if self.device_transforms:
                norm = T.to_tensor_normalize(sam_img)
                timg = T.gem_img_transform(sam_img) if want_gem else None
            else:
                norm = pin_upload(synth.imagenet_normalize(img), self.dev)
                timg = pin_upload(self.preprocess(img), self.dev) if want_gem else None
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with self._lock:
                self.decoded += 1
            slot.value = (ev, (sam_img, norm, timg))
            return slot.value[1]
        except BaseException as e:
            slot.error = e
            with self._lock:
                self._imgs.pop(iid, None)
            raise
        finally:
            slot.done.set()

    def load(self, i):
        import numpy as np
        from .gem import GEMWrapper, resize_antialias
        from .loader import pin_upload
        from .pipeline import RefBatch, Sentence
        from .tokenizer import tokenize
        args, dev = self.args, self.dev
        sam_img, image_norm, tensor_img = self._image(i)
        H, W = sam_img.shape[:2]
        rid = self.ds.ref_ids[i]
        ref = self.ds.refer.Refs[rid]
        annot, sentences = self.ds.target(i), self.ds.sentence_raws[i]
        strings, sents = [], []
        t = lambda a: pin_upload(a, dev)
        for sent_id, raw in zip(ref["sent_ids"], sentences):
            rec = self.parse.get(str(sent_id), {})
            # Hybridgl_main.py:131-141 tokenises the lower-cased sentence re-joined from spaCy's tokens; the record carries it
            raw = rec.get("sentence_for_spacy", raw)
            row = len(strings)
            others = list(rec.get("other_nouns", []))   # extract_nouns' phrases, bare (utils.py:82-98)
            strings += sentence_strings(raw, rec)
            attn = None
            if args.heatmap_dir and os.path.exists(os.path.join(args.heatmap_dir, f"{sent_id}.npy")):
                a = t(np.load(os.path.join(args.heatmap_dir, f"{sent_id}.npy")).astype(np.float32))
                if tuple(a.shape) != (H, W):     # Hybridgl_main.py:201: T.Resize((h, w), antialias=True)
                    a = resize_antialias(a[None], (H, W))[0]
                attn = a
            gem_row = None
            if attn is None and args.heatmap == "device":
                gem_row = len(strings)                                         # Hybridgl_main.py:200 gem_model(tensor_img, [noun_phrase])
                strings += GEMWrapper.prompts([rec.get("noun_phrase", raw)])
            elif attn is None:
                attn = torch.ones((H, W), dtype=torch.float32, device=dev)     # uniform: no spatial guidance
            sents.append(Sentence(row, row + 1, list(range(row + 2, row + 2 + len(others))), rec.get("dirflag", "none"),
                                  rec.get("relaflag", "none"), len(others), attn, gem_row=gem_row))
        with self._tk_lock:
            tokens = tokenize(strings, context_length=self.context_length, tokenizer=self.tk)   # raises on over-long text, as clip.tokenize
        placeholder = torch.zeros((1, H, W), dtype=torch.bool, device=dev)
        return RefBatch(sam_img, None, image_norm, placeholder,
                        torch.zeros((1, 4), dtype=torch.int64, device=dev), t(tokens), t(annot), sents, None,
                        int(ref["image_id"]), tensor_img=tensor_img,
                        token_len=int(tokens.argmax(axis=1).max()) + 1, index=i)


def real_refs(args, dev, splitBy, context_length, rank=0, world=1):
    """RefBatch per dataset item of this rank, in the loader's order, prepared on the calling thread (no prefetching)."""
    rr = RealRefs(args, dev, splitBy, context_length)
    for i in rr.jobs(rank, world):
        yield rr.load(i)


AMG_DEFAULTS = {   # Hybridgl_main.py:67-73 / Hybridgl_main_PhraseCut.py:56-62
    "refer": dict(points_per_side=8, pred_iou_thresh=0.7, stability_score_thresh=0.7, min_mask_region_area=800, crop_n_layers=0,
                  crop_n_points_downscale_factor=1, points_per_batch=64, group=16),
    "phrasecut": dict(points_per_side=64, pred_iou_thresh=0.86, stability_score_thresh=0.92, min_mask_region_area=100,
                      crop_n_layers=1, crop_n_points_downscale_factor=2, points_per_batch=1024, group=4),
}


def resolve_defaults(args):
    """fill the flags left at None with the configuration of the reference script that --dataset selects"""
    for k, v in AMG_DEFAULTS["phrasecut" if args.dataset == "phrasecut" else "refer"].items():
        if getattr(args, k, None) is None:
            setattr(args, k, v)
    if getattr(args, "split", None) is None:
        # Hybridgl_main_PhraseCut.py:42 evaluates PhraseCutDataset(split='test'); the REFER scripts their val split
        args.split = "test" if args.dataset == "phrasecut" else "val"
    if getattr(args, "group", None) is not None and args.group <= 1 and getattr(args, "proposal_cap", 0):
        # the ref-by-ref path (HybridGLPipeline.step) keeps every proposal: silently ignoring the cap would give other results
        # than the grouped loop whenever the cap binds
        raise SystemExit("--proposal_cap applies to the grouped loop only: use --group >= 2 with it (or drop the cap)")
    return args


class RealPhraseCut:
    """The dataset side of Hybridgl_main_PhraseCut.py:40-43,67-119: one item per IMAGE (hybridgl_amd/phrasecut_io.py), all its
    phrases scored against one proposal set and one hybrid forward, a ground truth per phrase.  `load(i)` -> RefBatch whose
    sentences carry their own targets (None for an image whose phrases are all filtered out: the loop skips it)."""

    def __init__(self, args, dev, context_length):
        import json
        import threading
        from .phrasecut_io import PhraseCutDataset
        from .tokenizer import SimpleTokenizer
        self.args, self.dev, self.context_length = args, dev, context_length
        self.ds = PhraseCutDataset(args.phrasecut_root, args.split, unseen_mode=args.unseen_mode, seen_mode=args.seen_mode)
        self.tk = SimpleTokenizer(args.bpe_vocab or None)
        self._tk_lock = threading.Lock()
        self.parse = json.load(open(args.parse_json)) if args.parse_json else {}
        self.n = len(self.ds) if args.max_refs <= 0 else min(len(self.ds), args.max_refs)
        self.decoded = 0

    def jobs(self, rank=0, world=1):
        from .dist import shard_indices
        return shard_indices(self.n, rank, world)

    def load(self, i):
        from . import transforms as T
        from .gem import GEMWrapper
        from .loader import pin_upload
        from .pipeline import RefBatch, Sentence
        from .tokenizer import tokenize
        item = self.ds[i]
        if item is None:
            return None
        dev = self.dev
        H, W = item["height"], item["width"]
        sam_img = pin_upload(item["sam_img"], dev)
        file_img = sam_img if item["file_img"] is None else pin_upload(item["file_img"], dev)
        image_norm = T.phrasecut_image_norm(file_img, H, W)             # dataset_phrasecut.py:49-51 + Hybridgl_main_PhraseCut.py:69-70
        want_gem = self.args.heatmap == "device"
        tensor_img = T.gem_img_transform(file_img) if want_gem else None    # :44-45 preprocessor(image): the file's own pixels
        self.decoded += 1
        strings, sents = [], []
        for j, phrase in enumerate(item["phrases"]):
            rec = self.parse.get(phrase, self.parse.get(phrase.lower(), {}))
            sentence = rec.get("sentence_for_spacy", phrase.lower())          # Hybridgl_main_PhraseCut.py:118-129
            row = len(strings)
            others = list(rec.get("other_nouns", []))
            strings += sentence_strings(sentence, rec)
            gem_row, attn = None, None
            if want_gem:
                gem_row = len(strings)
                strings += GEMWrapper.prompts([rec.get("noun_phrase", sentence)])      # :200
            else:
                attn = torch.ones((H, W), dtype=torch.float32, device=dev)
            gt = pin_upload(self.ds.gt_mask(item, j).astype("uint8"), dev)
            sents.append(Sentence(row, row + 1, list(range(row + 2, row + 2 + len(others))), rec.get("dirflag", "none"),
                                  rec.get("relaflag", "none"), len(others), attn, gt, gem_row=gem_row))
        with self._tk_lock:
            tokens = tokenize(strings, context_length=self.context_length, tokenizer=self.tk)
        placeholder = torch.zeros((1, H, W), dtype=torch.bool, device=dev)
        return RefBatch(sam_img, None, image_norm, placeholder, torch.zeros((1, 4), dtype=torch.int64, device=dev),
                        pin_upload(tokens, dev), sents[0].target, sents, None, int(item["image_id"]), tensor_img=tensor_img,
                        token_len=int(tokens.argmax(axis=1).max()) + 1, index=i)


def split_by(dataset):
    return "umd" if dataset == "refcocog" else "unc"          # Hybridgl_main.py:26-29


def build_models(args, dev):
    """(CLIPViTFM, SamAutomaticMaskGenerator | None, GEM model | None) as Hybridgl_main.py:36-38,47-48,66-74 builds them"""
    from .backbone import CLIPViTFM
    resolve_defaults(args)
    model = CLIPViTFM(model_name=args.clip_model, device=dev).eval()
    gen = None
    if args.sam or args.real:
        from .sam import SamAutomaticMaskGenerator, sam_model_registry
        sam = sam_model_registry[args.sam_model](device=dev)
        # Hybridgl_main.py:67-73
        gen = SamAutomaticMaskGenerator(sam, points_per_side=args.points_per_side, points_per_batch=args.points_per_batch,
                                        pred_iou_thresh=args.pred_iou_thresh, stability_score_thresh=args.stability_score_thresh,
                                        box_nms_thresh=args.box_nms_thresh, crop_n_layers=args.crop_n_layers,
                                        crop_n_points_downscale_factor=args.crop_n_points_downscale_factor,
                                        min_mask_region_area=args.min_mask_region_area)
    gem_model = None
    if args.heatmap == "device":
        from .gem import create_gem_model
        gem_model = create_gem_model(args.clip_model, clip=model)          # Hybridgl_main.py:36-38 (same checkpoint: shared weights)
    return model, gen, gem_model


def evaluate(args, model, gen, gem_model, dev, rank=0, world=1, dist=None):
    """The loop of Hybridgl_main.py:79-247 over this rank's share of the dataset: loader threads -> HybridGLPipeline.run ->
    one exchange of the metric rows.  Returns (metrics of the whole job, stats of this rank: refs, seconds of the loop,
    seconds the loop waited for the loader, host seconds spent preparing items, images decoded, image-cache hits,
    refs skipped)."""
    import time
    from .loader import Prefetcher
    from .pipeline import EmptyProposals, HybridGLPipeline, synthetic_ref
    from . import dist as D
    resolve_defaults(args)
    k_clamp = args.k_clamp if args.k_clamp != "auto" else ("persistent" if world == 1 else "per_ref")
    pipe = HybridGLPipeline(model, fusion_mode=args.fusion_mode, masking_block=getattr(args, "masking_block", 9), mask_generator=gen,
                            use_sam_masks=args.real, gem_model=gem_model, k_clamp=k_clamp)
    rr = None
    if args.real and args.dataset == "phrasecut":
        from .weights import CLIP_CONFIGS
        rr = RealPhraseCut(args, dev, CLIP_CONFIGS[args.clip_model]["context_length"])
        jobs, make = rr.jobs(rank, world), rr.load
    elif args.real:
        from .weights import CLIP_CONFIGS
        rr = RealRefs(args, dev, split_by(args.dataset), CLIP_CONFIGS[args.clip_model]["context_length"],
                      device_transforms=not getattr(args, "host_transforms", False))
        jobs, make = rr.jobs(rank, world), rr.load
    else:
        jobs = D.shard_indices(args.synthetic, rank, world)
        make = lambda i: synthetic_ref(i, dev, N=args.proposals, sam_img_size=1024 if gen else 0, gem=gem_model is not None,
                                       device_blur=True)[0]
    # Hybridgl_main.py:45,79: DataLoader(num_workers=4) feeding the loop; here loader threads feed the grouped loop
    loader = Prefetcher(jobs, make, workers=args.workers, depth=2 * args.group + 2, device=dev)
    cap = getattr(args, "proposal_cap", 0) or None
    if getattr(args, "prepare", False) and args.group > 1:
        # workspaces, allocator blocks and kernel instantiations of a full group, before the first item arrives
        pipe.prepare(group=args.group, H=640, W=640, proposals=cap or args.proposals)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if args.group <= 1:      # ref by ref on one stream (Hybridgl_main.py:79-230 as written)
        n = 0
        for ref in loader:
            if ref is None:
                continue
            try:
                pipe.step(ref)
                n += 1
            except EmptyProposals:
                pipe.skipped = getattr(pipe, "skipped", 0) + 1
    else:
        n = pipe.run(loader, group=args.group, proposal_cap=cap)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    m = pipe.metrics(dist)      # one all-gather of the metric rows; identical on every rank
    stats = {"refs": n, "seconds": dt, "seconds_job": D.max_over_ranks(dt, dist, dev), "loader_wait_s": loader.wait_s, "loader_make_s": loader.make_s,
             "images_decoded": rr.decoded if rr is not None else None, "image_cache_hits": pipe.cache_hits,
             "skipped": getattr(pipe, "skipped", 0), "groups": getattr(pipe, "groups_run", None),
             "workers": args.workers, "group": args.group}
    return m, stats


def main(args):
    from . import dist as D
    if not torch.cuda.is_available():
        # the reference falls back to the CPU (Hybridgl_main.py:30-34); this package is the MI355X hot path and nothing else:
        # libhybridgl.so's entries return HGL_ENODEVICE without a HIP device (tests/test_abi.py)
        raise SystemExit("hybridgl_amd.main: no HIP device is visible.  This package has no CPU implementation of the hot path (by "
                         "design: a CPU fallback would void every parity claim); the reference's CPU configuration (BASELINE "
                         "configs[0]) is covered by the oracle tests: python -m pytest tests -m 'not gpu'")
    rank, local_rank, world = D.env_rank()
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    cores = D.pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))   # launch + loader threads of a rank on its own cores
    D.size_host_threads(cores, args.workers)
    torch.cuda.set_device(dev)
    dist = D.init_process_group(os.environ.get("HYBRIDGL_DIST_BACKEND", "nccl"), dev) if world > 1 else None
    splitBy = split_by(args.dataset)
    model, gen, gem_model = build_models(args, dev)
    if rank == 0:
        print(f"fusion mode={args.fusion_mode}")
        if world > 1:
            print(f"ranks: {dist.get_world_size()} ({dist.get_backend()}), host cores per rank: {len(cores) or 'unpinned'}")
    m, stats = evaluate(args, model, gen, gem_model, dev, rank, world, dist)
    if stats["skipped"]:
        # the reference would fail on an image without proposals; count and go on
        print(f"{stats['skipped']} refs skipped: the proposal stage returned no mask")
    if dist is not None:
        dist.destroy_process_group()
    if rank != 0:
        return m
    if getattr(args, "stats_json", ""):
        import json
        stats["refs_per_s"] = stats["refs"] / stats["seconds"] if stats["seconds"] > 0 else 0.0
        stats["world"] = world
        stats["host_cores_per_rank"] = len(cores) if cores else len(os.sched_getaffinity(0))
        json.dump({"stats": stats, "metrics": m}, open(args.stats_json, "w"))
    pc = args.dataset == "phrasecut"
    text = (f"\n\n fusion_mode={args.fusion_mode} "
            + (f"\nDataset: PhraseCut / {args.split}" if pc else f"\nDataset: {args.dataset} / {args.split} / {splitBy}") +
            f"\nOverall IoU / mean IoU"
            f"\npure hybridgl: {m['oIoU']:.2f} / {m['mIoU']:.2f}"
            f"\nhybridgl w/ spatial guidance: {m['oIoU_final']:.2f} / {m['mIoU_final']:.2f}")
    os.makedirs(args.result_dir, exist_ok=True)                         # Hybridgl_main.py:233-248
    # Hybridgl_main.py:233-248 / Hybridgl_main_PhraseCut.py:224-238
    with open(os.path.join(args.result_dir, "result_log_PhraseCut.txt" if pc else f"result_log_{args.dataset}_{args.split}.txt"), "a") as f:
        f.write(text)
    print(text)
    return m


if __name__ == "__main__":
    main(default_argument_parser().parse_args())
