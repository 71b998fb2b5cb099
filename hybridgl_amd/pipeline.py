"""Per-ref evaluation step of Hybridgl_main.main (Hybridgl_main.py:79-230) on the GPU.

One `RefBatch` = one dataset item: an image, its mask proposals, its sentences (already
tokenised; the spaCy parse records are inputs -- an external package in the reference,
SURVEY.md 8c).  `HybridGLPipeline.step` runs, without any host synchronisation, the
reference's per-ref work:

    views   Hybridgl_main.py:93-125   hgl_synthesize_views
    hybrid  Hybridgl_main.py:128      hgl_clip_hybrid_forward
    text    Hybridgl_main.py:146-161  hgl_clip_encode_text (all strings of the ref in one batch)
    heat    Hybridgl_main.py:200-201  hgl_gem_image_features (once per image), hgl_gem_heatmap and
                                      hgl_resize_bilinear_aa (all sentences in one call) when a gem model is
                                      given; otherwise Sentence.imgattn is an input
    tail    Hybridgl_main.py:153-230  hgl_coherence_scores, hgl_score_sentence, hgl_iou_select

and accumulates cum_I/cum_U and the per-sentence IoUs on the device (Hybridgl_main.py:52-55,
240-247).  Metrics are read back once, at the end.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import ops, synth


@dataclass
class Sentence:
    """One referring expression after the (external) spaCy pre-processing of
    Hybridgl_main.py:131-146,156,176: token rows index into RefBatch.tokens."""
    sentence_row: int
    noun_phrase_row: int
    other_noun_rows: List[int]
    dirflag: str = "none"          # extract_dir_phrase (utils.py:100-133)
    relaflag: str = "none"         # extract_rela_word  (utils.py:206-238)
    n_nouns: int = 0               # len(nouns) of extract_nouns (Hybridgl_main.py:184)
    imgattn: Optional[torch.Tensor] = None  # [H,W] fp32: gem_model(...) resized to the image (:200-202)
    target: Optional[torch.Tensor] = None   # [H,W] per-phrase ground truth (Hybridgl_main_PhraseCut.py:117-119); else RefBatch.target
    gem_row: Optional[int] = None  # token row of "a photo of a {noun_phrase}." (the GEM prompt) when the heat-map is computed here


@dataclass
class RefBatch:
    sam_img: torch.Tensor      # [H,W,3] uint8   image['sam_img']
    blurred: Optional[torch.Tensor]  # [H,W,3] uint8   cv2.GaussianBlur(sam_img,(15,15),0)  (:99); None = computed in the step
    image_norm: torch.Tensor   # [3,H,W] fp32    image['image'] (ImageNet-normalised)
    masks: torch.Tensor        # [N,H,W] bool    SAM proposals (:86-87)
    boxes: torch.Tensor        # [N,4] int64     XYWH (:89-90)
    tokens: torch.Tensor       # [T,77] int32    every string of the ref
    target: torch.Tensor       # [H,W] bool/uint8 ground truth
    sentences: List[Sentence] = field(default_factory=list)
    sam_resized: Optional[torch.Tensor] = None  # [h,w,3] uint8: sam_img after ResizeLongestSide (PIL, host)
    image_id: Optional[int] = None  # COCO image id.  CONTRACT: items with the same image_id carry the same image tensors
                                    # (sam_img, image_norm, tensor_img); with a mask generator their proposals, hybrid and GEM
                                    # features are computed once per image (unit merging in run(), the image cache, step()'s
                                    # one-entry cache).  With GIVEN proposals run() never merges items.
    tensor_img: Optional[torch.Tensor] = None   # [3,448,448] fp32: image['tensor_img'] = gem.get_gem_img_transform()(img)
    token_len: Optional[int] = None   # 1 + the largest EOT position of `tokens` (host knowledge of the tokenizer): the text
                                      # encoder then computes only that prefix of the 77 positions (exact: causal mask)
    index: Optional[int] = None       # position in the loader's order (Hybridgl_main.py:45,79); under sharding the metric rows
                                      # of all ranks are put back into this order (hybridgl_amd/dist.py)
    ready: Optional[object] = None    # torch.cuda.Event recorded behind the uploads of this item (hybridgl_amd/loader.py: the
                                      # copies run on a loader thread's stream); consumers make their stream wait for it


class EmptyProposals(RuntimeError):
    """the proposal stage kept no mask for this image"""


def _adopt(ref, streams, produced=None):
    """An item whose tensors were produced on another stream than the ones that consume it -- uploaded on a loader thread's
    stream (RefBatch.ready), or built lazily on the caller's stream while the loop runs (`produced`: an event recorded
    there behind it): the consuming streams wait for that event, and the tensors are recorded on them so that the caching
    allocator does not hand their blocks to the next upload while a consumer still reads them."""
    ev = ref.ready if ref.ready is not None else produced
    if ev is None:
        return
    ts = [ref.sam_img, ref.blurred, ref.image_norm, ref.masks, ref.boxes, ref.tokens, ref.target, ref.sam_resized, ref.tensor_img]
    for s in ref.sentences:
        ts += [s.imgattn, s.target]
    for st in streams:
        st.wait_event(ev)
        for t in ts:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(st)


def _rows(text, rows):
    """rows of the text-feature matrix without a host->device index copy when they are consecutive."""
    if not rows:
        return None
    if list(rows) == list(range(rows[0], rows[0] + len(rows))):
        return text[rows[0]:rows[0] + len(rows)]
    return text.index_select(0, torch.as_tensor(rows, device=text.device)).contiguous()


def black_for(relaflag):
    """Hybridgl_main.py:211-216."""
    return 1.95 if relaflag == "big" else (1.5 if relaflag == "small" else 1.8)


_SIDE_STREAMS = {}


def _side_streams(device):
    """The three side streams of the loop (proposal stage, CLIP stage, text / GEM stage), ONE set per device for every
    pipeline object of the process: ops.workspace keeps one grow-only arena per (tag, stream), so pipelines that each
    brought their own streams would each leave ~30 GB of arenas behind (bench.py builds a dozen pipelines)."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = tuple(torch.cuda.Stream(dev) for _ in range(3))
    return _SIDE_STREAMS[key]


class HybridGLPipeline:
    def __init__(self, model, fusion_mode="G2L", masking_block=9, r=0.5, alpha=0.6, k1=3, k2=6, res=224,
                 mask_generator=None, use_sam_masks=False, fixed_proposals=None, cleanup_given_masks=False,
                 gem_model=None, k_clamp="persistent", image_cache=32):
        """mask_generator: a hybridgl_amd.sam.SamAutomaticMaskGenerator; when given, every step runs the
        SAM proposal stage (encoder, decoder, post-processing, NMS) on ref.sam_img first.
        use_sam_masks=False keeps ref.masks for the CLIP stage (fixed N; synthetic benchmark, where
        random SAM weights give an arbitrary number of proposals); True feeds the SAM proposals.
        k_clamp: "persistent" = the reference's quirk (Hybridgl_main.py:178-181: once an image yields fewer than k1 / k2
        proposals, k1 / k2 stay clamped for every LATER item of the process -- under sharding "later" means later on the
        same rank, so a run that meets such an image depends on the number of ranks); "per_ref" = the clamp applies
        to that item only (order- and sharding-independent; differs from the reference after such an image)."""
        if k_clamp not in ("persistent", "per_ref"):
            raise ValueError("k_clamp must be 'persistent' or 'per_ref'")
        # run(): proposals, hybrid features and GEM features of the last `image_cache` images with an image_id (the dataset
        # yields one item per REF, Hybridgl_main.py:79, and the refs of an image are not always neighbours: RefCOCO has 2.6
        # refs per image).  0 = off.  Entries are device tensors: ~25 MB per 640 x 480 image with 64 proposals.
        self.image_cache = int(image_cache)
        self._img_cache = {}
        self.cache_hits = 0
        self.k_clamp = k_clamp
        import os
        # the per-sentence tail as ONE hgl_score_ref call per ref (HYBRIDGL_FUSED_TAIL=0: the per-sentence launches)
        self.fused_tail = os.environ.get("HYBRIDGL_FUSED_TAIL", "1") != "0"
        # run(): the tails of ALL refs of a group in one hgl_score_group call (HYBRIDGL_GROUP_TAIL=0: one hgl_score_ref per ref)
        self.group_tail = os.environ.get("HYBRIDGL_GROUP_TAIL", "1") != "0"
        self.stagger = os.environ.get("HYBRIDGL_STAGGER", "decoder")   # which part of the next group's SAM stage the CLIP stage runs beside
        self._k0 = (k1, k2)
        self.model = model
        self.gem_model = gem_model              # hybridgl_amd.gem.GEMWrapper: heat-maps computed on the device
        self.mask_generator = mask_generator
        self.use_sam_masks = use_sam_masks
        self.fixed_proposals = fixed_proposals
        self.cleanup_given_masks = cleanup_given_masks
        self.fusion_mode = fusion_mode
        self.masking_block = masking_block
        self.r, self.alpha, self.k1, self.k2 = r, alpha, k1, k2  # Hybridgl_main.py:57-63
        self.res = res
        dev = model.device
        # metric accumulators stay on the device: [cum_I, cum_U, cum_I_final, cum_U_final]
        self.cum = torch.zeros(4, dtype=torch.int64, device=dev)
        self.iu_log = []  # per sentence (IU_pure, IU_final) device tensors
        self.iu_owner = []  # per sentence (dataset position of its ref, sentence number)
        self.idx_log = []   # per sentence: device int32 [2] = (index of the pure-CLIP winner, index with spatial guidance)
        self._n_refs = 0

    def _join_side_streams(self, cur, outs):
        """The caller's stream waits for both side streams; every tensor that leaves them is recorded on the caller's
        stream so that the caching allocator does not hand its block to later side-stream work while the caller reads it."""
        e1, e2 = torch.cuda.Event(), torch.cuda.Event()
        e1.record(self._s_sam)
        e2.record(self._s_clip)
        cur.wait_event(e1)
        cur.wait_event(e2)

        def rec(x):
            if isinstance(x, torch.Tensor):
                if x.is_cuda:
                    x.record_stream(cur)
            elif isinstance(x, (tuple, list)):
                for y in x:
                    rec(y)
        rec(outs)
        rec(getattr(self, "last_proposals", None))

    def step(self, ref: RefBatch, run_sam=True):
        """One dataset item; returns the device tensors of the last sentence (idx, scores).
        run_sam=False: the proposal stage of this ref is not part of this call (it ran earlier / on another stream)."""
        import dataclasses
        m = self.model
        _adopt(ref, (torch.cuda.current_stream(),))
        gen_here = self.mask_generator if run_sam else None
        # The text encoder (9 strings: small, latency-bound kernels) is independent of the image path: it runs on its
        # own stream underneath the SAM / CLIP image kernels and is joined before the scoring tail.
        cur = torch.cuda.current_stream()
        if not hasattr(self, "_s_text"):
            self._streams()
        ev_in, ev_text = torch.cuda.Event(), torch.cuda.Event()
        ev_in.record(cur)
        self._s_text.wait_event(ev_in)
        heat = None
        with torch.cuda.stream(self._s_text):
            text = m.model.encode_text(ref.tokens, seq_len=ref.token_len)
            gem_rows = [s.gem_row for s in ref.sentences if s.imgattn is None]
            if gem_rows:
                # Hybridgl_main.py:200-201: gem_model(tensor_img, [noun_phrase])[0] -> T.Resize((h, w), antialias=True).
                # The image tower depends on the image only (the reference re-runs it for every sentence); the
                # prompts' text features come out of the same text-encoder batch.
                from . import gem as G
                if self.gem_model is None or ref.tensor_img is None or any(r is None for r in gem_rows):
                    raise ValueError("a sentence without imgattn needs gem_model, RefBatch.tensor_img and Sentence.gem_row")
                if ref.image_id is not None and getattr(self, "_gem_cache_id", None) == ref.image_id:
                    gfeat = self._gem_cache_feat
                else:
                    gfeat = self.gem_model.image_features(ref.tensor_img)
                    if ref.image_id is not None:
                        self._gem_cache_id, self._gem_cache_feat = ref.image_id, gfeat
                maps = self.gem_model.heatmap(gfeat, _rows(text, gem_rows), ref.tensor_img.shape[-1])
                heat = G.resize_antialias(maps, ref.sam_img.shape[:2])
                heat.record_stream(cur)
            ev_text.record(self._s_text)
        text.record_stream(cur)
        # Per-image caching: the dataset yields one item per REF and the same image backs
        # several consecutive refs (Hybridgl_main.py:79); proposals, views and hybrid features depend on the
        # image only, so they are computed once per image.  Results are identical.
        from_gen = self.mask_generator is not None and self.use_sam_masks     # given proposals belong to the ITEM, not to the image
        if from_gen and ref.image_id is not None and getattr(self, "_cache_id", None) == ref.image_id:
            hybrid = self._cache_hybrid
            ref = dataclasses.replace(self._cache_ref, tokens=ref.tokens, token_len=ref.token_len, sentences=ref.sentences,
                                      target=ref.target, index=ref.index)
        else:
            if gen_here is not None:
                # Hybridgl_main.py:85 mask_generator.generate(sam_img), kept on the device
                if self.use_sam_masks and getattr(gen_here, "crop_n_layers", 0) > 0:
                    # PhraseCut configuration (Hybridgl_main_PhraseCut.py:56-62): crop layers, cross-crop NMS
                    prop = gen_here.generate_device_crops(ref.sam_img)[:4]
                elif self.use_sam_masks or self.fixed_proposals is not None:
                    prop = gen_here.generate_device(ref.sam_img, resized=ref.sam_resized,
                                                    fixed_n=self.fixed_proposals)
                else:  # proposal kernels only, nothing read back
                    prop = gen_here.propose(ref.sam_img, resized=ref.sam_resized)
                if self.use_sam_masks:
                    if prop[0].shape[0] == 0:
                        raise EmptyProposals("no proposals")
                    ref = dataclasses.replace(ref, masks=prop[0].view(torch.bool) if prop[0].dtype == torch.uint8 else prop[0],
                                              boxes=prop[1].contiguous())
                self.last_proposals = prop
                if self.cleanup_given_masks and not self.use_sam_masks:
                    # synthetic benchmark: the small-region clean-up runs on the proposal-shaped seeded masks
                    # (random-weight SAM logits are pixel noise, which is not what the clean-up sees in practice)
                    cm, _ = gen_here.cleanup_fixed(ref.masks.view(torch.uint8))
                    ref = dataclasses.replace(ref, masks=cm.view(torch.bool))
            blurred = ref.blurred if ref.blurred is not None else ops.gaussian_blur_u8(ref.sam_img, 15)   # :99
            local, glob = ops.synthesize_views(ref.sam_img, blurred, ref.image_norm, ref.masks, self.res)
            hybrid = m(local, glob, ref.masks, masking_block=self.masking_block, fusion_mode=self.fusion_mode)
            if from_gen and ref.image_id is not None:
                self._cache_id, self._cache_ref, self._cache_hybrid = ref.image_id, ref, hybrid
        cur.wait_event(ev_text)
        return hybrid, text, self._score_ref(ref, hybrid, text, heat)

    _DEFERRED = object()

    def _log_tail(self, ref_index, n, idx, iu):
        for j in range(n):
            self.iu_log.append((iu[j, 0:2], iu[j, 2:4]))
            self.iu_owner.append((ref_index, j))
            self.idx_log.append(idx[j])

    def _flush_tails(self, defer):
        """the deferred tails of a group's refs in one hgl_score_group call; returns each ref's last-sentence tensors, in order"""
        if not defer:
            return []
        outs = ops.score_group(defer, self.model.model._logit_scale_exp, self.r, self.alpha, cum=self.cum, want_scores=True)
        last = []
        for q, (idx, iu, sc, sn, gm) in zip(defer, outs):
            self._log_tail(q["ref_index"], len(q["sentences"]), idx, iu)
            last.append((idx[-1], sc[-1], sn[-1], gm[-1]))
        defer.clear()
        return last

    def _score_ref(self, ref, hybrid, text, heat, defer=None):
        """the per-sentence tail of Hybridgl_main.py:153-230 for one ref; returns the tensors of its last sentence.  defer (a
        list): a ref whose tail the fused kernels serve is appended to it instead (for _flush_tails) and _DEFERRED returned"""
        m = self.model
        # the k1/k2 clamp of Hybridgl_main.py:178-181 persists across refs in the reference
        if self.k_clamp == "per_ref":
            self.k1, self.k2 = self._k0
        self.k1 = min(self.k1, hybrid.shape[0])
        self.k2 = min(self.k2, hybrid.shape[0])
        ref_index = ref.index if ref.index is not None else self._n_refs
        self._n_refs += 1
        if not ref.sentences:        # an item without a sentence scores nothing (the reference's inner loop does not run)
            return None
        heats = iter(heat) if heat is not None else None
        attn = [s.imgattn if s.imgattn is not None else next(heats) for s in ref.sentences]
        fused = self.fused_tail and text.is_contiguous() and all(
            list(s.other_noun_rows) == list(range(s.other_noun_rows[0], s.other_noun_rows[0] + len(s.other_noun_rows)))
            for s in ref.sentences if s.other_noun_rows)
        if fused:
            # hgl_score_ref: every mask byte read once for all sentences' heat-maps, one scoring workgroup per sentence,
            # both IoUs of every sentence and the accumulators of Hybridgl_main.py:52-55 in the same four launches
            recs = [dict(sentence_row=s.sentence_row, noun_phrase_row=s.noun_phrase_row,
                         other_row0=s.other_noun_rows[0] if s.other_noun_rows else 0, n_other=len(s.other_noun_rows),
                         dirflag=s.dirflag, relaword=s.relaflag, has_other_nouns=s.n_nouns != 0, black=black_for(s.relaflag),
                         imgattn=a if a.is_contiguous() else a.contiguous(), target=s.target if s.target is not None else ref.target)
                    for s, a in zip(ref.sentences, attn)]
            if defer is not None and 1 <= len(recs) <= ops.SCORE_GROUP_MAX_SENTENCES:
                defer.append(dict(hybrid=hybrid, text=text, boxes=ref.boxes, masks=ref.masks, sentences=recs, k1=self.k1, k2=self.k2,
                                  ref_index=ref_index))
                return self._DEFERRED
            idx, iu, sc, sn, gm = ops.score_ref(hybrid, text, ref.boxes, ref.masks, recs, m.model._logit_scale_exp, self.r, self.k1,
                                                self.k2, self.alpha, cum=self.cum, want_scores=True)
            self._log_tail(ref_index, len(recs), idx, iu)
            return (idx[-1], sc[-1], sn[-1], gm[-1]) if recs else None
        last = None
        for sent_no, (s, imgattn) in enumerate(zip(ref.sentences, attn)):
            gem = ops.coherence_scores(imgattn, ref.masks, s.dirflag, black_for(s.relaflag))
            others = _rows(text, s.other_noun_rows)
            idx, sc, sn = ops.score_sentence(hybrid, text[s.sentence_row], text[s.noun_phrase_row], others,
                                             ref.boxes, gem, m.model._logit_scale_exp, self.r, self.k1,
                                             self.k2, self.alpha, s.relaflag, s.n_nouns != 0)
            tgt = s.target if s.target is not None else ref.target
            iu0 = ops.iou_select(ref.masks, idx, 0, tgt)
            iu1 = ops.iou_select(ref.masks, idx, 1, tgt)
            self.cum[0:2] += iu0
            self.cum[2:4] += iu1
            self.iu_log.append((iu0, iu1))
            self.iu_owner.append((ref_index, sent_no))
            self.idx_log.append(idx)
            last = (idx, sc, sn, gem)
        return last

    # ---- the evaluation loop at the grouped rate ----------------------------------------------------------------------
    def _streams(self):
        if not hasattr(self, "_s_sam"):
            self._s_sam, self._s_clip, self._s_text = _side_streams(self.model.device)
            self._ev = torch.cuda.Event()
        return self._s_sam, self._s_clip

    @staticmethod
    def _units(loader, group, merge=True):
        """Groups of up to `group` IMAGE UNITS in the loader's order; a unit = the consecutive items that share an
        image_id (the dataset yields one item per ref, Hybridgl_main.py:79; proposals, views and hybrid features depend on
        the image only).  Items without an image_id are units of their own; merge=False makes every item its own unit
        (proposals GIVEN per item: two items of one image may carry different masks / boxes, and a unit uses its first
        item's)."""
        units, cur = [], None
        for ref in loader:
            if ref is None:      # an item the dataset filtered out entirely (PhraseCut seen / unseen modes)
                continue
            if merge and cur is not None and ref.image_id is not None and ref.image_id == cur[0].image_id:
                cur.append(ref)
                continue
            if cur is not None:
                units.append(cur)
                if len(units) == group:
                    yield units
                    units = []
            cur = [ref]
        if cur is not None:
            units.append(cur)
        if units:
            yield units

    @staticmethod
    def balanced_group(total, group):
        """The group size run() uses for `total` items at most `group` at a time: the same number of groups, equally
        full (20 refs, group 16 -> 10 + 10 instead of 16 + 4: the proposal stage of the second group then has a CLIP stage
        of its own size beside it, and no stage meets a shape it has not seen)."""
        total, group = int(total), max(int(group), 1)
        if total <= group:
            return max(total, 1)
        n_groups = -(-total // group)
        return -(-total // n_groups)

    def prepare(self, group=16, H=640, W=640, proposals=64, n_sent=3, tail=None, serial=False, slack=1 << 30):
        """Everything run() would otherwise do on first use, done once, up front: every grow-only workspace (ops.workspace,
        one arena per stage and stream) and every block of the caching allocator sized for a full group of `group` images
        of H x W with up to `proposals` masks each AND for a ragged last group of `tail` images (default: a quarter of a
        group); every kernel instantiation, LDS reservation and per-shape table of those two group sizes touched.  After
        it, the first run() costs what the hundredth does, whatever the length of the caller's warm-up.  It runs the
        product loop itself on seeded synthetic items (synthetic_ref), so nothing can be missed by construction; metric
        rows, logs and the image cache are restored afterwards.  ~1 s per call at the benchmark geometry."""
        dev = self.model.device
        gen = self.mask_generator
        tail = max(1, group // 4) if tail is None else int(tail)
        use_gem = self.gem_model is not None
        items = [synthetic_ref(90000 + j, dev, N=proposals, H=H, W=W, n_sent=n_sent, sam_img_size=1024 if gen is not None else 0,
                               gem=use_gem, device_blur=True)[0] for j in range(min(group, 4))]
        keep = (self.cum.clone(), list(self.iu_log), list(self.iu_owner), list(self.idx_log), self._n_refs,
                getattr(self, "skipped", 0), getattr(self, "groups_run", 0), dict(self._img_cache), self.cache_hits,
                (self.k1, self.k2), getattr(self, "group_marks", None), getattr(self, "stage_marks", None))
        self.group_marks = self.stage_marks = None
        cap = proposals if (gen is not None and self.use_sam_masks) else None
        try:
            # two full groups (the second one's proposal stage runs beside the first one's CLIP stage: both streams' arenas
            # reach their steady size), then a ragged group
            for n in (2 * group, tail):
                self.run((items[i % len(items)] for i in range(n)), group=group, proposal_cap=cap, serial=serial)
            if cap is not None:
                # the generator decides how many masks the synthetic images yield; the CLIP stage is sized for the full
                # `proposals` per image by one more pass over the items' own (seeded) masks
                self.use_sam_masks = False
                try:
                    for n in (group, tail):
                        self.run((items[i % len(items)] for i in range(n)), group=group, serial=serial)
                finally:
                    self.use_sam_masks = True
            torch.cuda.synchronize(dev)
            # The caching allocator keeps one pool per stream and the loop's tensors differ a little from group to group (the
            # number of proposals is data): leave `slack` bytes of cached, splittable blocks behind on every stream of the loop
            # so that a request a few MB above anything seen here is served from the pool, not by hipMalloc in the middle of a
            # group (sized for 288 GB of HBM: 4 x 1 GiB is 1.4 % of the device)
            if slack > 0:
                streams = [torch.cuda.current_stream(dev)] + ([] if serial else list(self._streams()) + [self._s_text])
                for st in streams:
                    with torch.cuda.stream(st):
                        blocks = [torch.empty(int(slack) // 2, dtype=torch.uint8, device=dev) for _ in range(2)]
                        del blocks
            torch.cuda.synchronize(dev)
        finally:
            self.cum.copy_(keep[0])
            self.iu_log[:], self.iu_owner[:], self.idx_log[:] = keep[1], keep[2], keep[3]
            self._n_refs, self.skipped, self.groups_run = keep[4], keep[5], keep[6]
            self._img_cache, self.cache_hits = keep[7], keep[8]
            self.k1, self.k2 = keep[9]
            self.group_marks, self.stage_marks = keep[10], keep[11]
        return self

    def run(self, loader, group=16, proposal_cap=None, collect=False, serial=False, total=None):
        """The loop of Hybridgl_main.py:79-230 over a whole loader, taken `group` images at a time on two streams:

            SAM stream   group g+1: ONE encoder pass over its images, per image decoder + post-processing + NMS, small-region
                         clean-up + second NMS (SamAutomaticMaskGenerator.group_begin / group_cleanup / group_finish: the
                         survivor counts of the whole group are read back in TWO device->host copies, not 2 per image)
            CLIP stream  group g: one text-encoder batch over the strings of all its refs, one GEM tower pass over its
                         images, one hybrid forward over the ACTUAL proposals of all its images (ragged counts, images of
                         different sizes), then the per-sentence tail of every ref in the loader's order

        The CLIP stage of group g is enqueued before the host waits for the counts of group g+1, so the device always has
        work.  `loader` yields RefBatch items (device tensors; hybridgl_amd.loader.Prefetcher prepares them on background
        threads, Hybridgl_main.py:45).  Results per ref are those of step() -- every mask row, string and image is
        independent -- up to the summation order of the SAM encoder's split-K, which only a single-image pass uses.
        Without a mask generator (or with run_sam=False semantics: use_sam_masks=False and no generator) the proposals are
        RefBatch.masks / boxes.  Images for which the generator keeps no mask are skipped and counted (self.skipped; the
        reference would fail on them).  Returns the number of refs scored; collect=True also keeps step()'s per-ref
        return values in self.collected.  serial=True: the same work with every stage on the CURRENT stream, back to back
        (per-kernel timing with events on one stream).  total: the number of items the loader will yield, when known -- the
        groups are then equally full (balanced_group) instead of full groups and a remainder."""
        gen = self.mask_generator
        if total is not None:
            group = self.balanced_group(total, group)
        cur = torch.cuda.current_stream()
        self._serial = bool(serial)
        if serial:
            s_sam = s_clip = cur
        else:
            s_sam, s_clip = self._streams()
            self._ev.record(cur)
            s_sam.wait_event(self._ev)
            s_clip.wait_event(self._ev)
        self.collected = [] if collect else None
        self.skipped = getattr(self, "skipped", 0)
        done = 0
        pending = None
        self.groups_run = getattr(self, "groups_run", 0)
        # items of one image are merged into a unit only when the proposals come from the generator (then they are a function
        # of the image); with given proposals every item keeps its own masks / boxes, as step() does
        for units in self._units(loader, group, merge=gen is not None and self.use_sam_masks):
            self.groups_run += 1
            if getattr(self, "group_marks", None) is not None:      # host-side stamp of every group boundary (tools/first_use.py)
                import time
                self.group_marks.append(time.perf_counter())
            produced = None
            if not serial:      # items a lazy loader built on the caller's stream just now
                produced = torch.cuda.Event()
                produced.record(cur)
            for u in units:
                for r in u:
                    _adopt(r, (s_sam, s_clip), produced)
            state = None
            ev_enc = None
            # images seen lately (or in the group whose CLIP stage is about to be enqueued: it files them before this
            # group's CLIP stage looks): no proposal stage, no hybrid forward for them
            cached = [False] * len(units)
            held = [None] * len(units)      # the cache entry itself, pinned at look-up (a later put may evict it from the dict)
            if self.image_cache > 0 and gen is not None and self.use_sam_masks:
                self._cache_cap = max(self.image_cache, 2 * group)
                in_pending = {u[0].image_id for u in pending[0]} if pending is not None else set()
                for i, u in enumerate(units):
                    iid = u[0].image_id
                    if iid is None:
                        continue
                    if iid in self._img_cache:
                        cached[i], held[i] = True, ("entry", self._img_cache[iid])
                    elif iid in in_pending:
                        cached[i], held[i] = True, ("late",)
            fresh = [i for i, c in enumerate(cached) if not c]
            if gen is not None and fresh:
                with torch.cuda.stream(s_sam):
                    self._stage_mark("sam_begin")
                    imgs = [units[i][0].sam_img for i in fresh]
                    if self.use_sam_masks:
                        if getattr(gen, "crop_n_layers", 0) > 0:
                            # PhraseCut configuration (crop layers): the same begin / finish split -- every crop of every image
                            # of the group is enqueued here without a wait, the three count read-backs come after the CLIP
                            # stage of the previous group has been enqueued
                            if self.stagger == "decoder" and not serial:
                                ev_enc = torch.cuda.Event()
                            state = ("crops", gen.crops_begin(imgs, ev_enc))
                        else:
                            # stagger = "decoder": the CLIP stage of the previous group starts when THIS group's encoder pass
                            # is through, so its large GEMMs run beside the latency-bound rest of the proposal stage (decoder,
                            # post-processing, NMS, clean-up) instead of beside the encoder's equally matrix-bound GEMMs
                            if self.stagger == "decoder" and not serial:
                                ev_enc = torch.cuda.Event()
                            state = ("group", gen.group_begin(imgs, proposal_cap, ev_enc))
                    else:    # proposal kernels only; their output is not consumed (synthetic benchmark, seeded masks)
                        self.last_proposals = gen.propose_batch(imgs)[-1]
            if pending is not None:
                with torch.cuda.stream(s_clip):
                    if ev_enc is not None:
                        s_clip.wait_event(ev_enc)
                    done += self._clip_group(*pending)
            props, ready = None, None
            if state is not None:
                with torch.cuda.stream(s_sam):
                    if state[0] == "group":
                        stc = gen.group_cleanup(state[1])
                        if stc.overflow:
                            # fail at THIS group, not in metrics() after the whole dataset: the counter rode on the group's
                            # count read-back (everything the device had finished by then, on any stream).  The counters are
                            # process-global: cleared here, or every later run() of the process -- the f32 rerun this message
                            # recommends included -- would trip over the same count at its first group
                            ops.split_overflow_count(reset=True)
                            raise ops.SplitOverflow(
                                f"activations exceeded the fp16 range (|x| > 65504) in f16x3 mode by group {self.groups_run} of the "
                                f"loop ({stc.overflow} GPU threads saw one; refs up to dataset position {units[-1][-1].index}): the "
                                "results from the previous group on contain inf / NaN; rerun with HYBRIDGL_PRECISION=f32 (or precision='f32')")
                        got = [p[:2] for p in gen.group_finish(stc)]
                    else:
                        stc = gen.crops_mid(state[1])
                        if stc.overflow:
                            ops.split_overflow_count(reset=True)
                            raise ops.SplitOverflow(
                                f"activations exceeded the fp16 range (|x| > 65504) in f16x3 mode by group {self.groups_run} of the "
                                f"loop ({stc.overflow} GPU threads saw one): rerun with HYBRIDGL_PRECISION=f32 (or precision='f32')")
                        got = [tuple(t[:proposal_cap] if proposal_cap is not None else t for t in p[:2])
                               for p in gen.crops_finish(gen.crops_post(stc))]
                    ready = torch.cuda.Event()
                    ready.record(s_sam)
                    self._stage_mark("sam_end")
                props = [None] * len(units)          # None = take it from the image cache
                for i, pr in zip(fresh, got):
                    props[i] = pr
            elif any(cached):
                props = [None] * len(units)
            pending = (units, props, ready, held)
        if pending is not None:
            with torch.cuda.stream(s_clip):
                done += self._clip_group(*pending)
        if not serial:
            self._join_side_streams(cur, self.collected)
        return done

    def _clip_group(self, units, props, ready, held=None):
        """CLIP + scoring stage of one group on the current stream; props[i] = (masks u8 [n,H,W], boxes XYWH) of unit i from
        the proposal stage (None: the items' own masks / boxes).  Returns the number of refs scored."""
        import dataclasses
        m = self.model
        cur = torch.cuda.current_stream()
        if ready is not None:
            cur.wait_event(ready)
        self._stage_mark("clip_begin")
        gen = self.mask_generator
        live = []     # [refs of the unit, masks bool [n,H,W], boxes, hybrid features or None, GEM features or None]
        for i, refs in enumerate(units):
            if props is not None and props[i] is None:      # an image of the cache
                hit = held[i][1] if held[i][0] == "entry" else self._img_cache.get(refs[0].image_id)
                if hit is None:                              # filed as "no proposals"
                    self.skipped += len(refs)
                    continue
                self.cache_hits += 1
                live.append([refs, hit[0], hit[1], hit[2], hit[3]])
            elif props is not None:
                mk, bx = props[i]
                if mk.shape[0] == 0:
                    self.skipped += len(refs)
                    if self.image_cache > 0 and refs[0].image_id is not None:
                        self._cache_put(refs[0].image_id, None)
                    continue
                mk.record_stream(cur)
                bx.record_stream(cur)
                live.append([refs, mk.view(torch.bool) if mk.dtype == torch.uint8 else mk, bx.contiguous(), None, None])
            else:
                mk = refs[0].masks
                if self.cleanup_given_masks and gen is not None:
                    mk = gen.cleanup_fixed(mk.view(torch.uint8))[0].view(torch.bool)
                live.append([refs, mk, refs[0].boxes, None, None])
        if not live:
            return 0
        if not hasattr(self, "_s_text"):
            self._streams()
        s_text = cur if getattr(self, "_serial", False) else self._s_text
        ev_in, ev_text = torch.cuda.Event(), torch.cuda.Event()
        ev_in.record(cur)
        s_text.wait_event(ev_in)
        all_refs = [r for u in live for r in u[0]]
        offs = np.cumsum([0] + [r.tokens.shape[0] for r in all_refs])
        heats = []
        with torch.cuda.stream(s_text):
            lens = [r.token_len for r in all_refs]
            text_all = m.model.encode_text(torch.cat([r.tokens for r in all_refs], dim=0) if len(all_refs) > 1 else all_refs[0].tokens,
                                           seq_len=None if any(v is None for v in lens) else max(lens))
            # GEM image tower once per IMAGE that has a sentence without a given heat-map
            wants = [i for i, u in enumerate(live) if any(s.imgattn is None for r in u[0] for s in r.sentences)]
            gfeats = {i: live[i][4] for i in wants if live[i][4] is not None}
            need = [i for i in wants if i not in gfeats]
            if need:
                if self.gem_model is None or any(live[i][0][0].tensor_img is None for i in need):
                    raise ValueError("a sentence without imgattn needs gem_model, RefBatch.tensor_img and Sentence.gem_row")
                timgs = [live[i][0][0].tensor_img for i in need]
                if len(need) > 1 and all(t.shape == timgs[0].shape for t in timgs):
                    fb = self.gem_model.image_features_batch(torch.stack(timgs, dim=0))
                    gfeats.update({i: fb[j] for j, i in enumerate(need)})
                else:
                    gfeats.update({i: self.gem_model.image_features(t) for i, t in zip(need, timgs)})
                for i in need:
                    gfeats[i].record_stream(cur)
                    live[i][4] = gfeats[i]
            k = 0
            for i, u in enumerate(live):
                for ref in u[0]:
                    text = text_all[offs[k]:offs[k + 1]]
                    k += 1
                    gem_rows = [s.gem_row for s in ref.sentences if s.imgattn is None]
                    heat = None
                    if gem_rows:
                        from . import gem as G
                        if any(r is None for r in gem_rows):
                            raise ValueError("a sentence without imgattn needs gem_model, RefBatch.tensor_img and Sentence.gem_row")
                        maps = self.gem_model.heatmap(gfeats[i], _rows(text, gem_rows), ref.tensor_img.shape[-1])
                        heat = G.resize_antialias(maps, ref.sam_img.shape[:2])     # Hybridgl_main.py:201
                        heat.record_stream(cur)
                    heats.append(heat)
            ev_text.record(s_text)
        text_all.record_stream(cur)
        todo = [i for i, u in enumerate(live) if u[3] is None]       # images whose hybrid features are not cached
        if todo:
            ns = [live[i][1].shape[0] for i in todo]
            moff = np.cumsum([0] + ns)
            dev = live[todo[0]][1].device
            local = torch.empty((int(moff[-1]), 3, self.res, self.res), dtype=torch.float32, device=dev)
            glob = torch.empty_like(local)
            for j, i in enumerate(todo):
                r0, mk = live[i][0][0], live[i][1]
                blurred = r0.blurred if r0.blurred is not None else ops.gaussian_blur_u8(r0.sam_img, 15)   # :99
                ops.synthesize_views(r0.sam_img, blurred, r0.image_norm, mk, self.res,
                                     out=(local[moff[j]:moff[j + 1]], glob[moff[j]:moff[j + 1]]))
            hybrid_all = m(local, glob, [live[i][1] for i in todo], masking_block=self.masking_block, fusion_mode=self.fusion_mode)
            for j, i in enumerate(todo):
                live[i][3] = hybrid_all[moff[j]:moff[j + 1]]
        cur.wait_event(ev_text)
        k = 0
        defer = [] if self.group_tail else None      # the refs whose tails go into ONE hgl_score_group call at the end
        waiting = []                                 # their (hybrid, text) for `collected`, in order
        for refs, mk, bx, hybrid, gfeat in live:
            if self.image_cache > 0 and props is not None and refs[0].image_id is not None:
                self._cache_put(refs[0].image_id, (mk, bx, hybrid, gfeat))
            for ref in refs:
                text = text_all[offs[k]:offs[k + 1]]
                out = self._score_ref(dataclasses.replace(ref, masks=mk, boxes=bx), hybrid, text, heats[k], defer)
                k += 1
                if out is self._DEFERRED:
                    waiting.append((hybrid, text))
                    continue
                if defer:      # a ref the group call does not serve: the deferred ones before it are logged first (order)
                    for (h0, t0), o0 in zip(waiting, self._flush_tails(defer)):
                        if self.collected is not None:
                            self.collected.append((h0, t0, o0))
                    waiting = []
                if self.collected is not None:
                    self.collected.append((hybrid, text, out))
        if defer:
            for (h0, t0), o0 in zip(waiting, self._flush_tails(defer)):
                if self.collected is not None:
                    self.collected.append((h0, t0, o0))
        self._stage_mark("clip_end")
        return len(all_refs)

    def _stage_mark(self, label, stream=None):
        """opt-in timeline of the loop (self.stage_marks = [] before run(); bench.py's timed_region, tools/first_use.py): a
        timing event on `stream` (default: the current one) plus the host time of the call"""
        marks = getattr(self, "stage_marks", None)
        if marks is None:
            return
        import time
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream if stream is not None else torch.cuda.current_stream())
        marks.append((label, self.groups_run, ev, time.perf_counter()))

    def stage_timeline(self, base_event, t0):
        """[(label, group number, device ms since base_event, host ms since t0)] of the marks; call after a synchronize"""
        return [(lb, g, round(base_event.elapsed_time(ev), 2), round((th - t0) * 1e3, 2)) for lb, g, ev, th in (self.stage_marks or [])]

    def _cache_put(self, image_id, entry):
        """most recently used last; the oldest entry leaves when the cache is full"""
        self._img_cache.pop(image_id, None)
        self._img_cache[image_id] = entry
        while len(self._img_cache) > getattr(self, "_cache_cap", self.image_cache):
            self._img_cache.pop(next(iter(self._img_cache)))

    def partial_rows(self):
        """This process's per-sentence rows [n, 6] int64 = (dataset position, sentence, I, U, I_final, U_final)
        (hybridgl_amd.dist.ROW_FIELDS): the unit that ranks exchange.  One device->host copy."""
        if not self.iu_log:
            return np.zeros((0, 6), dtype=np.int64)
        iu = torch.stack([torch.cat([a.reshape(2), b.reshape(2)]) for a, b in self.iu_log]).cpu().numpy().astype(np.int64)
        return np.concatenate([np.asarray(self.iu_owner, dtype=np.int64).reshape(-1, 2), iu], axis=1)

    def winning_indices(self):
        """[n_sentences, 2] int64: per sentence the proposal indices (pure CLIP, with spatial guidance) into that ref's
        proposal list, in the order the sentences were scored.  One device->host copy."""
        if not self.idx_log:
            return np.zeros((0, 2), dtype=np.int64)
        return torch.stack(self.idx_log).cpu().numpy().astype(np.int64)

    def metrics(self, dist=None):
        """Hybridgl_main.py:240-247: overall IoU and mean IoU, pure and with spatial guidance; with an initialised
        torch.distributed (`dist`) the rows of all ranks are gathered first (the job's metrics, on every rank)."""
        from . import dist as D
        rows = self.partial_rows()
        overflow = 0
        if getattr(self.model.model, "precision", "f32") == "f16x3":
            overflow = ops.split_overflow_count(reset=True)
        # exchange first, raise afterwards and on EVERY rank: a rank that raised before the all-gather would leave the others
        # waiting in the collective
        m = D.gather_metrics(rows, dist, self.model.device)
        overflow = int(D.max_over_ranks(float(overflow), dist, self.model.device))
        if overflow:     # an activation beyond the fp16 range voids the run: raise, do not report
            from ._lib import HybridGLError
            raise HybridGLError(f"activations exceeded the fp16 range (|x| > 65504) in f16x3 mode ({overflow} GPU threads saw one on "
                                "some rank): the results contain inf / NaN; rerun with HYBRIDGL_PRECISION=f32 (or precision='f32')")
        return m


def synthetic_ref(i, device, N=64, H=640, W=640, n_sent=3, context=77, vocab=49408, sam_img_size=0, gem=False, gem_size=448,
                  device_blur=False):
    """The benchmark item of SURVEY.md 8d: 640x640 image, 64 proposals, 3 queries, each with a
    sentence, a noun phrase and one other noun (9 token rows).  Returns (RefBatch, numpy dict).
    gem=True: the heat-map is NOT an input; the ref carries tensor_img [3,gem_size,gem_size] and one more token
    row per sentence (the GEM prompt), and the pipeline computes the heat-maps on the device.
    device_blur=True: RefBatch.blurred is None and the step runs cv2.GaussianBlur's fixed-point filter itself."""
    img = synth.synth_image(H, W, 1000 + i)
    blur = synth.box_blur_u8(img)
    norm = synth.imagenet_normalize(img)
    masks = synth.synth_masks(N, H, W, 2000 + i)
    boxes = synth.boxes_from_masks(masks)
    tokens = synth.synth_tokens(3 * n_sent, context, vocab, 3000 + i)
    tensor_img = None
    if gem:
        from .gem import get_gem_img_transform
        tokens = np.concatenate([tokens, synth.synth_tokens(n_sent, context, vocab, 5000 + i)], axis=0)
        tensor_img = get_gem_img_transform(gem_size)(img).numpy()
    gt = masks[(7 * i) % N]
    sents, attn_np = [], []
    for j in range(n_sent):
        dirflag, relaflag, n_nouns = synth.PARSE_RECORDS[j % len(synth.PARSE_RECORDS)]
        attn = synth.synth_heatmap(H, W, 4000 + 10 * i + j)
        attn_np.append(attn)
        sents.append(Sentence(3 * j, 3 * j + 1, [3 * j + 2], dirflag, relaflag, n_nouns,
                              None if gem else torch.from_numpy(attn).to(device), gem_row=3 * n_sent + j if gem else None))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    resized = None   # ResizeLongestSide runs on the device (hgl_resize_pil_bilinear), inside the step
    ref = RefBatch(t(img), None if device_blur else t(blur), t(norm), t(masks), t(boxes), t(tokens), t(gt), sents, resized,
                   tensor_img=t(tensor_img) if gem else None, token_len=int(tokens.argmax(axis=1).max()) + 1, index=i)
    host = dict(img=img, blur=blur, norm=norm, masks=masks, boxes=boxes, tokens=tokens, gt=gt, attn=attn_np,
                tensor_img=tensor_img)
    return ref, host
