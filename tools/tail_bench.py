#!/usr/bin/env python3
"""Timing of the scoring tail on the benchmark's shape (64 proposals, 640 x 640, 3 sentences, E = 512): hgl_score_ref (one call per
ref) against the per-sentence launches; HIP events around 20 calls each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hybridgl_amd import ops
from hybridgl_amd.pipeline import synthetic_ref

dev = torch.device("cuda:0")
ref, _ = synthetic_ref(0, dev, N=64)
torch.manual_seed(0)
hybrid = torch.randn(64, 512, device=dev)
text = torch.randn(9, 512, device=dev)
cum = torch.zeros(4, dtype=torch.int64, device=dev)
recs = [dict(sentence_row=3 * j, noun_phrase_row=3 * j + 1, other_row0=3 * j + 2, n_other=1, dirflag=s.dirflag, relaword=s.relaflag,
             has_other_nouns=s.n_nouns != 0, black=1.8, imgattn=s.imgattn, target=ref.target) for j, s in enumerate(ref.sentences)]


def timed(fn, name, n=20, refs=1):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    print(f"{name:52s} {us / refs:8.1f} us per ref" + (f"  ({us:8.1f} us per call over {refs} refs)" if refs > 1 else ""))


def fused():
    ops.score_ref(hybrid, text, ref.boxes, ref.masks, recs, 100.0, 0.5, 3, 6, 0.6, cum=cum)


def per_sentence():
    for j, s in enumerate(ref.sentences):
        gem = ops.coherence_scores(s.imgattn, ref.masks, s.dirflag, 1.8)
        idx, _, _ = ops.score_sentence(hybrid, text[3 * j], text[3 * j + 1], text[3 * j + 2:3 * j + 3], ref.boxes, gem, 100.0, 0.5, 3, 6, 0.6,
                                       s.relaflag, s.n_nouns != 0)
        cum[0:2] += ops.iou_select(ref.masks, idx, 0, ref.target)
        cum[2:4] += ops.iou_select(ref.masks, idx, 1, ref.target)


# sixteen refs with their own masks / heat-maps / targets (0.5 GB of mask planes: nothing is served from a cache twice)
grp = []
for i in range(16):
    r, _ = synthetic_ref(i, dev, N=64)
    grp.append(dict(hybrid=hybrid, text=text, boxes=r.boxes, masks=r.masks, k1=3, k2=6,
                    sentences=[dict(sentence_row=3 * j, noun_phrase_row=3 * j + 1, other_row0=3 * j + 2, n_other=1, dirflag=s.dirflag,
                                    relaword=s.relaflag, has_other_nouns=s.n_nouns != 0, black=1.8, imgattn=s.imgattn, target=r.target)
                               for j, s in enumerate(r.sentences)]))


def group16():
    ops.score_group(grp, 100.0, 0.5, 0.6, cum=cum)


def per_ref16():
    for q in grp:
        ops.score_ref(q["hybrid"], q["text"], q["boxes"], q["masks"], q["sentences"], 100.0, 0.5, 3, 6, 0.6, cum=cum)


timed(group16, "hgl_score_group, 16 distinct refs (4 launches)", refs=16)
timed(per_ref16, "hgl_score_ref x 16 distinct refs (64 launches)", refs=16)
timed(fused, "hgl_score_ref (4 launches)")
timed(per_sentence, "per-sentence launches (3 sentences)")
timed(lambda: ops.coherence_scores(ref.sentences[0].imgattn, ref.masks, "left", 1.8), "hgl_coherence_scores, one sentence")
