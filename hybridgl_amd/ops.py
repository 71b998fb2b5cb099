"""Torch-tensor front ends of the C-ABI operators (device memory + stream plumbing only).

Every function validates device/dtype/contiguity, then passes raw pointers to
libhybridgl.so.  Nothing here computes: a missing library or a CPU tensor raises.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import check

ACT = {"none": 0, "quickgelu": 1, "gelu": 2, "relu": 3}
MASK = {"none": 0, "causal": 1, "cls_keep": 2}
DIRFLAG = {"none": 0, "left": 1, "right": 2, "middle": 3}
# utils.py:240-268 relation words; anything else falls through to "none" semantics there
RELAWORD = {"none": 0, "left": 1, "right": 2, "up": 3, "down": 4, "big": 5, "small": 6, "within": 7}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise _lib.HybridGLError(f"{name}: tensor must live on the GPU (no CPU path exists)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def _u8(t, name):
    """bool / uint8 mask tensor -> pointer (torch.bool is one byte, 0/1)."""
    if t.dtype == torch.bool:
        t = t.view(torch.uint8)
    return _dev(t, torch.uint8, name), t


def gemm(a, w, bias=None, residual=None, act="none", out=None):
    """out = act(a @ w.T + bias) + residual   (torch.nn.functional.linear semantics)."""
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    check(lib.hgl_gemm_f32(_dev(a, torch.float32, "a"), _dev(w, torch.float32, "w"),
                           _dev(bias, torch.float32, "bias") if bias is not None else None,
                           _dev(residual, torch.float32, "residual") if residual is not None else None,
                           _dev(out, torch.float32, "out"), M, N, K, K, K, N, N, 1, 0, 0, 0, 0,
                           ACT[act], _stream()), "hgl_gemm_f32")
    return out


_split_cache = {}  # fp32 weight data_ptr -> (weight, hi, lo): keeps the registered halves alive while a model owns them


def register_split_weight(w, scale_log2=None):
    """Register the fp16 hi/lo split of a [N,K] fp32 weight for the HGL_PREC_F16X3 GEMM path.
    The power-of-two scale puts max|w| at <= 2^14 (lo halves stay normal fp16); scale_log2=0 registers the UNSCALED split (the
    values a kernel's own hgl_split_hi_lo gives: SAM's rel-pos tables for the windowed attention).  Returns the registry key;
    the owner hands its keys to release_split_weights when it dies (the models do that through weakref.finalize)."""
    import math
    lib = _lib.load()
    key = w.data_ptr()
    if key in _split_cache:
        return key
    N, K = w.shape
    amax = float(w.abs().max().item())
    s = 0 if amax == 0 else max(-24, min(24, 14 - math.ceil(math.log2(amax))))
    if scale_log2 is not None:
        s = int(scale_log2)
    hi = torch.empty((N, K), dtype=torch.float16, device=w.device)
    lo = torch.empty((N, K), dtype=torch.float16, device=w.device)
    check(lib.hgl_register_split_weight(_dev(w, torch.float32, "w"), N, K, s, hi.data_ptr(), lo.data_ptr(),
                                        _stream()), "hgl_register_split_weight")
    _split_cache[key] = (w, hi, lo)
    return key


def release_split_weights(keys):
    """Drop the registered splits of `keys` (library registry + the hi/lo tensors): called when their model dies."""
    try:
        lib = _lib.load()
    except Exception:   # interpreter shutdown
        return
    for key in keys:
        if _split_cache.pop(key, None) is not None:
            lib.hgl_unregister_split_weight(key)


def split_weight_keys_since(before):
    """keys registered after the snapshot `before` (= set(_split_cache)): what a constructor has added"""
    return [k for k in _split_cache if k not in before]


def default_precision():
    """'f16x3' unless HYBRIDGL_PRECISION=f32: both meet the parity bar (tests run both)."""
    import os
    return os.environ.get("HYBRIDGL_PRECISION", "f16x3")


_precision_now = None


def set_precision(mode):
    """'f32' (exact fp32 MFMA) or 'f16x3' (split-fp16 MFMA, fp32-class accuracy).  The library keeps ONE current mode;
    every model carries its own and re-asserts it on entry (use_precision), so models of different precision can live
    in one process."""
    global _precision_now
    check(_lib.load().hgl_set_precision({"f32": 0, "f16x3": 1}[mode]), "hgl_set_precision")
    _precision_now = mode


def use_precision(mode):
    """make `mode` the library's current precision if it is not already (called on entry by every model method)"""
    if mode != _precision_now:
        set_precision(mode)


def split_overflow_count(reset=True, sync=True):
    """Number of GPU threads of the f16x3 path that met an activation beyond the fp16 range since the last reset
    (hgl_split_overflow_count).  Synchronises the device first unless sync=False."""
    if sync:
        torch.cuda.synchronize()
    n = C.c_ulonglong(0)
    check(_lib.load().hgl_split_overflow_count(1 if reset else 0, C.byref(n)), "hgl_split_overflow_count")
    return int(n.value)


def split_overflow_peek(buf):
    """Enqueue, on the current stream, a copy of the two overflow counters into `buf` (pinned int32 [2] host tensor): no
    wait, no reset.  Read buf after an event recorded behind this call has completed."""
    assert buf.is_pinned() and buf.dtype == torch.int32 and buf.numel() >= 2
    check(_lib.load().hgl_split_overflow_peek_async(buf.data_ptr(), _stream()), "hgl_split_overflow_peek_async")


class SplitOverflow(_lib.HybridGLError):
    """an activation left the fp16 range in f16x3 mode: the results since the last clean check contain inf / NaN"""


def check_split_overflow():
    """raise if the f16x3 path met a value it cannot represent (the results since the last check are then not
    fp32-class); the cure is HYBRIDGL_PRECISION=f32"""
    n = split_overflow_count(reset=True)
    if n:
        raise SplitOverflow(f"activations exceeded the fp16 range (|x| > 65504) in f16x3 mode ({n} GPU threads saw one): "
                            "the results contain inf / NaN; rerun with HYBRIDGL_PRECISION=f32 (or precision='f32')")


X3_KERNELS = {"auto": -1, "v1": 0, "P": 1}


def select_x3_kernel(kind="auto"):
    """Pins the tiling of the f16x3 GEMM ('auto' = cost model; all tilings are bit-identical)."""
    check(_lib.load().hgl_gemm_f16x3_select(X3_KERNELS[kind]), "hgl_gemm_f16x3_select")


def gemm_f16x3(a, w, bias=None, residual=None, act="none", out=None, balanced=False):
    """gemm() through the split-fp16 matrix-core path (registers w on first use).  balanced=True: the row-balanced launch of
    the model code's residual GEMMs (whole rounds of the persistent tiling + a split-K tail; hgl_gemm_f16x3 with scratch for
    the tail's partial sums)."""
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    register_split_weight(w)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    # the tail is at most half a round of 256 x 256 tiles in k K slices with k x tiles <= CUs: <= 256 slice-tiles of 256 KiB
    extra = (2 * 256 * 256 * 256 * 4 + 4096) if balanced else 0
    need = (M * K * 4 + 255) // 256 * 256 + extra      # the SIZE handed over selects the launch, not the (grow-only) buffer's
    ws = workspace(need, a.device, "gemm_f16x3")
    check(lib.hgl_gemm_f16x3(_dev(a, torch.float32, "a"), _dev(w, torch.float32, "w"),
                             _dev(bias, torch.float32, "bias") if bias is not None else None,
                             _dev(residual, torch.float32, "residual") if residual is not None else None,
                             _dev(out, torch.float32, "out"), M, N, K, ACT[act], ws.data_ptr(), need,
                             _stream()), "hgl_gemm_f16x3")
    return out


def layernorm(x, w, b, eps=1e-5):
    lib = _lib.load()
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x)
    check(lib.hgl_layernorm_f32(_dev(x, torch.float32, "x"), _dev(w, torch.float32, "w"),
                                _dev(b, torch.float32, "b"), _dev(y, torch.float32, "y"), rows, D,
                                float(eps), _stream()), "hgl_layernorm_f32")
    return y


def attention(q, k, v, heads, scale=None, mask="none", keep=None, keep_b0=0, keep_n=0,
              rel_h=None, rel_w=None):
    """q: [B,Sq,H*hd], k,v: [B,Sk,H*hd] contiguous -> [B,Sq,H*hd]."""
    lib = _lib.load()
    B, Sq, Dm = q.shape
    Sk = k.shape[1]
    hd = Dm // heads
    if scale is None:
        scale = hd ** -0.5
    out = torch.empty_like(q)
    kp = None
    if keep is not None:
        kp, keep = _u8(keep, "keep")
    kh = kw = 0
    if rel_h is not None:
        kh, kw = rel_h.shape[-1], rel_w.shape[-1]
    check(lib.hgl_attention_f32(_dev(q, torch.float32, "q"), _dev(k, torch.float32, "k"),
                                _dev(v, torch.float32, "v"), _dev(out, torch.float32, "out"),
                                B, heads, Sq, Sk, hd, Dm, Dm, Dm, Dm,
                                Sq * Dm, Sk * Dm, Sk * Dm, Sq * Dm, float(scale), MASK[mask], kp,
                                keep_b0, keep_n,
                                _dev(rel_h, torch.float32, "rel_h") if rel_h is not None else None,
                                _dev(rel_w, torch.float32, "rel_w") if rel_w is not None else None,
                                kh, kw, _stream()), "hgl_attention_f32")
    return out


def attention_presplit(qkv, heads, scale=None, mask="none", keep=None, keep_b0=0, keep_n=0):
    """qkv: [B,S,3*H*hd] fp32 contiguous (the in-projection output, q | k | v) -> [B,S,H*hd] through the kernels that take
    the fp16 hi / lo planes (csrc/attention_ps.hip); split-fp16 mode, hd in {64, 80}."""
    lib = _lib.load()
    B, S, D3 = qkv.shape
    Dm = D3 // 3
    hd = Dm // heads
    if scale is None:
        scale = hd ** -0.5
    out = torch.empty((B, S, Dm), dtype=torch.float32, device=qkv.device)
    scratch = torch.empty(B * S * D3, dtype=torch.float32, device=qkv.device)
    kp = None
    if keep is not None:
        kp, keep = _u8(keep, "keep")
    check(lib.hgl_attention_presplit_f32(_dev(qkv, torch.float32, "qkv"), D3, B, heads, S, hd, _dev(out, torch.float32, "out"), Dm,
                                         float(scale), MASK[mask], kp, keep_b0, keep_n, scratch.data_ptr(), scratch.numel() * 4,
                                         _stream()), "hgl_attention_presplit_f32")
    return out


def mask_resize(masks, g):
    """TF.resize(masks.float(), (g,g)) of model/backbone.py:160 -> [N, g*g] fp32."""
    lib = _lib.load()
    N, H, W = masks.shape
    mp, masks = _u8(masks, "masks")
    pm = torch.empty((N, g * g), dtype=torch.float32, device=masks.device)
    check(lib.hgl_mask_resize(mp, N, H, W, g, _dev(pm, torch.float32, "pm"), _stream()), "hgl_mask_resize")
    return pm


def calculate_score(img, txt, logit_scale):
    lib = _lib.load()
    N, E = img.shape
    T = txt.shape[0]
    out = torch.empty((N, T), dtype=torch.float32, device=img.device)
    check(lib.hgl_calculate_score(_dev(img, torch.float32, "img"), _dev(txt, torch.float32, "txt"), N, T, E,
                                  float(logit_scale), _dev(out, torch.float32, "out"), _stream()),
          "hgl_calculate_score")
    return out


_ws_cache = {}


def workspace(nbytes, device, tag="default"):
    """Grow-only device workspace (allocated outside the timed/hot path, reused).  One buffer per (tag, stream): the
    stages of the evaluation loop run on their own streams, and a buffer that grows (a larger image, more proposals)
    is replaced while kernels of ANOTHER stream could still be using the old one -- the caching allocator only orders
    reuse within the allocating stream."""
    key = (str(device), tag, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def coherence_scores(imgattn, masks, dirflag="none", black=1.8):
    """Hybridgl_main.py:201-223 for all masks at once -> [N] fp32."""
    lib = _lib.load()
    N, H, W = masks.shape
    assert imgattn.shape == (H, W)
    mp, masks = _u8(masks, "masks")
    need = lib.hgl_coherence_workspace_bytes(N, H, W)
    ws = workspace(need, masks.device, "coherence")
    out = torch.empty((N,), dtype=torch.float32, device=masks.device)
    check(lib.hgl_coherence_scores(_dev(imgattn, torch.float32, "imgattn"), mp, N, H, W,
                                   DIRFLAG.get(dirflag, 0), float(black), _dev(out, torch.float32, "out"),
                                   ws.data_ptr(), ws.numel(), _stream()), "hgl_coherence_scores")
    return out


def iou_counts(pred, gt):
    """Compute_IoU (utils.py:365-384) counts: tensor([I, U]) int64 on device."""
    lib = _lib.load()
    pp, pred = _u8(pred, "pred")
    gp, gt = _u8(gt, "gt")
    assert pred.numel() == gt.numel()
    out = torch.empty((2,), dtype=torch.int64, device=pred.device)
    check(lib.hgl_iou(pp, gp, pred.numel(), out.data_ptr(), _stream()), "hgl_iou")
    return out


def iou_select(masks, idx, which, gt):
    """Compute_IoU(masks[idx[which]], gt) with the index resolved on the device -> tensor([I,U])."""
    lib = _lib.load()
    mp, masks = _u8(masks, "masks")
    gp, gt = _u8(gt, "gt")
    HW = gt.numel()
    assert masks.numel() % HW == 0
    out = torch.empty((2,), dtype=torch.int64, device=masks.device)
    check(lib.hgl_iou_select(mp, _dev(idx, torch.int32, "idx"), int(which), gp, HW, out.data_ptr(), _stream()),
          "hgl_iou_select")
    return out


def score_sentence(hybrid, sentence_feat, noun_phrase_feat, other_noun_feats, boxes, gem_score,
                   logit_scale=100.0, r=0.5, k1=3, k2=6, alpha=0.6, relaword="none", has_other_nouns=False):
    """Per-sentence tail (Hybridgl_main.py:153-196,225-228).

    other_noun_feats: [K,E] tensor or None.  Returns (idx[2] int32: pure argmax, final index;
    score_clip [N]; score_neg [N])."""
    lib = _lib.load()
    N, E = hybrid.shape
    dev = hybrid.device
    idx = torch.empty((2,), dtype=torch.int32, device=dev)
    sc = torch.empty((N,), dtype=torch.float32, device=dev)
    sn = torch.empty((N,), dtype=torch.float32, device=dev)
    need = lib.hgl_score_sentence_workspace_bytes(N, E)
    ws = workspace(need, dev, "score_sentence")
    n_other = 0 if other_noun_feats is None else int(other_noun_feats.shape[0])
    sent = sentence_feat.reshape(-1)
    nphr = noun_phrase_feat.reshape(-1)
    check(lib.hgl_score_sentence(_dev(hybrid, torch.float32, "hybrid"),
                                 _dev(sent, torch.float32, "sentence_feat"),
                                 _dev(nphr, torch.float32, "noun_phrase_feat"),
                                 _dev(other_noun_feats, torch.float32, "other_noun_feats") if n_other else None,
                                 n_other, float(r),
                                 _dev(boxes, torch.int64, "boxes"), _dev(gem_score, torch.float32, "gem_score"),
                                 N, E, float(logit_scale), int(k1), int(k2), float(alpha),
                                 RELAWORD.get(relaword, 0), int(bool(has_other_nouns)),
                                 idx.data_ptr(), sc.data_ptr(), sn.data_ptr(), ws.data_ptr(), ws.numel(),
                                 _stream()), "hgl_score_sentence")
    return idx, sc, sn


def score_ref(hybrid, text, boxes, masks, sentences, logit_scale=100.0, r=0.5, k1=3, k2=6, alpha=0.6, cum=None, want_scores=False):
    """The whole tail of one dataset item (Hybridgl_main.py:153-230) in four launches: hgl_score_ref.

    hybrid [N,E], text [T,E] (every string of the ref), boxes [N,4] int64 XYWH, masks [N,H,W] bool / uint8;
    sentences: list of dicts {sentence_row, noun_phrase_row, other_row0, n_other, dirflag, relaword (strings as in the
    reference), has_other_nouns, black, imgattn [H,W] fp32 tensor, target [H,W] bool / uint8 tensor}.
    cum: int64 [4] device tensor incremented in place by (I, U, I_final, U_final) summed over the sentences.
    Returns (idx [S,2] int32, iu [S,4] int64) and, with want_scores, (score_clip, score_neg, gem) [S,N] each."""
    lib = _lib.load()
    N, E = hybrid.shape
    T = text.shape[0]
    S = len(sentences)
    mp, masks = _u8(masks, "masks")
    _, H, W = masks.shape
    dev = hybrid.device
    recs = (_lib.HglSentence * S)()
    keep = []       # tensors whose pointers sit in the records
    for j, q in enumerate(sentences):
        a = q["imgattn"]
        if tuple(a.shape) != (H, W):
            raise ValueError(f"sentence {j}: imgattn {tuple(a.shape)} != masks {(H, W)}")
        tp, tt = _u8(q["target"], "target")
        if tuple(tt.shape) != (H, W):
            raise ValueError(f"sentence {j}: target {tuple(tt.shape)} != masks {(H, W)}")
        keep += [a, tt]
        recs[j] = _lib.HglSentence(int(q["sentence_row"]), int(q["noun_phrase_row"]), int(q.get("other_row0", 0)), int(q.get("n_other", 0)),
                                   DIRFLAG.get(q.get("dirflag", "none"), 0), RELAWORD.get(q.get("relaword", "none"), 0),
                                   int(bool(q.get("has_other_nouns", False))), float(q.get("black", 1.8)),
                                   _dev(a, torch.float32, "imgattn"), tp)
    idx = torch.empty((S, 2), dtype=torch.int32, device=dev)
    iu = torch.empty((S, 4), dtype=torch.int64, device=dev)
    sc = sn = gm = None
    if want_scores:
        sc, sn, gm = (torch.empty((S, N), dtype=torch.float32, device=dev) for _ in range(3))
    need = lib.hgl_score_ref_workspace_bytes(S, N, E, H, W)
    ws = workspace(need, dev, "score_ref")
    check(lib.hgl_score_ref(_dev(hybrid, torch.float32, "hybrid"), _dev(text, torch.float32, "text"), T,
                            _dev(boxes, torch.int64, "boxes"), mp, N, E, H, W, recs, S, float(logit_scale), float(r), int(k1), int(k2),
                            float(alpha), idx.data_ptr(), iu.data_ptr(), _dev(cum, torch.int64, "cum") if cum is not None else None,
                            sc.data_ptr() if sc is not None else None, sn.data_ptr() if sn is not None else None,
                            gm.data_ptr() if gm is not None else None, ws.data_ptr(), ws.numel(), _stream()), "hgl_score_ref")
    return (idx, iu, sc, sn, gm) if want_scores else (idx, iu)


SCORE_GROUP_MAX_SENTENCES = 16      # per ref and call (csrc/scoring.hip REF_MAXS)


def score_group(refs, logit_scale=100.0, r=0.5, alpha=0.6, cum=None, want_scores=False):
    """The tails of the R refs of a group (Hybridgl_main.py:153-230 each) in ONE set of four launches: hgl_score_group.
    refs: list of dicts {hybrid [N,E], text [T,E], boxes [N,4] int64, masks [N,H,W], sentences (as for score_ref, at most
    16), k1, k2}; shapes may differ from ref to ref.  cum as for score_ref.  Returns one tuple per ref, as score_ref would
    (rows identical to its rows)."""
    lib = _lib.load()
    R = len(refs)
    recs = (_lib.HglGroupRef * R)()
    keep, outs = [], []
    E = refs[0]["hybrid"].shape[1]
    dev = refs[0]["hybrid"].device
    for i, q in enumerate(refs):
        hybrid, text = q["hybrid"], q["text"]
        N, T = hybrid.shape[0], text.shape[0]
        mp, masks = _u8(q["masks"], "masks")
        _, H, W = masks.shape
        S = len(q["sentences"])
        if not 1 <= S <= SCORE_GROUP_MAX_SENTENCES:
            raise ValueError(f"ref {i}: {S} sentences (1 .. {SCORE_GROUP_MAX_SENTENCES} per call)")
        sent = (_lib.HglSentence * S)()
        for j, t in enumerate(q["sentences"]):
            a = t["imgattn"]
            if tuple(a.shape) != (H, W):
                raise ValueError(f"ref {i} sentence {j}: imgattn {tuple(a.shape)} != masks {(H, W)}")
            tp, tt = _u8(t["target"], "target")
            if tuple(tt.shape) != (H, W):
                raise ValueError(f"ref {i} sentence {j}: target {tuple(tt.shape)} != masks {(H, W)}")
            keep += [a, tt]
            sent[j] = _lib.HglSentence(int(t["sentence_row"]), int(t["noun_phrase_row"]), int(t.get("other_row0", 0)), int(t.get("n_other", 0)),
                                       DIRFLAG.get(t.get("dirflag", "none"), 0), RELAWORD.get(t.get("relaword", "none"), 0),
                                       int(bool(t.get("has_other_nouns", False))), float(t.get("black", 1.8)),
                                       _dev(a, torch.float32, "imgattn"), tp)
        idx = torch.empty((S, 2), dtype=torch.int32, device=dev)
        iu = torch.empty((S, 4), dtype=torch.int64, device=dev)
        sc = sn = gm = None
        if want_scores:
            sc, sn, gm = (torch.empty((S, N), dtype=torch.float32, device=dev) for _ in range(3))
        keep += [sent, masks]
        recs[i] = _lib.HglGroupRef(_dev(hybrid, torch.float32, "hybrid"), _dev(text, torch.float32, "text"), T,
                                   _dev(q["boxes"], torch.int64, "boxes"), mp, N, H, W, sent, S, int(q["k1"]), int(q["k2"]),
                                   idx.data_ptr(), iu.data_ptr(), sc.data_ptr() if sc is not None else None,
                                   sn.data_ptr() if sn is not None else None, gm.data_ptr() if gm is not None else None)
        outs.append((idx, iu, sc, sn, gm) if want_scores else (idx, iu))
    need = lib.hgl_score_group_workspace_bytes(recs, R, E)
    ws = workspace(need, dev, "score_group")
    check(lib.hgl_score_group(recs, R, E, float(logit_scale), float(r), float(alpha),
                              _dev(cum, torch.int64, "cum") if cum is not None else None, ws.data_ptr(), ws.numel(), _stream()),
          "hgl_score_group")
    return outs


def synthesize_views(sam_img, blurred, image_norm, masks, res=224, out=None):
    """Hybridgl_main.py:93-125 -> (local_imgs, global_imgs) [N,3,res,res] fp32 (written into `out` when given)."""
    lib = _lib.load()
    N, H, W = masks.shape
    mp, masks = _u8(masks, "masks")
    dev = masks.device
    if out is not None:
        loc, glo = out
        assert tuple(loc.shape) == (N, 3, res, res) == tuple(glo.shape) and loc.is_contiguous() and glo.is_contiguous()
    else:
        loc = torch.empty((N, 3, res, res), dtype=torch.float32, device=dev)
        glo = torch.empty((N, 3, res, res), dtype=torch.float32, device=dev)
    check(lib.hgl_synthesize_views(_dev(sam_img, torch.uint8, "sam_img"), _dev(blurred, torch.uint8, "blurred"),
                                   _dev(image_norm, torch.float32, "image_norm"), mp, N, H, W, res,
                                   loc.data_ptr(), glo.data_ptr(), _stream()), "hgl_synthesize_views")
    return loc, glo


def cv_gaussian_kernel_q8(k=15, sigma=0.0):
    """the k 8.8 fixed-point taps of cv2.GaussianBlur on uint8 (sum 256) -> ctypes uint16 array (host)"""
    import ctypes as C
    taps = (C.c_uint16 * k)()
    check(_lib.load().hgl_cv_gaussian_kernel_q8(k, float(sigma), taps), "hgl_cv_gaussian_kernel_q8")
    return taps


def gaussian_blur_u8(img, k=15, sigma=0.0, mode="cv2"):
    """cv2.GaussianBlur(img, (k, k), sigma) on a [H,W,C] uint8 device tensor (Hybridgl_main.py:99).
    mode="cv2": OpenCV's 8-bit fixed-point path restated (integer arithmetic; parity with the package unpinned);
    mode="float": the double-precision separable Gaussian of synth.box_blur_u8, bit-identical to it."""
    import ctypes as C
    import numpy as np
    lib = _lib.load()
    H, W, Cc = img.shape
    out = torch.empty_like(img)
    ws = workspace(lib.hgl_gaussian_blur_u8_workspace_bytes(H, W, Cc), img.device, "blur")
    if mode == "cv2":
        taps = cv_gaussian_kernel_q8(k, sigma)
        check(lib.hgl_gaussian_blur_u8_q8(_dev(img, torch.uint8, "img"), H, W, Cc, taps, taps, k, out.data_ptr(),
                                          ws.data_ptr(), ws.numel(), _stream()), "hgl_gaussian_blur_u8_q8")
        return out
    assert mode == "float"
    sigma = sigma if sigma > 0 else 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    r = k // 2
    x = np.arange(-r, r + 1, dtype=np.float64)
    g = np.exp(-(x * x) / (2 * sigma * sigma))
    g /= g.sum()
    taps = (C.c_double * k)(*[float(v) for v in g])
    check(lib.hgl_gaussian_blur_u8(_dev(img, torch.uint8, "img"), H, W, Cc, taps, k, out.data_ptr(), ws.data_ptr(),
                                   ws.numel(), _stream()), "hgl_gaussian_blur_u8")
    return out
