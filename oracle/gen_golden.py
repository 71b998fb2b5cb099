"""Golden-vector generator: runs the REFERENCE's own Python (read-only import from
/root/reference, build container only) on seeded inputs and stores inputs/outputs as small
.npz fixtures under tests/golden/.  Nothing from the reference (source or bytecode) is copied.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--only clip_tiny,...]

Stubs (packages absent offline; semantics restated from the pinned versions in
environment.yaml): torchvision 0.15.2 `TF.resize` on a float tensor == F.interpolate(bilinear,
align_corners=False, antialias=False); `clip.load` returns the reference's own CLIP class built
from our seeded state_dict; spacy/cv2/matplotlib/gem are import-only stubs (never called).
"""
import argparse
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn.functional as F

REF = os.environ.get("HYBRIDGL_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hybridgl_amd import weights  # noqa: E402
from hybridgl_amd.synth import synth_masks, synth_image  # noqa: E402
from oracle.cases import edge_masks, views_for_case, resize_case, RESIZE_CASES, tail_case, tie_case, TIE_PLAN, nan_case, NAN_PLAN, GLUE_PLAN, glue_tokens  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def install_stubs():
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")

    def resize(img, size, interpolation=None, max_size=None, antialias=None):
        if isinstance(size, int):
            size = (size, size)
        x = img
        squeeze = 0
        while x.dim() < 4:
            x = x.unsqueeze(0)
            squeeze += 1
        y = F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False, antialias=bool(antialias))
        for _ in range(squeeze):
            y = y.squeeze(0)
        return y

    tvf.resize = resize
    tvf._tensor_resize = resize
    tvt.functional = tvf
    tv.transforms = tvt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvf})
    for name in ("spacy", "cv2", "gem", "ftfy"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["ftfy"].fix_text = lambda s: s
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        mpl = types.ModuleType("matplotlib")
        sys.modules["matplotlib"] = mpl
        sys.modules["matplotlib.pyplot"] = types.ModuleType("matplotlib.pyplot")
        sys.modules["matplotlib.gridspec"] = types.ModuleType("matplotlib.gridspec")


def ref_clip_model_module():
    return _load("ref_clip_model", os.path.join(REF, "third_party/modified_CLIP/clip/model.py"))


def build_ref_clip(cfg_name, seed):
    """The reference's CLIP class with our seeded weights (fp32, as clip/model.py:509 leaves it)."""
    m = ref_clip_model_module()
    cfg = weights.CLIP_CONFIGS[cfg_name]
    model = m.CLIP(cfg["embed_dim"], cfg["image_resolution"], cfg["vision_layers"], cfg["vision_width"],
                   cfg["vision_patch_size"], cfg["context_length"], cfg["vocab_size"],
                   cfg["transformer_width"], cfg["transformer_heads"], cfg["transformer_layers"])
    sd = weights.clip_state_dict(cfg_name, seed)
    missing = model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    return model.eval().float()


def build_ref_backbone(cfg_name, seed):
    """The reference's CLIPViTFM with `clip.load` stubbed to return build_ref_clip()."""
    clip_stub = types.ModuleType("clip")
    clip_stub.load = lambda name, *a, **k: (build_ref_clip(cfg_name, seed), None)
    sys.modules["clip"] = clip_stub
    bb = _load("ref_backbone", os.path.join(REF, "model/backbone.py"))
    model = bb.CLIPViTFM(model_name="ViT-B/16")
    # make_attn_mask expands to N*num_heads rows (model/backbone.py:113-114): must equal the
    # transformer's real head count for geometries other than ViT-B
    model.num_heads = weights.CLIP_CONFIGS[cfg_name]["vision_width"] // 64
    return model.eval()


MODES = ["G2L", "L2G", "G2L&L2G", "token_masking", "attn_masking", "crop"]


def gen_clip(cfg_name, seed, Ns, H, W, modes, tag):
    model = build_ref_backbone(cfg_name, seed)
    res = weights.CLIP_CONFIGS[cfg_name]["image_resolution"]
    out = {}
    for N in Ns:
        # inputs are NOT stored: tests regenerate them from the same seeds (views_for_case)
        loc, glo, masks = views_for_case(N, res, H, W)
        out[f"N{N}_masks"] = np.packbits(masks, axis=-1)
        loc_t, glo_t = torch.from_numpy(loc), torch.from_numpy(glo)
        for mode in modes:
            with torch.no_grad():
                y = model(local_imgs=loc_t, global_imgs=glo_t, pred_masks=torch.from_numpy(masks),
                          fusion_mode=mode, masking_block=9)
            out[f"N{N}_{mode}"] = y.numpy().astype(np.float32)
            print(tag, N, mode, y.shape, float(y.abs().mean()))
    out["meta"] = np.array([seed, H, W], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, f"{tag}.npz"), **out)


MB_PLAN = [(None, MODES), (0, MODES), (1, MODES), (10, MODES), (11, ["G2L", "L2G", "G2L&L2G", "token_masking", "crop"])]


def gen_clip_mb(tag="clip_tiny_mb"):
    """CLIPViTFM.forward on the tiny geometry (12 blocks, last_layer 10) with masking_block at the ends of its range: None
    (= last_layer), 0, 1, last_layer and last_layer + 1 (where the reference still returns features: not attn_masking,
    whose return sits at block last_layer, model/backbone.py:197-203)."""
    model = build_ref_backbone("tiny", 0)
    loc, glo, masks = views_for_case(3, 64, 97, 130)
    out = {"last_layer": np.array([model.last_layer], dtype=np.int64)}
    for mb, modes in MB_PLAN:
        for mode in modes:
            with torch.no_grad():
                y = model(local_imgs=torch.from_numpy(loc), global_imgs=torch.from_numpy(glo), pred_masks=torch.from_numpy(masks),
                          fusion_mode=mode, masking_block=mb)
            out[f"mb{mb}_{mode}"] = y.numpy().astype(np.float32)
            print(tag, mb, mode, y.shape)
    np.savez_compressed(os.path.join(GOLD, f"{tag}.npz"), **out)


def gen_text(cfg_name, seed, tag):
    model = build_ref_clip(cfg_name, seed)
    cfg = weights.CLIP_CONFIGS[cfg_name]
    rng = np.random.default_rng(7)
    B, S, V = 6, cfg["context_length"], cfg["vocab_size"]
    tok = np.zeros((B, S), dtype=np.int64)
    for b in range(B):
        n = int(rng.integers(1, S - 2))
        tok[b, 0] = V - 2                       # SOT (49406 for the real vocab)
        tok[b, 1:1 + n] = rng.integers(1, V - 2, size=n)
        tok[b, 1 + n] = V - 1                   # EOT = highest id -> argmax pooling
    with torch.no_grad():
        y = model.encode_text(torch.from_numpy(tok))
    np.savez_compressed(os.path.join(GOLD, f"{tag}.npz"), tokens=tok.astype(np.int32),
                        out=y.numpy().astype(np.float32), meta=np.array([seed], dtype=np.int64))
    print(tag, y.shape)


def gen_text_pool(cfg_name, seed, tag, masking_blocks):
    """The two text-tower variants off the Hybridgl_main path: CLIP.encode_text(text, target_noun_index)
    (clip/model.py:426-428: pooled at target_noun_index + 1; 0 / None fall through to the EOT) and
    CLIPViTFM.text_masking_feature (model/backbone.py:34-56: positions masking_index + 1 zeroed before every block
    >= masking_block)."""
    model = build_ref_clip(cfg_name, seed)
    bb = build_ref_backbone(cfg_name, seed)
    cfg = weights.CLIP_CONFIGS[cfg_name]
    rng = np.random.default_rng(17)
    B, S, V = 5, cfg["context_length"], cfg["vocab_size"]
    tok = np.zeros((B, S), dtype=np.int64)
    for b in range(B):
        n = int(rng.integers(5, min(S - 2, 12)))
        tok[b, 0] = V - 2
        tok[b, 1:1 + n] = rng.integers(1, V - 2, size=n)
        tok[b, 1 + n] = V - 1
    out = {"tokens": tok.astype(np.int32), "masking_blocks": np.array(masking_blocks, dtype=np.int64)}
    t = torch.from_numpy(tok)
    with torch.no_grad():
        for k in (0, 1, 3):
            out[f"pool_{k}"] = model.encode_text(t, target_noun_index=k).numpy().astype(np.float32)
        for mb in masking_blocks:
            out[f"mask_{mb}_idx12"] = bb.text_masking_feature(t, masking_index=[1, 2], masking_block=mb).numpy().astype(np.float32)
            out[f"mask_{mb}_idx0"] = bb.text_masking_feature(t, masking_index=[0], masking_block=mb).numpy().astype(np.float32)
        out["mask_none"] = bb.text_masking_feature(t, masking_index=[], masking_block=masking_blocks[0]).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, f"{tag}.npz"), **out)
    print(tag, {k: v.shape for k, v in out.items()})


def gen_scoring():
    """utils.py functions + CLIPViTFM.calculate_score + the inline tail of Hybridgl_main.py:153-230
    driven through the reference's own helpers (relation_boxes, gen_dir_mask, Compute_IoU)."""
    sys.path.insert(0, REF)
    utils = _load("ref_utils", os.path.join(REF, "utils.py"))
    bb = build_ref_backbone("tiny", 0)
    rng = np.random.default_rng(11)
    out = {}
    # --- calculate_score
    img = rng.standard_normal((9, 32)).astype(np.float32)
    txt = rng.standard_normal((2, 32)).astype(np.float32)
    with torch.no_grad():
        out["cs_img"], out["cs_txt"] = img, txt
        out["cs_out"] = bb.calculate_score(torch.from_numpy(img), torch.from_numpy(txt)).numpy()
        out["cs_logit_scale"] = np.float32(bb.model.logit_scale.exp().item())
    # --- relation_boxes table over all words (+ an unknown word)
    boxes = np.array([[10, 10, 50, 40], [100, 20, 30, 30], [5, 5, 200, 200], [60, 60, 50, 40]], dtype=np.int64)
    scores = np.array([0.5, 0.3, 0.2, 0.1], dtype=np.float32)
    words = ["none", "left", "right", "up", "down", "big", "small", "within", "other"]
    tab = np.zeros((len(words), 4, 4), dtype=np.float32)
    bt, st = torch.from_numpy(boxes), torch.from_numpy(scores)
    for w, word in enumerate(words):
        for i in range(4):
            for j in range(4):
                tab[w, i, j] = float(utils.relation_boxes(bt[i], bt[j], st[i], st[j], word))
    out["rb_boxes"], out["rb_scores"], out["rb_table"] = boxes, scores, tab
    out["rb_words"] = np.array(words)
    # --- gen_dir_mask
    for flag in ["left", "right", "middle", "none", "up"]:
        for (h, w) in [(3, 5), (4, 8), (2, 640), (2, 427)]:
            out[f"dm_{flag}_{h}_{w}"] = utils.gen_dir_mask(flag, h, w, None).numpy().astype(np.float32)
    # --- Compute_IoU
    p = rng.random((24, 31)) > 0.6
    t = rng.random((24, 31)) > 0.5
    iou, lst, cI, cU = utils.Compute_IoU(torch.from_numpy(p), torch.from_numpy(t[None].astype(np.uint8)), 0, 0, [])
    out["iou_pred"], out["iou_gt"] = p, t
    out["iou_IU"] = np.array([int(cI), int(cU)], dtype=np.int64)
    # --- whole tail (restated glue, reference helpers) on synthetic inputs
    softmax0 = torch.nn.Softmax(0)
    cases = []
    H, W, N, E = 96, 128, 12, 32
    for ci, (rela, dirflag, has_other) in enumerate([("none", "none", False), ("left", "left", True),
                                                      ("big", "middle", False), ("within", "right", True),
                                                      ("small", "none", True), ("up", "left", False),
                                                      ("down", "none", True), ("right", "none", False)]):
        hybrid, t_pos, t_neg, masks, boxes, attn, gt = tail_case(ci, N, E, H, W)
        with torch.no_grad():
            vf = torch.from_numpy(hybrid)
            score_clip = bb.calculate_score(vf, torch.from_numpy(t_pos))
            score_neg = bb.calculate_score(vf, torch.from_numpy(t_neg))
            idx_pure = int(torch.argmax(score_clip))
            sc_raw, sn_raw = score_clip.numpy()[:, 0].copy(), score_neg.numpy()[:, 0].copy()
            score_clip, score_neg = softmax0(score_clip), softmax0(score_neg)
            k1, k2 = min(3, N), min(6, N)
            _, maxidxs = torch.topk(score_clip.view(-1), k=k1)
            _, maxneg = torch.topk(score_neg.view(-1), k=k2)
            bx = torch.from_numpy(boxes)
            topscores = np.zeros(k1)
            for i in range(k1):
                for j in (maxidxs if not has_other else maxneg):
                    sj = score_clip[j][0] if not has_other else score_neg[j][0]
                    topscores[i] = topscores[i] + utils.relation_boxes(bx[maxidxs[i]], bx[j], score_clip[maxidxs[i]][0], sj, rela)
            topscores = softmax0(torch.Tensor(topscores))
            a = torch.from_numpy(attn)
            a = (a - a.min()) / (a.max() - a.min())
            a = a * utils.gen_dir_mask(dirflag, H, W, None)
            a = a / a.mean()
            black = 1.95 if rela == "big" else (1.5 if rela == "small" else 1.8)
            gem = []
            for pm_ in torch.from_numpy(masks):
                pm_ = pm_.type(torch.uint8)
                gem.append(float((a * (2 - black) * pm_ / (pm_.sum())).sum() - (a * black * (1 - pm_) / ((1 - pm_).sum())).sum()))
            gem = np.array(gem, dtype=np.float32)
            for i in range(k1):
                topscores[i] = topscores[i] * (1 - 0.6) + 0.6 * float(gem[maxidxs[i]])
            idx_final = int(maxidxs[torch.argmax(topscores)])
            _, _, cI, cU = utils.Compute_IoU(torch.from_numpy(masks[idx_final]), torch.from_numpy(gt[None].astype(np.uint8)), 0, 0, [])
        out[f"tail{ci}_gem"] = gem
        out[f"tail{ci}_sc"], out[f"tail{ci}_sn"] = sc_raw, sn_raw
        out[f"tail{ci}_idx"] = np.array([idx_pure, idx_final], dtype=np.int64)
        out[f"tail{ci}_IU"] = np.array([int(cI), int(cU)], dtype=np.int64)
        cases.append(f"{rela},{dirflag},{int(has_other)}")
        print("tail", ci, rela, dirflag, has_other, idx_pure, idx_final)
    out["tail_cases"] = np.array(cases)
    out["tail_hw"] = np.array([H, W, N], dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, "scoring.npz"), **out)


def ref_tail(bb, utils, hybrid, t_pos, t_neg, masks, boxes, attn, gt, rela, dirflag, has_other, k1=3, k2=6, info=None):
    """One sentence of the tail of Hybridgl_main.py:153-230, glue restated, every helper the reference's own
    (calculate_score, relation_boxes, gen_dir_mask, Compute_IoU).  Returns (idx_pure, idx_final, (I, U), k1, k2) with the
    k1 / k2 the reference would carry on to the next sentence (:178-181)."""
    softmax0 = torch.nn.Softmax(0)
    H, W = masks.shape[1:]
    with torch.no_grad():
        vf = torch.from_numpy(hybrid)
        score_clip = bb.calculate_score(vf, torch.from_numpy(t_pos))
        score_neg = bb.calculate_score(vf, torch.from_numpy(t_neg))
        idx_pure = int(torch.argmax(score_clip))
        score_clip, score_neg = softmax0(score_clip), softmax0(score_neg)
        if k1 > len(score_clip):
            k1 = len(score_clip)
        if k2 > len(score_neg):
            k2 = len(score_neg)
        _, maxidxs = torch.topk(score_clip.view(-1), k=k1)
        _, maxneg = torch.topk(score_neg.view(-1), k=k2)
        bx = torch.from_numpy(boxes)
        topscores = np.zeros(k1)
        for i in range(k1):
            for j in (maxidxs if not has_other else maxneg):
                sj = score_clip[j][0] if not has_other else score_neg[j][0]
                topscores[i] = topscores[i] + utils.relation_boxes(bx[maxidxs[i]], bx[j], score_clip[maxidxs[i]][0], sj, rela)
        topscores = softmax0(torch.Tensor(topscores))
        a = torch.from_numpy(attn)
        a = (a - a.min()) / (a.max() - a.min())
        a = a * utils.gen_dir_mask(dirflag, H, W, None)
        a = a / a.mean()
        black = 1.95 if rela == "big" else (1.5 if rela == "small" else 1.8)
        gem = []
        for pm_ in torch.from_numpy(masks):
            pm_ = pm_.type(torch.uint8)
            gem.append(float((a * (2 - black) * pm_ / (pm_.sum())).sum() - (a * black * (1 - pm_) / ((1 - pm_).sum())).sum()))
        gem = np.array(gem, dtype=np.float32)
        for i in range(k1):
            topscores[i] = topscores[i] * (1 - 0.6) + 0.6 * float(gem[maxidxs[i]])
        idx_final = int(maxidxs[torch.argmax(topscores)])
        if info is not None:     # how decided the final arg-max was (fixture selection only)
            ts = torch.sort(topscores, descending=True).values
            info["final_margin"] = float(ts[0] - ts[1]) if len(ts) > 1 else float("inf")
        _, _, cI, cU = utils.Compute_IoU(torch.from_numpy(masks[idx_final]), torch.from_numpy(gt[None].astype(np.uint8)), 0, 0, [])
    return idx_pure, idx_final, (int(cI), int(cU)), k1, k2, gem


def gen_scoring_small():
    """Tail goldens with FEWER proposals than k1 = 3 / k2 = 6 (Hybridgl_main.py:178-181: the clamp, which the reference
    never undoes): a sequence of refs with N = 12, 5, 12, 2, 12, 2, 1, 12, 1 proposals run with the k1 / k2 carried from ref to ref,
    exactly as the reference's loop does."""
    sys.path.insert(0, REF)
    utils = _load("ref_utils", os.path.join(REF, "utils.py"))
    bb = build_ref_backbone("tiny", 0)
    out = {}
    plan = [(0, 12, "none", "none", False), (1, 5, "left", "left", True), (2, 12, "big", "middle", True),
            (3, 2, "within", "right", True), (4, 12, "small", "none", False), (5, 2, "none", "left", False),
            (6, 1, "left", "left", True), (7, 12, "within", "middle", True), (8, 1, "none", "none", False)]
    k1, k2 = 3, 6
    for step, (ci, N, rela, dirflag, has_other) in enumerate(plan):
        hybrid, t_pos, t_neg, masks, boxes, attn, gt = tail_case(ci, N, 32, 96, 128)
        ip, ifin, iu, k1, k2, gem = ref_tail(bb, utils, hybrid, t_pos, t_neg, masks, boxes, attn, gt, rela, dirflag, has_other, k1, k2)
        out[f"s{step}_idx"] = np.array([ip, ifin], dtype=np.int64)
        out[f"s{step}_IU"] = np.array(iu, dtype=np.int64)
        out[f"s{step}_k"] = np.array([k1, k2], dtype=np.int64)
        out[f"s{step}_gem"] = gem
        print("small tail", step, N, rela, dirflag, has_other, ip, ifin, k1, k2)
    out["plan"] = np.array([f"{ci},{N},{rela},{dirflag},{int(h)}" for ci, N, rela, dirflag, h in plan])
    np.savez_compressed(os.path.join(GOLD, "scoring_small.npz"), **out)


def gen_scoring_ties():
    """Tail goldens with EXACT score ties (duplicated proposals): pins which of the equal candidates torch.argmax /
    torch.topk (Hybridgl_main.py:163-183) return, and with them the winning index."""
    sys.path.insert(0, REF)
    utils = _load("ref_utils", os.path.join(REF, "utils.py"))
    bb = build_ref_backbone("tiny", 0)
    out = {}
    for step, (ci, dup, rela, dirflag, has_other) in enumerate(TIE_PLAN):
        hybrid, t_pos, t_neg, masks, boxes, attn, gt = tie_case(ci, dup)
        ip, ifin, iu, _, _, gem = ref_tail(bb, utils, hybrid, t_pos, t_neg, masks, boxes, attn, gt, rela, dirflag, has_other, 3, 6)
        out[f"t{step}_idx"] = np.array([ip, ifin], dtype=np.int64)
        out[f"t{step}_IU"] = np.array(iu, dtype=np.int64)
        print("tie tail", step, dup, rela, dirflag, has_other, ip, ifin)
    np.savez_compressed(os.path.join(GOLD, "scoring_ties.npz"), **out)


def gen_tail_glue():
    """The text glue in front of the tail, statement for statement (Hybridgl_main.py:146-165): sentence and noun phrase
    encoded separately and mixed with r, every other noun encoded and AVERAGED into one negative feature (zeros when the
    sentence has none), then the tail as in ref_tail.  Pins the device kernel's own ensemble / mean against the reference's."""
    sys.path.insert(0, REF)
    utils = _load("ref_utils", os.path.join(REF, "utils.py"))
    bb = build_ref_backbone("tiny", 0)
    r = 0.5
    out = {"r": np.array([r], dtype=np.float32)}
    for ci, n_other, rela, dirflag in GLUE_PLAN:
        hybrid, _, _, masks, boxes, attn, gt = tail_case(ci, 12, 32, 96, 128)
        tok = torch.from_numpy(glue_tokens(ci, n_other).astype(np.int64))
        with torch.no_grad():
            sentence_features = bb.model.encode_text(tok[0:1])
            noun_phrase_features = bb.model.encode_text(tok[1:2])
            text_ensemble = r * sentence_features + (1 - r) * noun_phrase_features
            other_noun_features = torch.zeros(1, sentence_features.shape[1])
            cnt = 0
            for j in range(n_other):
                other_noun_features += bb.model.encode_text(tok[2 + j:3 + j])
                cnt += 1
            if cnt != 0:
                other_noun_features = other_noun_features / cnt
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ip, ifin, iu, _, _, gem = ref_tail(bb, utils, hybrid, text_ensemble.numpy(), other_noun_features.numpy(), masks, boxes, attn,
                                                   gt, rela, dirflag, n_other > 0, 3, 6)
                sc = torch.nn.Softmax(0)(bb.calculate_score(torch.from_numpy(hybrid), text_ensemble)).numpy()
        out[f"g{ci}_idx"] = np.array([ip, ifin], dtype=np.int64)
        out[f"g{ci}_IU"] = np.array(iu, dtype=np.int64)
        out[f"g{ci}_score_clip"] = sc.astype(np.float32)
        out[f"g{ci}_ensemble"] = text_ensemble.numpy().astype(np.float32)
        out[f"g{ci}_other"] = other_noun_features.numpy().astype(np.float32)
        print("tail glue", ci, n_other, rela, dirflag, ip, ifin)
    np.savez_compressed(os.path.join(GOLD, "tail_glue.npz"), **out)


def gen_scoring_nan():
    """Tail goldens for the divisions by zero of Hybridgl_main.py:203-223 (constant heat-map, empty / full proposal masks,
    also as the best-scoring proposal): which index the reference reports when NaNs reach its soft-max / arg-max."""
    sys.path.insert(0, REF)
    utils = _load("ref_utils", os.path.join(REF, "utils.py"))
    bb = build_ref_backbone("tiny", 0)
    out = {}
    import warnings
    for step, (kind, rela, dirflag, has_other) in enumerate(NAN_PLAN):
        hybrid, t_pos, t_neg, masks, boxes, attn, gt = nan_case(kind)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ip, ifin, iu, _, _, gem = ref_tail(bb, utils, hybrid, t_pos, t_neg, masks, boxes, attn, gt, rela, dirflag, has_other, 3, 6)
        out[f"n{step}_idx"] = np.array([ip, ifin], dtype=np.int64)
        out[f"n{step}_IU"] = np.array(iu, dtype=np.int64)
        out[f"n{step}_gem_nan"] = np.isnan(gem)
        print("nan tail", step, kind, rela, dirflag, has_other, ip, ifin, int(np.isnan(gem).sum()))
    np.savez_compressed(os.path.join(GOLD, "scoring_nan.npz"), **out)


def gen_views():
    """Hybridgl_main.py:93-125 (the per-mask local / global view loop) with everything that CAN be pinned offline pinned:
    the loop's statements are kept one for one; cv2.bitwise_and / cv2.add are their documented uint8 semantics in numpy
    (masked copy, saturating add); T.ToTensor / T.Resize(antialias=None) / T.Normalize are torch arithmetic
    (torchvision 0.15 tensor path = F.interpolate bilinear, align_corners=False, no antialias).  Only the Gaussian blur
    itself is an INPUT here (cv2 is absent; its restatement is pinned separately as integer arithmetic, oracle/cv_oracle.py)."""
    from hybridgl_amd import synth
    from oracle import cv_oracle as CV
    out = {}
    for tag, (H, W, N, res) in {"a": (97, 130, 4, 64), "b": (120, 88, 3, 56)}.items():
        img = synth.synth_image(H, W, 900 + N)
        masks_np = edge_masks(N, H, W, 910 + N)
        blurred = CV.gaussian_blur_u8(img, 15)
        imagesrc = torch.from_numpy(img)[None]
        original_img = torch.from_numpy(synth.imagenet_normalize(img))[None]       # image['image'] (dataset_refer_bert.py:155)
        masks = torch.from_numpy(masks_np)
        pixel_mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).reshape(1, 3, 1, 1)
        to_tensor = lambda a: torch.from_numpy(a).permute(2, 0, 1).float().div(255)
        resize = lambda x: F.interpolate(x[None] if x.dim() == 3 else x, size=(res, res), mode="bilinear", align_corners=False)[0]
        normalize = lambda x: (x - torch.tensor([0.485, 0.456, 0.406])[:, None, None]) / torch.tensor([0.229, 0.224, 0.225])[:, None, None]
        global_imgs, local_imgs = [], []
        for pred_mask in masks:
            pred_mask = pred_mask.type(torch.uint8)
            global_img = imagesrc[0].numpy()
            mask = pred_mask.cpu().numpy()
            sharp_region = np.where(np.clip(mask, 0, 255).astype(np.uint8)[:, :, None] != 0, global_img, 0).astype(np.uint8)  # cv2.bitwise_and(img, img, mask=)
            inv_mask = 1 - mask
            blurred_region = (blurred * inv_mask[:, :, None]).astype(np.uint8)
            global_img = np.clip(sharp_region.astype(np.int32) + blurred_region.astype(np.int32), 0, 255).astype(np.uint8)  # cv2.add (saturating)
            global_img = normalize(resize(to_tensor(global_img)))
            global_imgs.append(global_img)
            masked_image = original_img * pred_mask[None, None, ...] + (1 - pred_mask[None, None, ...]) * pixel_mean
            masked_image = resize(masked_image.squeeze(0))
            local_imgs.append(masked_image.squeeze(0))
        out[f"{tag}_meta"] = np.array([H, W, N, res, 900 + N, 910 + N], dtype=np.int64)
        out[f"{tag}_global"] = torch.stack(global_imgs).numpy().astype(np.float32)
        out[f"{tag}_local"] = torch.stack(local_imgs).numpy().astype(np.float32)
        print("views", tag, out[f"{tag}_global"].shape)
    np.savez_compressed(os.path.join(GOLD, "views.npz"), **out)


def gen_resize():
    """bilinear no-antialias resize vs the real torch F.interpolate (mask down-sample and 224 views).
    Inputs are regenerated from the seed by the tests (oracle/cases.py:resize_case)."""
    out = {}
    for i in range(len(RESIZE_CASES)):
        x, (H, W, oh, ow) = resize_case(i)
        out[f"r{i}_out"] = F.interpolate(torch.from_numpy(x)[None], size=(oh, ow), mode="bilinear",
                                         align_corners=False)[0].numpy()
    np.savez_compressed(os.path.join(GOLD, "resize.npz"), **out)


def install_sam_stubs():
    """torchvision / cv2 entry points the segment_anything package imports (absent offline).
    batched_nms and connectedComponentsWithStats are OUR restatements -> NMS / CC parity stays
    unpinned; everything else below runs the reference's own code."""
    from oracle import sam_oracle as S
    from PIL import Image
    tv = sys.modules["torchvision"]
    ops = types.ModuleType("torchvision.ops")
    boxes = types.ModuleType("torchvision.ops.boxes")

    def batched_nms(b, s, idxs, iou_threshold):
        if b.numel() == 0:
            return torch.empty((0,), dtype=torch.int64)
        return torch.from_numpy(S.nms(b.numpy(), s.numpy(), iou_threshold))

    boxes.batched_nms = batched_nms
    boxes.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    ops.boxes = boxes
    tv.ops = ops
    sys.modules["torchvision.ops"] = ops
    sys.modules["torchvision.ops.boxes"] = boxes
    tvf = sys.modules["torchvision.transforms.functional"]
    tvf.to_pil_image = lambda a: Image.fromarray(a)
    tvf.resize = lambda img, size: img.resize((size[1], size[0]), Image.BILINEAR) if isinstance(img, Image.Image) else tvf._tensor_resize(img, size)
    cv2 = sys.modules["cv2"]

    def cc(mask, conn):
        from scipy import ndimage
        lab, n = ndimage.label(mask, structure=np.ones((3, 3), int))
        stats = np.zeros((n + 1, 5), dtype=np.int64)
        stats[:, -1] = np.bincount(lab.ravel(), minlength=n + 1)
        return n + 1, lab.astype(np.int32), stats, None

    cv2.connectedComponentsWithStats = cc


def build_ref_sam(cfg_name, seed):
    """The reference's Sam assembled as build_sam._build_sam does, at the geometry of
    weights.SAM_CONFIGS[cfg_name], with our seeded state_dict."""
    from functools import partial
    sys.path.insert(0, os.path.join(REF, "third_party/segment-anything"))
    import segment_anything  # noqa: F401
    from segment_anything.modeling import ImageEncoderViT, MaskDecoder, PromptEncoder, Sam, TwoWayTransformer
    cfg = weights.SAM_CONFIGS[cfg_name]
    g = cfg["img_size"] // cfg["patch_size"]
    sam = Sam(
        image_encoder=ImageEncoderViT(depth=cfg["depth"], embed_dim=cfg["embed_dim"], img_size=cfg["img_size"],
                                      mlp_ratio=4, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                                      num_heads=cfg["num_heads"], patch_size=cfg["patch_size"], qkv_bias=True,
                                      use_rel_pos=True, global_attn_indexes=cfg["global_attn_indexes"],
                                      window_size=cfg["window_size"], out_chans=256),
        prompt_encoder=PromptEncoder(embed_dim=256, image_embedding_size=(g, g),
                                     input_image_size=(cfg["img_size"], cfg["img_size"]), mask_in_chans=16),
        mask_decoder=MaskDecoder(num_multimask_outputs=3,
                                 transformer=TwoWayTransformer(depth=2, embedding_dim=256, mlp_dim=2048, num_heads=8),
                                 transformer_dim=256, iou_head_depth=3, iou_head_hidden_dim=256),
        pixel_mean=[123.675, 116.28, 103.53], pixel_std=[58.395, 57.12, 57.375])
    sd = weights.sam_state_dict(cfg_name, seed)
    sam.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    return sam.eval()


def gen_sam_tiny():
    """Stage tensors of the reference SAM at the tiny geometry."""
    from oracle.cases import sam_tiny_case
    sam = build_ref_sam("tiny", 0)
    from segment_anything import SamAutomaticMaskGenerator
    from segment_anything.utils import amg as ref_amg
    c = sam_tiny_case()
    out = {}
    with torch.no_grad():
        x = sam.preprocess(torch.from_numpy(c["resized"]).permute(2, 0, 1).float()[None])
        emb = sam.image_encoder(x)                                   # [1,256,16,16]
        out["emb_nhwc"] = emb[0].permute(1, 2, 0).numpy()[::2, ::2]
        # intermediate: first (windowed) and second (global) block outputs for bisecting
        t = sam.image_encoder.patch_embed(x) + sam.image_encoder.pos_embed
        t0 = sam.image_encoder.blocks[0](t)
        t1 = sam.image_encoder.blocks[1](t0)
        out["blk0"], out["blk1"] = t0[0].numpy()[::3, ::3], t1[0].numpy()[::3, ::3]
        pts = torch.as_tensor(c["points_in"])                         # float64 [P,2]
        lab = torch.ones(pts.shape[0], dtype=torch.int)
        sparse, dense = sam.prompt_encoder(points=(pts[:, None, :], lab[:, None]), boxes=None, masks=None)
        out["sparse"] = sparse.numpy()
        out["dense_pe"] = sam.prompt_encoder.get_dense_pe()[0].permute(1, 2, 0).reshape(-1, 256).numpy()[::5]
        low, iou = sam.mask_decoder(image_embeddings=emb, image_pe=sam.prompt_encoder.get_dense_pe(),
                                    sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense,
                                    multimask_output=True)
        out["low_res"], out["iou"] = low.numpy(), iou.numpy()
        full = sam.postprocess_masks(low, c["input_size"], c["orig_size"])
        out["full_logits"] = full.numpy()[:, :, ::4, ::4]
        out["stability"] = ref_amg.calculate_stability_score(full.flatten(0, 1), 0.0, 1.0).numpy()
        out["boxes"] = ref_amg.batched_mask_to_box(full.flatten(0, 1) > 0).numpy()
        rles = ref_amg.mask_to_rle_pytorch(full.flatten(0, 1)[:2] > 0)
        back = np.stack([ref_amg.rle_to_mask(r) for r in rles])
        assert np.array_equal(back, (full.flatten(0, 1)[:2] > 0).numpy())
        out["point_grid8"] = ref_amg.build_point_grid(8)
        # whole generator, thresholds relaxed so that random weights keep some masks
        gen = SamAutomaticMaskGenerator(sam, points_per_side=4, pred_iou_thresh=-1e9, stability_score_thresh=0.0,
                                        crop_n_layers=0, min_mask_region_area=20, box_nms_thresh=1.5)
        anns = gen.generate(c["image"])
        out["amg_n"] = np.array([len(anns)])
        if anns:
            out["amg_masks"] = np.packbits(np.stack([a["segmentation"] for a in anns]), axis=-1)
            out["amg_bbox"] = np.array([a["bbox"] for a in anns], dtype=np.int64)
            out["amg_iou"] = np.array([a["predicted_iou"] for a in anns], dtype=np.float32)
            out["amg_stab"] = np.array([a["stability_score"] for a in anns], dtype=np.float32)
            out["amg_points"] = np.array([a["point_coords"][0] for a in anns], dtype=np.float64)
            out["amg_area"] = np.array([a["area"] for a in anns], dtype=np.int64)
        print("sam_tiny: emb", out["emb_nhwc"].shape, "low", low.shape, "iou range", float(iou.min()), float(iou.max()),
              "logit range", float(full.min()), float(full.max()), "amg masks", len(anns))
        # the same generator with thresholds that DECIDE (Hybridgl_main.py:67-73 runs 0.7 / 0.7 / NMS 0.7 on trained weights;
        # with seeded weights the predicted-IoU and stability values live elsewhere, so the two thresholds are put at
        # quantiles of what the relaxed run produced; box_nms_thresh is the reference's 0.7): a 6 x 6 grid, every filter live
        # (mask threshold raised to a high logit quantile, as in gen_sam_crops: a mask is then a handful of pixels and the
        # boxes differ from mask to mask, so the NMS has real decisions to take)
        sam.mask_threshold = float(np.quantile(full.numpy(), 0.9995))
        out["dec_mask_threshold"] = np.array([sam.mask_threshold], dtype=np.float64)
        gen6 = SamAutomaticMaskGenerator(sam, points_per_side=6, pred_iou_thresh=-1e9, stability_score_thresh=0.0,
                                         stability_score_offset=0.25, crop_n_layers=0, min_mask_region_area=0, box_nms_thresh=1.5)
        a6 = gen6.generate(c["image"])
        iou_thr = float(np.quantile([a["predicted_iou"] for a in a6], 0.45))
        stab_thr = float(np.nanquantile([a["stability_score"] for a in a6], 0.35))
        gen_d = SamAutomaticMaskGenerator(sam, points_per_side=6, pred_iou_thresh=iou_thr, stability_score_thresh=stab_thr,
                                          stability_score_offset=0.25, crop_n_layers=0, min_mask_region_area=20, box_nms_thresh=0.7)
        ad = gen_d.generate(c["image"])
        out["dec_thr"] = np.array([iou_thr, stab_thr, 0.7], dtype=np.float64)
        out["dec_n_open"] = np.array([len(a6)])
        out["dec_masks"] = np.packbits(np.stack([a["segmentation"] for a in ad]), axis=-1)
        out["dec_bbox"] = np.array([a["bbox"] for a in ad], dtype=np.int64)
        out["dec_iou"] = np.array([a["predicted_iou"] for a in ad], dtype=np.float32)
        out["dec_stab"] = np.array([a["stability_score"] for a in ad], dtype=np.float32)
        out["dec_points"] = np.array([a["point_coords"][0] for a in ad], dtype=np.float64)
        sam.mask_threshold = 0.0
        print("sam_tiny deciding thresholds", out["dec_thr"], "kept", len(ad), "of", len(a6))
    np.savez_compressed(os.path.join(GOLD, "sam_tiny.npz"), **out)


TOKENIZER_STRINGS = [
    "the cat on left", "a photo of the bigger elephant", "Person in BLUE shirt, holding an umbrella!",
    "second banana from right", "woman's red hat isn't there", "they're we've i'm you'll he'd",
    "3 zebras   and 42\tgiraffes", "caf\u00e9 na\u00efve \u00fcber", "tom &amp; jerry &lt;3", "  leading and trailing  ",
    "the man standing inside the doorway near the larger window closest to the camera", "x", "",
    "emoji \U0001f600 ok", "under_score-dash/slash",
]


def gen_sam_prompts():
    """SamPredictor.predict_torch / predict of the reference (predictor.py:90-243) on the tiny geometry for the prompt kinds
    beyond the automatic generator's foreground points: labelled single points, boxes, multimask_output False."""
    from oracle.cases import sam_tiny_case, sam_prompts_case
    sam = build_ref_sam("tiny", 0)
    from segment_anything import SamPredictor
    c, q = sam_tiny_case(), sam_prompts_case()
    pred = SamPredictor(sam)
    out = {}
    with torch.no_grad():
        pred.set_image(c["image"])
        pts = pred.transform.apply_coords(q["points"], pred.original_size)
        bxs = pred.transform.apply_boxes(q["boxes"], pred.original_size)
        out["boxes_in"] = bxs
        for tag, kw in (("pts", dict(point_coords=torch.as_tensor(pts)[:, None, :], point_labels=torch.as_tensor(q["labels"])[:, None])),
                        ("box", dict(point_coords=None, point_labels=None, boxes=torch.as_tensor(bxs)))):
            for mm in (True, False):
                full, iou, low = pred.predict_torch(multimask_output=mm, return_logits=True, **kw)
                k = f"{tag}_{'multi' if mm else 'single'}"
                out[k + "_low"], out[k + "_iou"], out[k + "_full"] = low.numpy()[:, :, ::2, ::2], iou.numpy(), full.numpy()[:, :, ::8, ::8]
                print("sam_prompts", k, tuple(low.shape), "iou", iou.numpy().round(4).tolist()[:2])
        sparse, _ = sam.prompt_encoder(points=None, boxes=torch.as_tensor(bxs), masks=None)
        out["box_sparse"] = sparse.numpy()
        # three sparse tokens: two points (+ padding), a point and a box
        prs = pred.transform.apply_coords(q["pairs"], pred.original_size)
        full, iou, low = pred.predict_torch(torch.as_tensor(prs), torch.as_tensor(q["pair_labels"]), multimask_output=True,
                                            return_logits=True)
        out["pair_low"], out["pair_iou"] = low.numpy()[:, :, ::2, ::2], iou.numpy()
        full, iou, low = pred.predict_torch(torch.as_tensor(pts)[:, None, :], torch.as_tensor(q["labels"])[:, None],
                                            boxes=torch.as_tensor(bxs), multimask_output=False, return_logits=True)
        out["ptbox_low"], out["ptbox_iou"] = low.numpy()[:, :, ::2, ::2], iou.numpy()
        # more tokens: six points (+ padding: T = 12), three points and a box (T = 10)
        mny = pred.transform.apply_coords(q["many"], pred.original_size)
        full, iou, low = pred.predict_torch(torch.as_tensor(mny), torch.as_tensor(q["many_labels"]), multimask_output=True,
                                            return_logits=True)
        out["many_low"], out["many_iou"] = low.numpy()[:, :, ::2, ::2], iou.numpy()
        full, iou, low = pred.predict_torch(torch.as_tensor(mny)[:, :3], torch.as_tensor(q["many_labels"])[:, :3],
                                            boxes=torch.as_tensor(bxs)[:2], multimask_output=False, return_logits=True)
        out["manybox_low"], out["manybox_iou"] = low.numpy()[:, :, ::2, ::2], iou.numpy()
        # mask inputs: the single-mask logits of the box prompts fed back with the same boxes (predictor.py:106-110)
        _, _, low1 = pred.predict_torch(None, None, boxes=torch.as_tensor(bxs), multimask_output=False, return_logits=True)
        out["mask_in"] = low1.numpy()
        _, dense = sam.prompt_encoder(points=None, boxes=torch.as_tensor(bxs), masks=low1)
        out["mask_dense"] = dense.permute(0, 2, 3, 1).reshape(dense.shape[0], -1, dense.shape[1]).numpy()[:, ::7]
        full, iou, low = pred.predict_torch(None, None, boxes=torch.as_tensor(bxs), mask_input=low1, multimask_output=True,
                                            return_logits=True)
        out["maskin_low"], out["maskin_iou"], out["maskin_full"] = low.numpy()[:, :, ::2, ::2], iou.numpy(), full.numpy()[:, :, ::8, ::8]
        print("sam_prompts mask input: iou", iou.numpy().round(4).tolist()[:2])
        m, iou, low = pred.predict(point_coords=q["one_point"], point_labels=q["one_label"], multimask_output=True, return_logits=True)
        out["predict_pt_low"], out["predict_pt_iou"] = low, iou
        m, iou, low = pred.predict(box=q["one_box"], multimask_output=False, return_logits=True)
        out["predict_box_low"], out["predict_box_iou"], out["predict_box_full"] = low, iou, m[:, ::4, ::4]
        mb, _, _ = pred.predict(box=q["one_box"], multimask_output=False)
        out["predict_box_mask"] = np.packbits(mb, axis=-1)
        m, iou, low2 = pred.predict(point_coords=q["one_point"], point_labels=np.array([1]), box=q["one_box"], mask_input=low,
                                    multimask_output=True, return_logits=True)
        out["predict_all_low"], out["predict_all_iou"] = low2[:, ::2, ::2], iou
    np.savez_compressed(os.path.join(GOLD, "sam_prompts.npz"), **out)


def gen_sam_crops():
    """Crop-layer generator (automatic_mask_generator.py:197-267, amg.py:78-88,201-252) at the tiny geometry."""
    from oracle.cases import sam_crops_case
    sam = build_ref_sam("tiny", 0)
    from segment_anything import SamAutomaticMaskGenerator
    from segment_anything.utils import amg as ref_amg
    c = sam_crops_case()
    out = {}
    with torch.no_grad():
        # threshold: a high quantile of the full-image logits of the first 64 prompts
        probe = SamAutomaticMaskGenerator(sam, points_per_side=c["points_per_side"], pred_iou_thresh=-1e9,
                                          stability_score_thresh=0.0, box_nms_thresh=1.5)
        probe.predictor.set_image(c["image"])
        H, W = c["image"].shape[:2]
        pts = probe.point_grids[0] * np.array([[W, H]])
        tp = probe.predictor.transform.apply_coords(pts, (H, W))
        lg, _, _ = probe.predictor.predict_torch(torch.as_tensor(tp)[:, None, :], torch.ones(len(tp), 1, dtype=torch.int),
                                                 multimask_output=True, return_logits=True)
        thr = float(np.quantile(lg.numpy(), c["logit_quantile"]))
        sam.mask_threshold = thr
        out["mask_threshold"] = np.array([thr], np.float64)
        for tag, min_area in (("a", 0), ("b", 3)):
            gen = SamAutomaticMaskGenerator(sam, points_per_side=c["points_per_side"], pred_iou_thresh=-1e9,
                                            stability_score_thresh=0.0, box_nms_thresh=c["box_nms_thresh"],
                                            crop_n_layers=c["crop_n_layers"], crop_nms_thresh=c["crop_nms_thresh"],
                                            crop_n_points_downscale_factor=c["downscale"], min_mask_region_area=min_area)
            anns = gen.generate(c["image"])
            out[tag + "_n"] = np.array([len(anns)])
            out[tag + "_masks"] = np.packbits(np.stack([a["segmentation"] for a in anns]), axis=-1)
            out[tag + "_bbox"] = np.array([a["bbox"] for a in anns], dtype=np.int64)
            out[tag + "_iou"] = np.array([a["predicted_iou"] for a in anns], dtype=np.float32)
            out[tag + "_stab"] = np.array([a["stability_score"] for a in anns], dtype=np.float32)
            out[tag + "_points"] = np.array([a["point_coords"][0] for a in anns], dtype=np.float64)
            out[tag + "_area"] = np.array([a["area"] for a in anns], dtype=np.int64)
            out[tag + "_crop_box"] = np.array([a["crop_box"] for a in anns], dtype=np.int64)
            print("sam_crops", tag, "masks", len(anns), "from crops", sorted(set(map(tuple, out[tag + "_crop_box"].tolist()))))
        # helper known-answers
        cb, li = ref_amg.generate_crop_boxes((240, 320), 2, 512 / 1500)
        out["crop_boxes_240x320_l2"] = np.array(cb, dtype=np.int64)
        out["crop_layers_240x320_l2"] = np.array(li, dtype=np.int64)
        cb, li = ref_amg.generate_crop_boxes((640, 480), 1, 512 / 1500)
        out["crop_boxes_640x480_l1"] = np.array(cb, dtype=np.int64)
        grids = ref_amg.build_all_layer_point_grids(16, 2, 2)
        out["grid_sizes_16_2_2"] = np.array([len(gx) for gx in grids])
        out["grid_l2_16_2_2"] = grids[2]
        rng = np.random.default_rng(3)
        bx = rng.integers(0, 100, size=(200, 4))
        bx[:, 2:] = np.minimum(bx[:, :2] + rng.integers(0, 60, size=(200, 2)), 127)
        crop = [40, 30, 167, 137]
        near = ref_amg.is_box_near_crop_edge(torch.from_numpy(bx), crop, [0, 0, 320, 240])
        out["edge_boxes"], out["edge_crop"], out["edge_near"] = bx.astype(np.int64), np.array(crop), near.numpy()
    np.savez_compressed(os.path.join(GOLD, "sam_crops.npz"), **out)


def gen_tokenizer():
    """SimpleTokenizer / clip.tokenize of the reference on (1) our tiny synthetic merges file and
    (2) the real merges file that sits in the reference tree (token ids are data, the file is not copied)."""
    st = _load("ref_simple_tokenizer", os.path.join(REF, "third_party/modified_CLIP/clip/simple_tokenizer.py"))
    out = {"strings": np.array(TOKENIZER_STRINGS)}
    tiny = st.SimpleTokenizer(os.path.join(GOLD, "tiny_bpe_vocab.txt.gz"))
    real = st.SimpleTokenizer(os.path.join(REF, "third_party/modified_CLIP/clip/bpe_simple_vocab_16e6.txt.gz"))
    for tag, tk in (("tiny", tiny), ("real", real)):
        ids = [tk.encode(s) for s in TOKENIZER_STRINGS]
        out[f"{tag}_len"] = np.array([len(i) for i in ids], dtype=np.int64)
        out[f"{tag}_ids"] = np.array(sum(ids, []), dtype=np.int64)
        out[f"{tag}_sot_eot"] = np.array([tk.encoder["<|startoftext|>"], tk.encoder["<|endoftext|>"]], dtype=np.int64)
        out[f"{tag}_decoded0"] = np.array(tk.decode(ids[2]))
    print("tokenizer:", out["real_ids"][:6], out["real_sot_eot"], out["tiny_sot_eot"])
    np.savez_compressed(os.path.join(GOLD, "tokenizer.npz"), **out)


from oracle.gen_cases_e2e import E2E_CASES  # noqa: E402


def ref_views(img, blurred, masks_np, res):
    """Hybridgl_main.py:93-125, statement for statement (see gen_views for what stands in for cv2 / torchvision)."""
    from hybridgl_amd import synth
    imagesrc = torch.from_numpy(img)[None]
    original_img = torch.from_numpy(synth.imagenet_normalize(img))[None]       # image['image'] (dataset_refer_bert.py:155)
    masks = torch.from_numpy(masks_np)
    pixel_mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).reshape(1, 3, 1, 1)
    to_tensor = lambda a: torch.from_numpy(a).permute(2, 0, 1).float().div(255)
    resize = lambda x: F.interpolate(x[None] if x.dim() == 3 else x, size=(res, res), mode="bilinear", align_corners=False)[0]
    normalize = lambda x: (x - torch.tensor([0.485, 0.456, 0.406])[:, None, None]) / torch.tensor([0.229, 0.224, 0.225])[:, None, None]
    global_imgs, local_imgs = [], []
    for pred_mask in masks:
        pred_mask = pred_mask.type(torch.uint8)
        global_img = imagesrc[0].numpy()
        mask = pred_mask.cpu().numpy()
        sharp_region = np.where(np.clip(mask, 0, 255).astype(np.uint8)[:, :, None] != 0, global_img, 0).astype(np.uint8)  # cv2.bitwise_and(img, img, mask=)
        inv_mask = 1 - mask
        blurred_region = (blurred * inv_mask[:, :, None]).astype(np.uint8)
        global_img = np.clip(sharp_region.astype(np.int32) + blurred_region.astype(np.int32), 0, 255).astype(np.uint8)  # cv2.add (saturating)
        global_img = normalize(resize(to_tensor(global_img)))
        global_imgs.append(global_img)
        masked_image = original_img * pred_mask[None, None, ...] + (1 - pred_mask[None, None, ...]) * pixel_mean
        masked_image = resize(masked_image.squeeze(0))
        local_imgs.append(masked_image.squeeze(0))
    return torch.stack(local_imgs, dim=0), torch.stack(global_imgs, dim=0)


def gen_e2e_tiny():
    """WHOLE refs through the reference, stage into stage (Hybridgl_main.py:85-230): the reference's
    SamAutomaticMaskGenerator.generate (tiny SAM) -> masks / XYWH boxes exactly as :86-90 build them -> the view loop
    (:93-125) -> the reference's CLIPViTFM.forward (tiny CLIP, :128) -> encode_text + the text glue (:146-165) -> the tail
    (:153-230, ref_tail) with a seeded heat-map in the place of GEM's -> per sentence (idx_pure, idx_final, I, U).  The k1 / k2
    clamp is carried from ref to ref as the reference's loop does.  What this pins are the JOINS between the stages (bbox
    format into relation_boxes, the order of the masks after the second NMS, bool / uint8 conventions, which tensors the
    views are cut from); every stage alone has its own fixture.  Unpinned inputs, as everywhere: the blur taps (cv_oracle),
    NMS / connected components (stubs = our restatements), the heat-map (given)."""
    from hybridgl_amd import synth
    from oracle import cv_oracle as CV
    sys.path.insert(0, REF)
    utils = _load("ref_utils", os.path.join(REF, "utils.py"))
    bb = build_ref_backbone("tiny", 0)
    sam = build_ref_sam("tiny", 0)
    from segment_anything import SamAutomaticMaskGenerator
    import warnings
    out = {}
    # Random weights give noise logits.  The mask threshold sits at their 97 % quantile: the candidates then differ a lot
    # (areas from 0 to a sixth of the image), their hybrid features with them (logit margins of 1e-2 .. 1e-1 between the best
    # proposals, printed below), and few pixels lie near the threshold (0.5 per mask within 1e-4).  box_nms_thresh is opened
    # (1.5): the NMS then only ORDERS the candidates (by predicted IoU; with 0.7 the large ones suppress each other and what
    # is left are empty masks whose features coincide) -- deciding thresholds have their own fixture (sam_tiny.npz: dec_*).
    # measured once on the first image and stored
    with torch.no_grad():
        probe = SamAutomaticMaskGenerator(sam, points_per_side=6, pred_iou_thresh=-1e9, stability_score_thresh=0.0, box_nms_thresh=1.5)
        img0 = synth.synth_image(E2E_CASES[0][1], E2E_CASES[0][2], E2E_CASES[0][0])
        probe.predictor.set_image(img0)
        pts = probe.point_grids[0] * np.array([[img0.shape[1], img0.shape[0]]])
        tp = probe.predictor.transform.apply_coords(pts, img0.shape[:2])
        lg, _, _ = probe.predictor.predict_torch(torch.as_tensor(tp)[:, None, :], torch.ones(len(tp), 1, dtype=torch.int),
                                                 multimask_output=True, return_logits=True)
        thr = float(np.quantile(lg.numpy(), 0.97))
    sam.mask_threshold = thr
    out["mask_threshold"] = np.array([thr], np.float64)
    out["amg"] = np.array([6, -1e9, 0.0, 1.5, 6], dtype=np.float64)   # points_per_side, pred_iou, stability, box_nms, min_area
    gen = SamAutomaticMaskGenerator(sam, points_per_side=6, pred_iou_thresh=-1e9, stability_score_thresh=0.0,
                                    box_nms_thresh=1.5, crop_n_layers=0, crop_n_points_downscale_factor=1, min_mask_region_area=6)
    r = 0.5
    k1, k2 = 3, 6
    for ci, (iseed, H, W, tseed, sents, gseed) in enumerate(E2E_CASES):
        for mode in ("G2L", "G2L&L2G"):
            if mode != "G2L" and ci != 0:
                continue
            tag = f"c{ci}_{mode.replace('&', '_')}"
            kk1, kk2 = (k1, k2) if mode == "G2L" else (3, 6)
            with torch.no_grad(), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                sam_img = synth.synth_image(H, W, iseed)
                sam_masks = gen.generate(sam_img)                                            # :85
                masks = torch.stack([torch.tensor(m["segmentation"]) for m in sam_masks])   # :86-87
                boxes = torch.tensor([m["bbox"] for m in sam_masks])                         # :89-90
                blurred = CV.gaussian_blur_u8(sam_img, 15)                                   # :99 (restated taps)
                local_imgs, global_imgs = ref_views(sam_img, blurred, masks.numpy(), 64)
                hybrid = bb(local_imgs=local_imgs, global_imgs=global_imgs, pred_masks=masks, fusion_mode=mode, masking_block=9)  # :128
                n_rows = sum(2 + n for _, _, n in sents)
                gt = synth.synth_masks(1, H, W, gseed)[0]
                # the strings are seeded tokens; the seed is advanced until both decisions of every sentence are clear of
                # the device's error budget (raw logits 4e-3 apart, blended top-k scores 1e-4 apart) -- the chosen seed is stored
                for tries in range(200):
                    tok = synth.synth_tokens(n_rows, 16, 512, tseed + 1000 * tries)
                    row = 0
                    res_idx, res_iu, margins = [], [], []
                    c1, c2 = kk1_in, kk2_in = kk1, kk2
                    for j, (dirflag, rela, n_other) in enumerate(sents):
                        t = torch.from_numpy(tok[row:row + 2 + n_other].astype(np.int64))
                        row += 2 + n_other
                        sentence_features = bb.model.encode_text(t[0:1])
                        noun_phrase_features = bb.model.encode_text(t[1:2])
                        text_ensemble = r * sentence_features + (1 - r) * noun_phrase_features
                        other = torch.zeros(1, sentence_features.shape[1])
                        for q in range(n_other):
                            other += bb.model.encode_text(t[2 + q:3 + q])
                        if n_other:
                            other = other / n_other
                        attn = synth.synth_heatmap(H, W, 6000 + 10 * ci + j)
                        info = {}
                        ip, ifin, iu_f, c1, c2, gem_s = ref_tail(bb, utils, hybrid.numpy(), text_ensemble.numpy(), other.numpy(),
                                                                masks.numpy(), boxes.numpy(), attn, gt, rela, dirflag, n_other > 0, c1, c2, info)
                        _, _, cI, cU = utils.Compute_IoU(masks[ip], torch.from_numpy(gt[None].astype(np.uint8)), 0, 0, [])
                        res_idx.append([ip, ifin])
                        res_iu.append([int(cI), int(cU), iu_f[0], iu_f[1]])
                        sc = bb.calculate_score(hybrid, text_ensemble).view(-1)
                        top2 = torch.topk(sc, min(2, len(sc))).values
                        margins.append([float(top2[0] - top2[-1]), info["final_margin"]])
                    if all(a >= 4e-3 and b >= 1e-4 for a, b in margins):
                        break
                else:
                    raise RuntimeError(f"e2e {tag}: no text seed with clear decisions")
                kk1, kk2 = c1, c2
                for j in range(len(sents)):
                    print(f"  e2e {tag} sentence {j}: idx {res_idx[j]} IU {res_iu[j]} margins {margins[j][0]:.4f} / {margins[j][1]:.2e} (text seed {tseed + 1000 * tries})")
                out[f"{tag}_text_seed"] = np.array([tseed + 1000 * tries], dtype=np.int64)
                if mode == "G2L":
                    k1, k2 = kk1, kk2
            out[f"{tag}_masks"] = np.packbits(masks.numpy(), axis=-1)
            out[f"{tag}_boxes"] = boxes.numpy().astype(np.int64)
            out[f"{tag}_hybrid"] = hybrid.numpy().astype(np.float32)
            out[f"{tag}_idx"] = np.array(res_idx, dtype=np.int64)
            out[f"{tag}_IU"] = np.array(res_iu, dtype=np.int64)
            out[f"{tag}_k"] = np.array([kk1, kk2], dtype=np.int64)
            out[f"{tag}_margin"] = np.array(margins, dtype=np.float64)
            print("e2e", tag, "masks", tuple(masks.shape), "k after", kk1, kk2)
    np.savez_compressed(os.path.join(GOLD, "e2e_tiny.npz"), **out)


def state_dict_digest(sd):
    """order-independent digest of a state_dict: key names, dtypes, shapes and raw bytes"""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(sd):
        v = sd[k].detach().cpu().contiguous()
        h.update(k.encode())
        h.update(str(v.dtype).encode())
        h.update(str(tuple(v.shape)).encode())
        h.update(v.numpy().tobytes() if v.dtype != torch.bfloat16 else v.view(torch.int16).numpy().tobytes())
    return h.hexdigest()


def gen_ckpt():
    """Checkpoint files as the reference's loaders meet them, for the loaders of this package.  The files themselves are
    30 MB, so they are NOT stored: tests/ckpt_files.py rebuilds them from the seeded weights, and this fixture holds what
    makes that rebuild a reference-produced file -- the digest of the reference's own state_dicts and the list of tensors
    its convert_weights stores as fp16 -- plus the reference's outputs with those weights:
    * CLIP: the OpenAI archives hold fp16 weights (the reference's convert_weights, clip/model.py:434-459, says which);
      clip/clip.py:119-142 hands their state_dict -- incl. the three scalar entries -- to build_model
      (clip/model.py:474-511), which infers the geometry from the key set and leaves the weights up-cast to fp32
      because :509 is commented out.  Stored: outputs of build_model(that state_dict).
    * SAM: build_sam.py:103-106 torch.load()s a plain fp32 state_dict."""
    m = ref_clip_model_module()
    model = build_ref_clip("tiny", 0)
    m.convert_weights(model)                              # fp16 storage, as the published archives
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    cfg = weights.CLIP_CONFIGS["tiny"]
    sd["input_resolution"] = torch.tensor(cfg["image_resolution"])
    sd["context_length"] = torch.tensor(cfg["context_length"])
    sd["vocab_size"] = torch.tensor(cfg["vocab_size"])
    out = {"clip_fp16_keys": np.array(sorted(k for k, v in sd.items() if v.dtype == torch.float16)),
           "clip_digest": np.array(state_dict_digest(sd)),
           "clip_keys": np.array(sorted(sd))}
    # the reference's loader on that state_dict: build_model (clip/model.py:474-511) -> fp32 model
    ref = m.build_model({k: v.clone() for k, v in sd.items()}).float().eval()
    tok = torch.from_numpy(glue_tokens(31, 2).astype(np.int64))
    with torch.no_grad():
        out["text"] = ref.encode_text(tok).numpy().astype(np.float32)
    out["text_tokens_case"] = np.array([31, 2])
    clip_stub = types.ModuleType("clip")
    clip_stub.load = lambda name, *a, **k: (ref, None)
    sys.modules["clip"] = clip_stub
    bb = _load("ref_backbone_ckpt", os.path.join(REF, "model/backbone.py")).CLIPViTFM(model_name="ViT-B/16").eval()
    bb.num_heads = cfg["vision_width"] // 64
    loc, glo, masks = views_for_case(3, 64, 97, 130)
    with torch.no_grad():
        out["hybrid_G2L"] = bb(local_imgs=torch.from_numpy(loc), global_imgs=torch.from_numpy(glo), pred_masks=torch.from_numpy(masks),
                               fusion_mode="G2L", masking_block=9).numpy().astype(np.float32)
    # SAM: the reference module's own state_dict (what a released .pth holds)
    install_sam_stubs()
    sam = build_ref_sam("tiny", 0)
    ssd = sam.state_dict()
    out["sam_digest"] = np.array(state_dict_digest(ssd))
    out["sam_keys"] = np.array(sorted(ssd))
    print("ckpt: clip", len(sd), "entries,", len(out["clip_fp16_keys"]), "fp16; sam", len(ssd), "entries")
    np.savez_compressed(os.path.join(GOLD, "ckpt.npz"), **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sel = set(args.only.split(",")) if args.only else None
    want = lambda k: sel is None or k in sel
    if want("clip_tiny"):
        gen_clip("tiny", 0, [1, 3, 5], 97, 130, MODES, "clip_tiny")
    if want("clip_tiny_mb"):
        gen_clip_mb()
    if want("clip_b16"):
        gen_clip("ViT-B/16", 0, [4], 640, 640, ["G2L", "L2G", "G2L&L2G"], "clip_b16")
    if want("text_tiny"):
        gen_text("tiny", 0, "text_tiny")
    if want("text_b16"):
        gen_text("ViT-B/16", 0, "text_b16")
    if want("views"):
        gen_views()
    if want("scoring_small"):
        gen_scoring_small()
    if want("scoring_ties"):
        gen_scoring_ties()
    if want("scoring_nan"):
        gen_scoring_nan()
    if want("tail_glue"):
        gen_tail_glue()
    if want("text_pool_tiny"):
        gen_text_pool("tiny", 0, "text_pool_tiny", [1, 2])
    if want("text_pool_b16"):
        gen_text_pool("ViT-B/16", 0, "text_pool_b16", [11, 9])
    if want("scoring"):
        gen_scoring()
    if want("resize"):
        gen_resize()
    if want("sam_prompts"):
        install_sam_stubs()
        gen_sam_prompts()
    if want("tokenizer"):
        gen_tokenizer()
    if want("sam_tiny"):
        install_sam_stubs()
        gen_sam_tiny()
    if want("sam_crops"):
        install_sam_stubs()
        gen_sam_crops()
    if want("e2e_tiny"):
        install_sam_stubs()
        gen_e2e_tiny()
    if want("ckpt"):
        gen_ckpt()
