"""CPU: the numpy oracle against fixtures captured from the reference itself
(oracle/gen_golden.py, run in the build container with /root/reference imported read-only)."""
import os

import numpy as np
import pytest

from hybridgl_amd import weights
from oracle import clip_oracle as O
from oracle.cases import RESIZE_CASES, TAIL_CASES, resize_case, tail_case, views_for_case


def _load(golden_dir, name):
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated")
    return np.load(path)


def _unpack(bits, W):
    return np.unpackbits(bits, axis=-1)[..., :W].astype(bool)


MODES = ["G2L", "L2G", "G2L&L2G", "token_masking", "attn_masking", "crop"]


@pytest.mark.parametrize("N", [1, 3, 5])
@pytest.mark.parametrize("mode", MODES)
def test_clip_tiny_all_modes(golden_dir, N, mode):
    g = _load(golden_dir, "clip_tiny.npz")
    seed, H, W = (int(v) for v in g["meta"])
    sd = weights.clip_state_dict("tiny", seed)
    loc, glo, masks = views_for_case(N, 64, H, W)
    assert np.array_equal(masks, _unpack(g[f"N{N}_masks"], W))  # the seeded inputs are the fixture's
    y = O.clip_hybrid_forward(sd, loc, glo, masks, masking_block=9, fusion_mode=mode, last_layer=10)
    ref = g[f"N{N}_{mode}"]
    assert y.shape == ref.shape
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-5)


def test_clip_tiny_masking_block_range(golden_dir):
    """tests/golden/clip_tiny_mb.npz: the reference's forward with masking_block None (= last_layer), 0, 1, last_layer and
    last_layer + 1 in every mode in which it returns features"""
    g = _load(golden_dir, "clip_tiny_mb.npz")
    last = int(g["last_layer"][0])
    sd = weights.clip_state_dict("tiny", 0)
    loc, glo, masks = views_for_case(3, 64, 97, 130)
    keys = [k for k in g.files if k.startswith("mb")]
    assert len(keys) == 29
    for k in keys:
        mb, mode = k[2:].split("_", 1)
        mb = last if mb == "None" else int(mb)
        y = O.clip_hybrid_forward(sd, loc, glo, masks, masking_block=mb, fusion_mode=mode, last_layer=last)
        np.testing.assert_allclose(y, g[k], rtol=0, atol=2e-5, err_msg=k)


@pytest.mark.parametrize("mode", ["G2L", "L2G", "G2L&L2G"])
def test_clip_b16(golden_dir, mode):
    g = _load(golden_dir, "clip_b16.npz")
    seed, H, W = (int(v) for v in g["meta"])
    sd = weights.clip_state_dict("ViT-B/16", seed)
    loc, glo, masks = views_for_case(4, 224, H, W)
    y = O.clip_hybrid_forward(sd, loc, glo, masks, masking_block=9, fusion_mode=mode, last_layer=10)
    np.testing.assert_allclose(y, g[f"N4_{mode}"], rtol=0, atol=5e-5)


@pytest.mark.parametrize("name,cfg", [("text_tiny.npz", "tiny"), ("text_b16.npz", "ViT-B/16")])
def test_encode_text(golden_dir, name, cfg):
    g = _load(golden_dir, name)
    sd = weights.clip_state_dict(cfg, int(g["meta"][0]))
    y = O.encode_text(sd, g["tokens"], heads=weights.CLIP_CONFIGS[cfg]["transformer_heads"])
    np.testing.assert_allclose(y, g["out"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("name,cfg", [("text_pool_tiny.npz", "tiny"), ("text_pool_b16.npz", "ViT-B/16")])
def test_encode_text_pooling_and_token_masking(golden_dir, name, cfg):
    """CLIP.encode_text(text, target_noun_index) (clip/model.py:426-428) and CLIPViTFM.text_masking_feature
    (model/backbone.py:34-56) of the reference vs the oracle's restatement"""
    g = _load(golden_dir, name)
    sd = weights.clip_state_dict(cfg, 0)
    heads = weights.CLIP_CONFIGS[cfg]["transformer_heads"]
    for k in (0, 1, 3):
        y = O.encode_text(sd, g["tokens"], heads=heads, target_noun_index=k)
        np.testing.assert_allclose(y, g[f"pool_{k}"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(g["pool_0"], O.encode_text(sd, g["tokens"], heads=heads), rtol=0, atol=2e-5)   # 0 -> EOT
    for mb in g["masking_blocks"]:
        for tag, idx in (("idx12", [1, 2]), ("idx0", [0])):
            y = O.encode_text(sd, g["tokens"], heads=heads, masking_index=idx, masking_block=int(mb))
            np.testing.assert_allclose(y, g[f"mask_{int(mb)}_{tag}"], rtol=0, atol=2e-5)
    assert np.abs(g[f"mask_{int(g['masking_blocks'][0])}_idx12"] - g["mask_none"]).max() > 1e-3    # the masking does something


@pytest.mark.parametrize("i", range(len(RESIZE_CASES)))
def test_bilinear_resize(golden_dir, i):
    g = _load(golden_dir, "resize.npz")
    x, (H, W, oh, ow) = resize_case(i)
    y = O.bilinear_resize(x, oh, ow)
    np.testing.assert_allclose(y, g[f"r{i}_out"], rtol=0, atol=1e-6)


def test_calculate_score(golden_dir):
    g = _load(golden_dir, "scoring.npz")
    y = O.calculate_score(g["cs_img"], g["cs_txt"], float(g["cs_logit_scale"]))
    np.testing.assert_allclose(y, g["cs_out"], rtol=0, atol=2e-5)
    assert abs(float(g["cs_logit_scale"]) - 1 / 0.07) < 1e-4


def test_relation_boxes_table(golden_dir):
    g = _load(golden_dir, "scoring.npz")
    boxes, scores, tab = g["rb_boxes"], g["rb_scores"], g["rb_table"]
    for w, word in enumerate(g["rb_words"]):
        for i in range(4):
            for j in range(4):
                got = O.relation_boxes(boxes[i], boxes[j], scores[i], scores[j], str(word))
                assert abs(float(got) - float(tab[w, i, j])) <= 1e-7, (word, i, j)
    # SURVEY.md known answers: left:[0,.15,.10] big:[0,.15,0] within:[.25,0,.10] for box 0
    words = [str(w) for w in g["rb_words"]]
    np.testing.assert_allclose(tab[words.index("left"), 0, :3], [0, .15, .10], atol=1e-7)
    np.testing.assert_allclose(tab[words.index("big"), 0, :3], [0, .15, 0], atol=1e-7)
    np.testing.assert_allclose(tab[words.index("within"), 0, :3], [.25, 0, .10], atol=1e-7)


def test_gen_dir_mask(golden_dir):
    g = _load(golden_dir, "scoring.npz")
    for flag in ["left", "right", "middle", "none", "up"]:
        for (h, w) in [(3, 5), (4, 8), (2, 640), (2, 427)]:
            np.testing.assert_array_equal(O.gen_dir_mask(flag, h, w), g[f"dm_{flag}_{h}_{w}"])
    np.testing.assert_allclose(O.gen_dir_mask("left", 1, 5)[0], [1, .75, .5, .25, 0])
    np.testing.assert_allclose(O.gen_dir_mask("middle", 1, 5)[0], [0, 1, 1, .5, 0])


def test_compute_iou(golden_dir):
    g = _load(golden_dir, "scoring.npz")
    assert O.compute_iou(g["iou_pred"], g["iou_gt"]) == tuple(int(v) for v in g["iou_IU"])
    p = np.zeros((4, 4), bool); p[:2] = True
    t = np.zeros((4, 4), bool); t[1:] = True
    assert O.compute_iou(p, t) == (4, 16)
    assert O.compute_iou(np.zeros((3, 3), bool), np.zeros((3, 3), bool)) == (0, 0)


@pytest.mark.parametrize("ci", range(len(TAIL_CASES)))
def test_scoring_tail(golden_dir, ci):
    g = _load(golden_dir, "scoring.npz")
    H, W, N = (int(v) for v in g["tail_hw"])
    rela, dirflag, has_other = TAIL_CASES[ci]
    assert str(g["tail_cases"][ci]) == f"{rela},{dirflag},{int(has_other)}"
    hybrid, t_pos, t_neg, masks, boxes, attn, gt = tail_case(ci, N, 32, H, W)
    black = 1.95 if rela == "big" else (1.5 if rela == "small" else 1.8)
    gem = O.coherence_scores(attn, masks, dirflag, black)
    np.testing.assert_allclose(gem, g[f"tail{ci}_gem"], rtol=2e-5, atol=2e-5)
    ip, ifin, sc, sn = O.score_sentence(hybrid, t_pos, t_neg, boxes, gem, float(g["cs_logit_scale"]), 3, 6,
                                        0.6, rela, has_other)
    np.testing.assert_allclose(sc, g[f"tail{ci}_sc"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(sn, g[f"tail{ci}_sn"], rtol=0, atol=2e-5)
    assert [ip, ifin] == [int(v) for v in g[f"tail{ci}_idx"]]
    assert O.compute_iou(masks[ifin], gt) == tuple(int(v) for v in g[f"tail{ci}_IU"])


def test_scoring_tail_with_fewer_proposals_than_k(golden_dir):
    """refs with 5 and 2 proposals clamp k2 / k1 (Hybridgl_main.py:178-181) and the clamp persists to the later refs:
    the reference's winners for the whole sequence vs the oracle run with the carried k1 / k2"""
    g = _load(golden_dir, "scoring_small.npz")
    k1, k2 = 3, 6
    for step, rec in enumerate(g["plan"]):
        ci, N, rela, dirflag, has_other = str(rec).split(",")
        ci, N, has_other = int(ci), int(N), bool(int(has_other))
        hybrid, t_pos, t_neg, masks, boxes, attn, gt = tail_case(ci, N, 32, 96, 128)
        black = 1.95 if rela == "big" else (1.5 if rela == "small" else 1.8)
        gem = O.coherence_scores(attn, masks, dirflag, black)
        np.testing.assert_allclose(gem, g[f"s{step}_gem"], rtol=2e-5, atol=2e-5)
        k1, k2 = min(k1, N), min(k2, N)
        assert [k1, k2] == [int(v) for v in g[f"s{step}_k"]]
        ip, ifin, _, _ = O.score_sentence(hybrid, t_pos, t_neg, boxes, gem, float(_load(golden_dir, "scoring.npz")["cs_logit_scale"]),
                                          k1, k2, 0.6, rela, has_other)
        assert [ip, ifin] == [int(v) for v in g[f"s{step}_idx"]], step
        assert O.compute_iou(masks[ifin], gt) == tuple(int(v) for v in g[f"s{step}_IU"])


def test_scoring_tail_with_exact_ties(golden_dir):
    """tests/golden/scoring_ties.npz: the reference's tail on refs whose proposals contain exact copies (2-4 identical
    feature rows / masks / boxes): pins which of the equal candidates torch.argmax / torch.topk return"""
    from oracle.cases import TIE_PLAN, tie_case
    g = _load(golden_dir, "scoring_ties.npz")
    ls = float(_load(golden_dir, "scoring.npz")["cs_logit_scale"])
    for step, (ci, dup, rela, dirflag, has_other) in enumerate(TIE_PLAN):
        hybrid, t_pos, t_neg, masks, boxes, attn, gt = tie_case(ci, dup)
        black = 1.95 if rela == "big" else (1.5 if rela == "small" else 1.8)
        gem = O.coherence_scores(attn, masks, dirflag, black)
        ip, ifin, _, _ = O.score_sentence(hybrid, t_pos, t_neg, boxes, gem, ls, 3, 6, 0.6, rela, has_other)
        assert [ip, ifin] == [int(v) for v in g[f"t{step}_idx"]], step
        assert O.compute_iou(masks[ifin], gt) == tuple(int(v) for v in g[f"t{step}_IU"])


def test_tail_text_glue(golden_dir):
    """tests/golden/tail_glue.npz: the statements in front of the tail (Hybridgl_main.py:146-165) run by the reference --
    r * sentence + (1 - r) * noun phrase, the mean of 0..3 other-noun features -- then its tail: the oracle's text encoder +
    the same glue + score_sentence"""
    import warnings
    from oracle.cases import GLUE_PLAN, glue_tokens
    g = _load(golden_dir, "tail_glue.npz")
    r = float(g["r"][0])
    sd = weights.clip_state_dict("tiny", 0)
    ls = float(np.exp(sd["logit_scale"]))
    for ci, n_other, rela, dirflag in GLUE_PLAN:
        hybrid, _, _, masks, boxes, attn, gt = tail_case(ci, 12, 32, 96, 128)
        feats = O.encode_text(sd, glue_tokens(ci, n_other))
        ens = (np.float32(r) * feats[0:1] + np.float32(1 - r) * feats[1:2]).astype(np.float32)
        other = np.zeros((1, feats.shape[1]), dtype=np.float32)
        for j in range(n_other):
            other = other + feats[2 + j:3 + j]
        if n_other:
            other = other / np.float32(n_other)
        np.testing.assert_allclose(ens, g[f"g{ci}_ensemble"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(other, g[f"g{ci}_other"], rtol=0, atol=2e-5)
        black = 1.95 if rela == "big" else (1.5 if rela == "small" else 1.8)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            gem = O.coherence_scores(attn, masks, dirflag, black)
            ip, ifin, sc, _ = O.score_sentence(hybrid, ens, other, boxes, gem, ls, 3, 6, 0.6, rela, n_other > 0)
        np.testing.assert_allclose(O.softmax(sc, 0), g[f"g{ci}_score_clip"][:, 0], rtol=0, atol=2e-6)   # sc: the logits
        assert [ip, ifin] == [int(v) for v in g[f"g{ci}_idx"]], ci
        assert O.compute_iou(masks[ifin], gt) == tuple(int(v) for v in g[f"g{ci}_IU"])


def test_scoring_tail_divisions_by_zero(golden_dir):
    """tests/golden/scoring_nan.npz: constant heat-map (0/0 in the min-max), empty and full proposal masks (x/0 in the
    coherence terms), also as the best-scoring proposal: the reference's winners when NaNs reach its arg-max"""
    import warnings
    from oracle.cases import NAN_PLAN, nan_case
    g = _load(golden_dir, "scoring_nan.npz")
    ls = float(_load(golden_dir, "scoring.npz")["cs_logit_scale"])
    for step, (kind, rela, dirflag, has_other) in enumerate(NAN_PLAN):
        hybrid, t_pos, t_neg, masks, boxes, attn, gt = nan_case(kind)
        black = 1.95 if rela == "big" else (1.5 if rela == "small" else 1.8)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            gem = O.coherence_scores(attn, masks, dirflag, black)
            ip, ifin, _, _ = O.score_sentence(hybrid, t_pos, t_neg, boxes, gem, ls, 3, 6, 0.6, rela, has_other)
        assert np.array_equal(np.isnan(gem), g[f"n{step}_gem_nan"]), (step, kind)
        if kind.startswith("nan_row"):      # NaN features: only the pure-CLIP index is defined (oracle/cases.py)
            assert ip == int(g[f"n{step}_idx"][0]), (step, kind)
            continue
        assert [ip, ifin] == [int(v) for v in g[f"n{step}_idx"]], (step, kind)
        assert O.compute_iou(masks[ifin], gt) == tuple(int(v) for v in g[f"n{step}_IU"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_view_synthesis_vs_reference_loop(golden_dir, tag):
    """tests/golden/views.npz = the loop of Hybridgl_main.py:93-125 with its torch arithmetic (ToTensor / Resize / Normalize)
    and the uint8 compositing pinned; the blurred image is an input (ragged masks: single pixel, full, thin, empty)."""
    from hybridgl_amd import synth
    from oracle import cv_oracle as CV
    from oracle.cases import edge_masks
    g = _load(golden_dir, "views.npz")
    H, W, N, res, s_img, s_mask = (int(v) for v in g[f"{tag}_meta"])
    img = synth.synth_image(H, W, s_img)
    masks = edge_masks(N, H, W, s_mask)
    loc, glo = O.synthesize_views(img, CV.gaussian_blur_u8(img, 15), synth.imagenet_normalize(img), masks, res)
    np.testing.assert_allclose(glo, g[f"{tag}_global"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(loc, g[f"{tag}_local"], rtol=0, atol=2e-6)
