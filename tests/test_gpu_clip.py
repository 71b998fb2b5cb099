"""GPU parity of the CLIP hybrid encoder / text encoder (drop-in CLIPViTFM surface) against
(1) fixtures captured from the reference and (2) the numpy oracle on the same seeded inputs."""
import os

import numpy as np
import pytest
import torch

from hybridgl_amd import ops, weights
from hybridgl_amd.backbone import CLIPViTFM
from oracle import clip_oracle as O
from oracle.cases import views_for_case

pytestmark = pytest.mark.gpu

MODES = ["G2L", "L2G", "G2L&L2G", "token_masking", "attn_masking", "crop"]


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


from conftest import PRECISIONS  # noqa: E402


@pytest.fixture(scope="module", params=PRECISIONS)
def tiny(cuda, request):
    """the tiny-geometry model in BOTH arithmetic modes: every golden / oracle test below that takes it runs twice"""
    sd = weights.clip_state_dict("tiny", 0)
    return sd, CLIPViTFM("tiny", state_dict=sd, device=cuda, precision=request.param)


@pytest.fixture(scope="module")
def b16(cuda):
    sd = weights.clip_state_dict("ViT-B/16", 0)
    return sd, CLIPViTFM("ViT-B/16", state_dict=sd, device=cuda)


def _run(model, loc, glo, masks, mode, dev, mb=9):
    return model(torch.from_numpy(loc).to(dev), torch.from_numpy(glo).to(dev), torch.from_numpy(masks).to(dev),
                 masking_block=mb, fusion_mode=mode).cpu().numpy()


@pytest.mark.parametrize("N", [1, 3, 5])
@pytest.mark.parametrize("mode", MODES)
def test_tiny_vs_reference_golden(cuda, golden_dir, tiny, N, mode):
    g = np.load(os.path.join(golden_dir, "clip_tiny.npz"))
    _, H, W = (int(v) for v in g["meta"])
    loc, glo, masks = views_for_case(N, 64, H, W)
    y = _run(tiny[1], loc, glo, masks, mode, cuda)
    np.testing.assert_allclose(y, g[f"N{N}_{mode}"], rtol=0, atol=5e-5)


@pytest.mark.parametrize("mode", ["G2L", "L2G", "G2L&L2G"])
def test_b16_vs_reference_golden(cuda, golden_dir, b16, mode):
    g = np.load(os.path.join(golden_dir, "clip_b16.npz"))
    _, H, W = (int(v) for v in g["meta"])
    loc, glo, masks = views_for_case(4, 224, H, W)
    y = _run(b16[1], loc, glo, masks, mode, cuda)
    np.testing.assert_allclose(y, g[f"N4_{mode}"], rtol=0, atol=1e-4)


def test_b16_vs_reference_golden_f32_mode(cuda, golden_dir):
    """ViT-B/16, G2L, against the reference's output through the exact-fp32 matrix-core path (precision='f32')"""
    g = np.load(os.path.join(golden_dir, "clip_b16.npz"))
    _, H, W = (int(v) for v in g["meta"])
    loc, glo, masks = views_for_case(4, 224, H, W)
    m = CLIPViTFM("ViT-B/16", state_dict=weights.clip_state_dict("ViT-B/16", 0), device=cuda, precision="f32")
    np.testing.assert_allclose(_run(m, loc, glo, masks, "G2L", cuda), g["N4_G2L"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("name,last_layer,mb", [("ViT-B/32", 10, 9), ("ViT-L/14", 22, 18)])
def test_other_clip_geometries_vs_oracle(cuda, name, last_layer, mb):
    """the other backbones CLIPViTFM accepts (model/backbone.py:16-24): ViT-B/32 (7x7 grid, 50 tokens) and ViT-L/14
    (width 1024, 24 layers, 16 heads, 257 tokens) through the hybrid forward, against the oracle."""
    sd = weights.clip_state_dict(name, 0)
    model = CLIPViTFM(name, state_dict=sd, device=cuda)
    loc, glo, masks = views_for_case(2, 224, 97, 130)
    for mode in ["G2L", "G2L&L2G"]:
        y = _run(model, loc, glo, masks, mode, cuda, mb)
        ref = O.clip_hybrid_forward(sd, loc, glo, masks, mb, mode, last_layer)
        np.testing.assert_allclose(y, ref, rtol=0, atol=2e-4)
    # text tower of the same checkpoint (width 512 / 768, heads = width // 64: clip/model.py:338)
    from hybridgl_amd.synth import synth_tokens
    tok = synth_tokens(3, 77, 49408, 11)
    txt = model.model.encode_text(torch.from_numpy(tok).to(cuda)).cpu().numpy()
    np.testing.assert_allclose(txt, O.encode_text(sd, tok), rtol=0, atol=5e-5)
    del model
    torch.cuda.empty_cache()


def test_tiny_masking_block_range_vs_reference_golden(cuda, golden_dir, tiny):
    """tests/golden/clip_tiny_mb.npz: masking_block None (= last_layer), 0, 1, last_layer, last_layer + 1 in every mode in
    which the reference returns features; beyond that range the library refuses (the reference returns un-normalised
    tokens or crashes there)."""
    g = np.load(os.path.join(golden_dir, "clip_tiny_mb.npz"))
    loc, glo, masks = views_for_case(3, 64, 97, 130)
    for k in [k for k in g.files if k.startswith("mb")]:
        mb, mode = k[2:].split("_", 1)
        y = _run(tiny[1], loc, glo, masks, mode, cuda, None if mb == "None" else int(mb))
        np.testing.assert_allclose(y, g[k], rtol=0, atol=5e-5, err_msg=k)
    from hybridgl_amd._lib import HybridGLError
    for mb, mode in ((12, "G2L"), (11, "attn_masking"), (13, "token_masking")):
        with pytest.raises(HybridGLError):
            _run(tiny[1], loc, glo, masks, mode, cuda, mb)


@pytest.mark.parametrize("mb", [0, 5, 11])
def test_tiny_other_masking_blocks_vs_oracle(cuda, tiny, mb):
    loc, glo, masks = views_for_case(3, 64, 97, 130)
    for mode in ["G2L", "L2G", "G2L&L2G"]:
        y = _run(tiny[1], loc, glo, masks, mode, cuda, mb)
        ref = O.clip_hybrid_forward(tiny[0], loc, glo, masks, mb, mode, 10)
        np.testing.assert_allclose(y, ref, rtol=0, atol=5e-5)


def test_b16_scores_and_winner_vs_oracle(cuda, b16):
    """north_star bar: similarity scores within 1e-3 (logits are x100), winning index bit-exact."""
    sd, model = b16
    N = 8
    rng = np.random.default_rng(77)
    loc = rng.standard_normal((N, 3, 224, 224)).astype(np.float32)
    glo = rng.standard_normal((N, 3, 224, 224)).astype(np.float32)
    from hybridgl_amd.synth import synth_masks, synth_tokens
    masks = synth_masks(N, 640, 640, 5)
    tok = synth_tokens(2, 77, 49408, 6)
    feats = model(torch.from_numpy(loc).to(cuda), torch.from_numpy(glo).to(cuda), torch.from_numpy(masks).to(cuda),
                  masking_block=9, fusion_mode="G2L")
    txt = model.model.encode_text(torch.from_numpy(tok).to(cuda))
    logits = model.calculate_score(feats, txt).cpu().numpy()
    rf = O.clip_hybrid_forward(sd, loc, glo, masks, 9, "G2L", 10)
    rt = O.encode_text(sd, tok, heads=8)
    rl = O.calculate_score(rf, rt, float(np.exp(sd["logit_scale"])))
    np.testing.assert_allclose(txt.cpu().numpy(), rt, rtol=0, atol=5e-5)
    np.testing.assert_allclose(logits, rl, rtol=0, atol=1e-3)
    assert np.array_equal(logits.argmax(0), rl.argmax(0))


@pytest.mark.parametrize("name,cfg", [("text_tiny.npz", "tiny"), ("text_b16.npz", "ViT-B/16")])
def test_encode_text_vs_reference_golden(cuda, golden_dir, tiny, b16, name, cfg):
    g = np.load(os.path.join(golden_dir, name))
    model = tiny[1] if cfg == "tiny" else b16[1]
    y = model.model.encode_text(torch.from_numpy(g["tokens"]).to(cuda)).cpu().numpy()
    np.testing.assert_allclose(y, g["out"], rtol=0, atol=5e-5)


@pytest.mark.parametrize("name,cfg", [("text_pool_tiny.npz", "tiny"), ("text_pool_b16.npz", "ViT-B/16")])
def test_text_pooling_and_token_masking_vs_reference_golden(cuda, golden_dir, tiny, b16, name, cfg):
    """CLIP.encode_text(text, target_noun_index) (clip/model.py:426-428) and CLIPViTFM.text_masking_feature
    (model/backbone.py:34-56) on the device vs outputs of the imported reference"""
    g = np.load(os.path.join(golden_dir, name))
    model = tiny[1] if cfg == "tiny" else b16[1]
    tok = torch.from_numpy(g["tokens"]).to(cuda)
    for k in (0, 1, 3):
        y = model.model.encode_text(tok, target_noun_index=k).cpu().numpy()
        np.testing.assert_allclose(y, g[f"pool_{k}"], rtol=0, atol=5e-5)
    # one index per string (the reference's truth test admits a single element only): row-wise the same
    per_row = model.model.encode_text(tok, target_noun_index=torch.tensor([1, 3, 1, 3, 1])).cpu().numpy()
    np.testing.assert_allclose(per_row[0::2], g["pool_1"][0::2], rtol=0, atol=5e-5)
    np.testing.assert_allclose(per_row[1::2], g["pool_3"][1::2], rtol=0, atol=5e-5)
    for mb in g["masking_blocks"]:
        for tag, idx in (("idx12", [1, 2]), ("idx0", [0])):
            y = model.text_masking_feature(tok, masking_index=idx, masking_block=int(mb)).cpu().numpy()
            np.testing.assert_allclose(y, g[f"mask_{int(mb)}_{tag}"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(model.text_masking_feature(tok).cpu().numpy(), g["mask_none"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(model.text_feature(tok).cpu().numpy(), g["mask_none"], rtol=0, atol=5e-5)
    with pytest.raises(IndexError):
        model.model.encode_text(tok, target_noun_index=model.model.context_length)


def test_encode_text_prefix_is_exact(cuda, b16):
    """encode_text(seq_len=L): only the first L of the 77 positions are computed; under the causal mask the EOT feature
    does not depend on later positions, so the result equals the full computation when every EOT lies inside the
    prefix -- and is NaN (not a wrong feature) for a string whose EOT lies beyond it."""
    from hybridgl_amd.synth import synth_tokens
    _, model = b16
    tok = synth_tokens(12, 77, 49408, 21)
    L = int(tok.argmax(axis=1).max()) + 1
    assert L <= 14
    t = torch.from_numpy(tok).to(cuda)
    full = model.model.encode_text(t)
    pre = model.model.encode_text(t, seq_len=L)
    np.testing.assert_allclose(pre.cpu().numpy(), full.cpu().numpy(), rtol=0, atol=1e-5)   # different GEMM kernels at 168 vs 924 rows
    np.testing.assert_allclose(model.model.encode_text(t, seq_len=L + 9).cpu().numpy(), full.cpu().numpy(), rtol=0, atol=1e-5)
    short = model.model.encode_text(t, seq_len=L - 1).cpu().numpy()
    late = tok.argmax(axis=1) >= L - 1
    assert late.any() and np.isnan(short[late]).all() and np.isfinite(short[~late]).all()
    np.testing.assert_allclose(short[~late], full.cpu().numpy()[~late], rtol=0, atol=1e-5)


def test_linearity_of_head_and_determinism(cuda, b16):
    """size-independent properties at the BASELINE size (N=64): the result does not depend on the
    batch composition (each mask row is independent) and repeated runs are bit-identical."""
    _, model = b16
    N = 64
    rng = np.random.default_rng(78)
    loc = torch.from_numpy(rng.standard_normal((N, 3, 224, 224)).astype(np.float32)).to(cuda)
    glo = torch.from_numpy(rng.standard_normal((N, 3, 224, 224)).astype(np.float32)).to(cuda)
    from hybridgl_amd.synth import synth_masks
    masks = torch.from_numpy(synth_masks(N, 640, 640, 9)).to(cuda)
    y1 = model(loc, glo, masks, masking_block=9, fusion_mode="G2L")
    y2 = model(loc, glo, masks, masking_block=9, fusion_mode="G2L")
    assert torch.equal(y1, y2)
    sub = model(loc[10:14].contiguous(), glo[10:14].contiguous(), masks[10:14].contiguous(), masking_block=9,
                fusion_mode="G2L")
    np.testing.assert_allclose(sub.cpu().numpy(), y1[10:14].cpu().numpy(), rtol=0, atol=2e-5)
    assert torch.isfinite(y1).all()


@pytest.mark.parametrize("mode", ["L2G", "G2L&L2G", "G2L"])
def test_baseline_size_rows_vs_oracle(cuda, b16, mode):
    """BASELINE configs[1-3] shape: N = 64 proposals on a 640 x 640 image, ViT-B/16, every fusion mode.  The oracle on 64
    masks takes minutes; each mask row of CLIPViTFM.forward is independent of the others (model/backbone.py:206-306: no
    op mixes rows), so 8 rows sampled from the N = 64 device run are checked against the oracle run on those 8 masks,
    and the batch-composition independence of the other 56 is checked on the device (the sub-batch runs other GEMM
    tile counts)."""
    sd, model = b16
    N = 64
    rng = np.random.default_rng(7)
    loc = rng.standard_normal((N, 3, 224, 224)).astype(np.float32)
    glo = rng.standard_normal((N, 3, 224, 224)).astype(np.float32)
    from hybridgl_amd.synth import synth_masks
    masks = synth_masks(N, 640, 640, 11)
    y = model(T(loc, cuda), T(glo, cuda), T(masks, cuda), masking_block=9, fusion_mode=mode).cpu().numpy()
    pick = np.array([0, 5, 17, 22, 31, 40, 58, 63])
    ref = O.clip_hybrid_forward(sd, loc[pick], glo[pick], masks[pick], 9, mode, 10)
    np.testing.assert_allclose(y[pick], ref, rtol=0, atol=1e-4)
    txt = rng.standard_normal((3, 512)).astype(np.float32)
    lg = ops.calculate_score(T(y[pick], cuda), T(txt, cuda), 100.0).cpu().numpy()
    np.testing.assert_allclose(lg, O.calculate_score(ref, txt, 100.0), rtol=0, atol=1e-3)
    rest = np.setdiff1d(np.arange(N), pick)
    sub = model(T(loc[rest], cuda), T(glo[rest], cuda), T(masks[rest], cuda), masking_block=9, fusion_mode=mode).cpu().numpy()
    np.testing.assert_allclose(sub, y[rest], rtol=0, atol=2e-5)


@pytest.mark.parametrize("mode", ["L2G", "G2L&L2G", "G2L"])
def test_baseline_size_all_rows_vs_torch_cpu_restatement(cuda, b16, mode):
    """The same shape with NO sampling: all 64 rows of the device run against oracle/torch_cpu.py -- the plain-PyTorch CPU
    restatement of the reference's operators that tests/test_torch_cpu_baseline.py pins to the numpy oracle (which the
    imported reference pins: tests/test_oracle_golden.py) -- ten seconds for 64 masks where the numpy oracle takes minutes.
    Tolerances as in the sampled test: rows 1e-4, logits (x100) 1e-3, winners identical."""
    from oracle import torch_cpu as TC
    sd, model = b16
    N = 64
    rng = np.random.default_rng(7)
    loc = rng.standard_normal((N, 3, 224, 224)).astype(np.float32)
    glo = rng.standard_normal((N, 3, 224, 224)).astype(np.float32)
    from hybridgl_amd.synth import synth_masks
    masks = synth_masks(N, 640, 640, 11)
    y = model(T(loc, cuda), T(glo, cuda), T(masks, cuda), masking_block=9, fusion_mode=mode).cpu().numpy()
    before = torch.get_num_threads()
    torch.set_num_threads(min(16, before))      # torch's pool at every hardware thread of a 256-thread host is 10x slower
    try:
        with torch.no_grad():
            ref = TC.clip_hybrid_forward(TC.to_torch(sd), torch.from_numpy(loc), torch.from_numpy(glo), torch.from_numpy(masks),
                                         9, mode, 10).numpy()
    finally:
        torch.set_num_threads(before)
    np.testing.assert_allclose(y, ref, rtol=0, atol=1e-4)
    txt = rng.standard_normal((3, 512)).astype(np.float32)
    lg = ops.calculate_score(T(y, cuda), T(txt, cuda), 100.0).cpu().numpy()
    rl = O.calculate_score(ref, txt, 100.0)
    np.testing.assert_allclose(lg, rl, rtol=0, atol=1e-3)
    assert np.array_equal(lg.argmax(0), rl.argmax(0))


def test_pipeline_image_cache_identical(cuda, b16):
    """two refs of the same image: the cached second ref gives the same indices/metrics as recomputing"""
    import dataclasses
    from hybridgl_amd.pipeline import HybridGLPipeline, synthetic_ref
    _, model = b16
    ref0, _ = synthetic_ref(0, cuda, N=8)
    ref1, _ = synthetic_ref(1, cuda, N=8)
    # second ref: same image/masks as ref0, its own sentences
    ref_b = dataclasses.replace(ref0, tokens=ref1.tokens, token_len=ref1.token_len, sentences=ref1.sentences, target=ref0.target)
    p1 = HybridGLPipeline(model, "G2L", 9)
    p1.step(ref0); out_plain = p1.step(ref_b)
    p2 = HybridGLPipeline(model, "G2L", 9)
    p2.step(dataclasses.replace(ref0, image_id=7)); out_cached = p2.step(dataclasses.replace(ref_b, image_id=7))
    assert torch.equal(out_plain[2][0], out_cached[2][0])          # winning indices
    assert torch.equal(out_plain[0], out_cached[0])                # hybrid features bit-identical
    assert p1.metrics()["cum"] == p2.metrics()["cum"]


@pytest.mark.parametrize("H,W", [(40, 56), (157, 203), (480, 640)])
def test_gaussian_blur_device_matches_host_definition(cuda, H, W):
    """hgl_gaussian_blur_u8 == synth.box_blur_u8 bit for bit (the package's stand-in for cv2.GaussianBlur, whose
    own fixed-point arithmetic is unpinned: SURVEY.md 8f-2)."""
    from hybridgl_amd import ops, synth
    rng = np.random.default_rng(H)
    img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    img[: H // 3] = synth.synth_image(H, W, 1)[: H // 3]
    got = ops.gaussian_blur_u8(torch.from_numpy(img).to(cuda), mode="float").cpu().numpy()
    assert np.array_equal(got, synth.box_blur_u8(img))


@pytest.mark.parametrize("H,W,k", [(40, 56, 15), (157, 203, 15), (480, 640, 15), (64, 48, 5), (33, 47, 9), (64, 64, 31)])
def test_cv2_fixed_point_blur_bit_exact_vs_oracle(cuda, H, W, k):
    """hgl_cv_gaussian_kernel_q8 + hgl_gaussian_blur_u8_q8 == the integer restatement of OpenCV's 8-bit GaussianBlur
    (oracle/cv_oracle.py; parity with the cv2 package itself unpinned), bit for bit: taps and pixels."""
    from hybridgl_amd import ops, synth
    from oracle import cv_oracle as CV
    assert list(ops.cv_gaussian_kernel_q8(k)) == CV.gaussian_kernel_q8(k)
    rng = np.random.default_rng(H * k)
    img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    img[: H // 3] = synth.synth_image(H, W, 1)[: H // 3]
    img[-2:] = 255                                    # saturated border rows: the 16-bit row sums peak at 255 * 256
    got = ops.gaussian_blur_u8(torch.from_numpy(img).to(cuda), k).cpu().numpy()
    assert np.array_equal(got, CV.gaussian_blur_u8(img, k))
    flat = np.full((H, W, 3), 77, dtype=np.uint8)     # the taps sum to 256: a constant image is a fixed point
    assert np.array_equal(ops.gaussian_blur_u8(torch.from_numpy(flat).to(cuda), k).cpu().numpy(), flat)


@pytest.mark.gpu
@pytest.mark.parametrize("C", [1, 4, 6])
def test_cv2_fixed_point_blur_other_channel_counts(cuda, C):
    """1 and 4 channels go through the one-launch tile kernel like RGB, 6 through the two-launch form (rows of 16-bit sums in
    the workspace): both bit-exact against the oracle, on a size that is not a multiple of the 32 x 64 tile."""
    from hybridgl_amd import ops
    from oracle import cv_oracle as CV
    img = np.random.default_rng(C).integers(0, 256, size=(75, 131, C), dtype=np.uint8)
    for k in (15, 31):
        assert np.array_equal(ops.gaussian_blur_u8(torch.from_numpy(img).to(cuda), k).cpu().numpy(), CV.gaussian_blur_u8(img, k))


def test_forward_accepts_what_the_reference_accepts(cuda, tiny):
    """model/backbone.py:123,160 cast whatever comes in (`x.type(self.model.dtype)`, `pred_masks.type(torch.float32)`): image
    tensors of another float type, non-contiguous views, masks as bool / uint8 / int64 / float {0, 1} give the same features;
    host tensors and unknown modes are refused with a message (no CPU path exists)."""
    from hybridgl_amd._lib import HybridGLError
    loc, glo, masks = views_for_case(3, 64, 97, 130)
    L, G, M = T(loc, cuda), T(glo, cuda), T(masks, cuda)
    base = tiny[1](L, G, M, masking_block=9, fusion_mode="G2L")
    for Mx in (M.bool(), M.to(torch.uint8), M.long(), M.float()):
        assert torch.equal(tiny[1](L, G, Mx, masking_block=9, fusion_mode="G2L"), base)
    Ln = L.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    assert not Ln.is_contiguous() and torch.equal(tiny[1](Ln, G, M, masking_block=9, fusion_mode="G2L"), base)
    assert torch.equal(tiny[1](L.double(), G.double(), M, masking_block=9, fusion_mode="G2L"), base)
    y16 = tiny[1](L.half(), G.half(), M, masking_block=9, fusion_mode="G2L")      # the inputs were rounded to fp16 by the caller
    assert torch.isfinite(y16).all() and float((y16 - base).abs().max()) < 0.05
    with pytest.raises(HybridGLError):
        tiny[1](L.cpu(), G.cpu(), M.cpu(), masking_block=9, fusion_mode="G2L")
    with pytest.raises(HybridGLError):
        tiny[1](L, None, M, masking_block=9, fusion_mode="G2L")
    with pytest.raises(ValueError):
        tiny[1](L, G, M, masking_block=9, fusion_mode="nope")
    assert tiny[1](L, None, M, fusion_mode="crop").shape == base.shape


def test_reference_named_helpers_vs_golden(cuda, golden_dir):
    """hybridgl_amd.utils.{gen_dir_mask, relation_boxes, Compute_IoU}: the reference's utils.py names, device
    arithmetic, against the vectors captured from the reference (tests/golden/scoring.npz)."""
    from hybridgl_amd import utils as U
    g = np.load(os.path.join(golden_dir, "scoring.npz"))
    for flag in ["left", "right", "middle", "none", "up"]:
        for (h, w) in [(3, 5), (4, 8), (2, 640), (2, 427)]:
            got = U.gen_dir_mask(flag, h, w, cuda).cpu().numpy()
            assert np.array_equal(got, g[f"dm_{flag}_{h}_{w}"]), (flag, h, w)      # bit-exact (linspace as single fmas)
    boxes, scores, tab = g["rb_boxes"], g["rb_scores"], g["rb_table"]
    tb = torch.from_numpy(boxes.astype(np.int64)).to(cuda)
    ts = torch.from_numpy(scores.astype(np.float32)).to(cuda)
    ii, jj = np.meshgrid(np.arange(4), np.arange(4), indexing="ij")
    for w, word in enumerate(g["rb_words"]):
        got = U.relation_boxes(tb[ii.ravel()], tb[jj.ravel()], ts[ii.ravel()], ts[jj.ravel()], str(word)).cpu().numpy()
        np.testing.assert_allclose(got.reshape(4, 4), tab[w], rtol=0, atol=1e-7)
        one = U.relation_boxes(tb[0], tb[1], ts[0], ts[1], str(word))
        assert one.dim() == 0 and abs(float(one) - float(tab[w, 0, 1])) <= 1e-7
    pred, gt = torch.from_numpy(g["iou_pred"]).to(cuda), torch.from_numpy(g["iou_gt"]).to(cuda)
    cum_I, cum_U = torch.zeros((), dtype=torch.int64, device=cuda), torch.zeros((), dtype=torch.int64, device=cuda)
    iou, lst, cum_I, cum_U = U.Compute_IoU(pred, gt, cum_I, cum_U, [])
    I, Uc = (int(v) for v in g["iou_IU"])
    assert (int(cum_I), int(cum_U)) == (I, Uc) and abs(float(iou) - I / Uc) < 1e-6 and len(lst) == 1
    z = torch.zeros((3, 3), dtype=torch.bool, device=cuda)
    iou0, _, _, _ = U.Compute_IoU(z, z, cum_I, cum_U, [])
    assert iou0 == 0.0
