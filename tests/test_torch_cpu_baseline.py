"""bench.py's CPU baseline (oracle/torch_cpu.py: the heavy stages restated in plain PyTorch on the CPU) against the numpy oracle,
which is pinned against outputs of the imported reference (tests/test_oracle_golden.py, tests/test_oracle_sam_golden.py):
the number on the bench line is only as good as the code it times."""
import numpy as np
import pytest
import torch

from hybridgl_amd import weights
from oracle import clip_oracle as O
from oracle import sam_oracle as S
from oracle import torch_cpu as T
from oracle.cases import sam_tiny_case, views_for_case


@pytest.mark.parametrize("mode", ["G2L", "L2G", "G2L&L2G"])
def test_clip_hybrid_forward_torch_cpu_equals_numpy_oracle(mode):
    sd = weights.clip_state_dict("tiny", 0)
    loc, glo, masks = views_for_case(5, 64, 97, 130)
    masks[1] = False                      # a mask that keeps nothing after the resize
    masks[2] = True
    ref = O.clip_hybrid_forward(sd, loc, glo, masks, 9, mode, 10)
    with torch.no_grad():
        got = T.clip_hybrid_forward(T.to_torch(sd), torch.from_numpy(loc), torch.from_numpy(glo), torch.from_numpy(masks), 9, mode, 10)
    np.testing.assert_allclose(got.numpy(), ref, rtol=0, atol=2e-5 * max(1.0, float(np.abs(ref).max())))


def test_encode_text_torch_cpu_equals_numpy_oracle():
    sd = weights.clip_state_dict("tiny", 0)
    rng = np.random.default_rng(3)
    vocab = sd["token_embedding.weight"].shape[0]
    ctx = sd["positional_embedding"].shape[0]
    tok = np.zeros((4, ctx), dtype=np.int64)
    for i in range(4):
        n = 3 + 2 * i
        tok[i, 0] = vocab - 2
        tok[i, 1:n] = rng.integers(1, vocab - 2, n - 1)
        tok[i, n] = vocab - 1
    ref = O.encode_text(sd, tok)
    with torch.no_grad():
        got = T.encode_text(T.to_torch(sd), tok)
    np.testing.assert_allclose(got.numpy(), ref, rtol=0, atol=2e-5 * max(1.0, float(np.abs(ref).max())))


def test_sam_encoder_and_decoder_torch_cpu_equal_numpy_oracle():
    cfg = weights.SAM_CONFIGS["tiny"]
    sd = weights.sam_state_dict("tiny", 0)
    c = sam_tiny_case()
    x = S.preprocess(c["resized"], cfg["img_size"])
    emb_ref = S.image_encoder(sd, x, cfg)
    sdt = T.to_torch(sd)
    with torch.no_grad():
        emb = T.image_encoder(sdt, torch.from_numpy(x), cfg)
    np.testing.assert_allclose(emb.numpy(), emb_ref, rtol=0, atol=2e-4 * max(1.0, float(np.abs(emb_ref).max())))
    sparse = S.embed_points(sd, c["points_in"], cfg["img_size"])
    low_ref, iou_ref = S.mask_decoder(sd, emb_ref, sparse)
    with torch.no_grad():
        low, iou = T.mask_decoder(sdt, torch.from_numpy(emb_ref), torch.from_numpy(sparse))
    np.testing.assert_allclose(low.numpy(), low_ref, rtol=0, atol=2e-4 * max(1.0, float(np.abs(low_ref).max())))
    np.testing.assert_allclose(iou.numpy(), iou_ref, rtol=0, atol=1e-4)


def test_postprocess_torch_cpu_equals_numpy_oracle():
    rng = np.random.default_rng(11)
    low = (rng.standard_normal((2, 3, 64, 64)) * 3).astype(np.float32)
    ref = S.postprocess_masks(low, (1024, 800), (120, 94))
    st_ref, _, _ = S.stability_score(ref.reshape(6, 120, 94))
    box_ref = S.mask_to_box(ref.reshape(6, 120, 94) > 0)
    with torch.no_grad():
        b, st, boxes = T.postprocess_and_stats(torch.from_numpy(low), (1024, 800), (120, 94))
    assert (b.numpy() != (ref.reshape(6, 120, 94) > 0)).mean() < 1e-3
    np.testing.assert_allclose(st.numpy(), st_ref, atol=2e-3)
    assert np.abs(boxes.numpy() - box_ref).max() <= 1
