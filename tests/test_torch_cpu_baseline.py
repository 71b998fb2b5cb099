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


def test_cpu_thread_calibration_walks_down_to_the_fastest_setting(monkeypatch):
    """bench.py: the CPU leg runs at the best of {all, 1/2, 1/4 ...} threads (on the GPU box torch's pool at 256 threads took
    five times the 8-vCPU container's time).  The probe is replaced by a clock that is fastest at 16 threads: the walk must
    visit 256 .. 8, return 16, stop once the time has clearly turned, and restore nothing it did not set (the caller sets
    the thread count afterwards)."""
    import bench
    cost = {256: 10.0, 128: 3.0, 64: 1.5, 32: 1.2, 16: 1.0, 8: 1.4, 4: 2.6}
    now = [0.0]
    seen = []

    def fake_encoder(sd, x, cfg):
        seen.append(torch.get_num_threads())
        now[0] += cost.get(torch.get_num_threads(), 5.0)
        return x

    monkeypatch.setattr(T, "image_encoder", fake_encoder)
    monkeypatch.setattr(T, "to_torch", lambda sd: sd)
    monkeypatch.setattr(weights, "sam_state_dict", lambda name, seed: {})
    monkeypatch.setattr(bench.time, "perf_counter", lambda: now[0])
    set_calls = []
    real_set = torch.set_num_threads
    state = [torch.get_num_threads()]
    monkeypatch.setattr(torch, "set_num_threads", lambda n: (set_calls.append(n), state.__setitem__(0, n))[0])
    monkeypatch.setattr(torch, "get_num_threads", lambda: state[0])
    best, tried = bench.cpu_thread_calibration(256)
    assert best == 16
    assert list(tried) == ["256", "128", "64", "32", "16", "8"] and tried["16"] == 1.0      # 8 is 1.4x the best: the walk stops
    assert seen[0] == 128 and seen[1:] == [256, 128, 64, 32, 16, 8]                           # one untimed warm-up pass first
    assert real_set is not None


def test_gem_tower_and_heatmap_torch_cpu_equal_numpy_oracle():
    """the GEM stage of the CPU baseline (self-self attention in the last six blocks, heat-map, min-max) against gem_oracle"""
    from oracle import gem_oracle as GO
    sd = weights.clip_state_dict("tiny", 0)
    rng = np.random.default_rng(5)
    img = rng.standard_normal((2, 3, 128, 128)).astype(np.float32)
    txt = rng.standard_normal((3, sd["visual.proj"].shape[1])).astype(np.float32)
    ref_feat, _ = GO.gem_vit_forward(sd, img)
    p = sd["visual.conv1.weight"].shape[2]
    pos = torch.from_numpy(GO.interpolate_pos_encoding(sd["visual.positional_embedding"], 128 // p, 128 // p))
    with torch.no_grad():
        feat = T.gem_vit_forward(T.to_torch(sd), torch.from_numpy(img), pos)
        heat = T.gem_heatmap(feat[0], torch.from_numpy(txt), 128)
    np.testing.assert_allclose(feat.numpy(), ref_feat, rtol=0, atol=2e-5 * max(1.0, float(np.abs(ref_feat).max())))
    np.testing.assert_allclose(heat.numpy(), GO.gem_heatmap(ref_feat[0], txt, 128), rtol=0, atol=2e-4)
