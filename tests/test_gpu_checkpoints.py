"""Models built from checkpoint FILES (clip/clip.py:119-142 -> build_model clip/model.py:474-511; build_sam.py:103-106)
through `checkpoint=` and through HYBRIDGL_CLIP_CHECKPOINT / HYBRIDGL_SAM_CHECKPOINT reproduce the reference's outputs
with the same files (tests/golden/ckpt.npz: the reference's build_model on the fp16-stored archive, up-cast to fp32 as
clip/model.py:509 leaves it; tests/golden/sam_tiny.npz for the SAM file)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import ckpt_files as CK   # noqa: E402


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_clip_from_checkpoint_file_vs_reference(cuda, golden_dir, tmp_path, fmt, monkeypatch):
    """0: plain state_dict file, 1: {'state_dict': ...}, 2: TorchScript archive (the OpenAI download).  Format 2 goes
    through the environment variable, as `python -m hybridgl_amd.main --real` gets it."""
    from hybridgl_amd.backbone import CLIPViTFM
    from oracle.cases import glue_tokens, views_for_case
    g = np.load(os.path.join(golden_dir, "ckpt.npz"))
    path = CK.write_clip_files(golden_dir, str(tmp_path))[fmt]
    if fmt == 2:
        monkeypatch.setenv("HYBRIDGL_CLIP_CHECKPOINT", path)
        model = CLIPViTFM("tiny", device=cuda)
    else:
        model = CLIPViTFM("tiny", checkpoint=path, device=cuda)
    assert model.model.cfg["vision_layers"] == 12 and model.model.cfg["transformer_heads"] == 1     # inferred from the file
    ci, n_other = (int(v) for v in g["text_tokens_case"])
    tok = torch.from_numpy(glue_tokens(ci, n_other)).to(cuda)
    np.testing.assert_allclose(model.model.encode_text(tok).cpu().numpy(), g["text"], rtol=0, atol=5e-5)
    loc, glo, masks = views_for_case(3, 64, 97, 130)
    y = model(torch.from_numpy(loc).to(cuda), torch.from_numpy(glo).to(cuda), torch.from_numpy(masks).to(cuda),
              masking_block=9, fusion_mode="G2L").cpu().numpy()
    np.testing.assert_allclose(y, g["hybrid_G2L"], rtol=0, atol=5e-5)
    # and the file's fp16 rounding is visible: the seeded fp32 weights give other features
    from hybridgl_amd import weights
    y32 = CLIPViTFM("tiny", state_dict=weights.clip_state_dict("tiny", 0), device=cuda)(
        torch.from_numpy(loc).to(cuda), torch.from_numpy(glo).to(cuda), torch.from_numpy(masks).to(cuda),
        masking_block=9, fusion_mode="G2L").cpu().numpy()
    assert np.abs(y32 - g["hybrid_G2L"]).max() > np.abs(y - g["hybrid_G2L"]).max()


@pytest.mark.parametrize("via_env", [False, True])
def test_sam_from_checkpoint_file_vs_reference(cuda, golden_dir, tmp_path, via_env, monkeypatch):
    from hybridgl_amd import sam as hsam
    from oracle.cases import sam_tiny_case
    g = np.load(os.path.join(golden_dir, "sam_tiny.npz"))
    path = CK.write_sam_file(golden_dir, str(tmp_path))
    if via_env:
        monkeypatch.setenv("HYBRIDGL_SAM_CHECKPOINT", path)
        m = hsam.sam_model_registry["tiny"](device=cuda)
    else:
        m = hsam.sam_model_registry["tiny"](checkpoint=path, device=cuda)
    c = sam_tiny_case()
    emb = m.encode(torch.from_numpy(c["resized"]).to(cuda)).cpu().numpy().reshape(16, 16, 256)
    np.testing.assert_allclose(emb[::2, ::2], g["emb_nhwc"], rtol=0, atol=1e-4)
