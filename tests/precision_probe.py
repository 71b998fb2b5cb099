#!/usr/bin/env python3
"""Error of the device encoders against the numpy oracle at full depth / under trained-checkpoint-like weight statistics
(prints the numbers the tolerances of tests/test_gpu_precision.py are set from).  usage: precision_probe.py [vith] [clip]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

from hybridgl_amd import ops, weights
from stress_weights import stress_clip_state_dict, stress_sam_state_dict

dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def vith(depth_cfg="vit_h", stress=False):
    from hybridgl_amd import sam as hsam
    from hybridgl_amd.synth import synth_image
    from oracle import sam_oracle as S
    cfg = weights.SAM_CONFIGS[depth_cfg]
    sd = weights.sam_state_dict(depth_cfg, 0)
    if stress:
        sd = stress_sam_state_dict(sd, cfg, 1)
    img = synth_image(683, 1024, 5)
    t0 = time.time()
    ref = S.image_encoder(sd, S.preprocess(img, 1024), cfg)
    t1 = time.time()
    for prec in ("f16x3", "f32"):
        m = hsam.Sam(sd, cfg, dev, precision=prec)
        emb = m.encode(T(img)).cpu().numpy().reshape(64, 64, 256)
        n_ovf = ops.split_overflow_count()
        print(f"{depth_cfg} stress={stress} {prec}: max|err| {np.abs(emb - ref).max():.3e}  rms err {np.sqrt(((emb - ref) ** 2).mean()):.3e}  "
              f"max|ref| {np.abs(ref).max():.3f} rms ref {np.sqrt((ref ** 2).mean()):.3f}  overflow {n_ovf}  (oracle {t1 - t0:.0f}s)", flush=True)
        del m


def clip(stress=True):
    from hybridgl_amd.backbone import CLIPViTFM
    from oracle import clip_oracle as O
    from oracle.cases import views_for_case
    sd = weights.clip_state_dict("ViT-B/16", 0)
    if stress:
        sd = stress_clip_state_dict(sd, 1)
    loc, glo, masks = views_for_case(4, 224, 160, 200)
    rng = np.random.default_rng(0)
    txt = rng.standard_normal((3, 512)).astype(np.float32)
    for mode in ("G2L", "G2L&L2G"):
        ref = O.clip_hybrid_forward(sd, loc, glo, masks, 9, mode, 10)
        rl = O.calculate_score(ref, txt, 100.0)
        for prec in ("f16x3", "f32"):
            m = CLIPViTFM("ViT-B/16", state_dict=sd, device=dev, precision=prec)
            y = m(T(loc), T(glo), T(masks), masking_block=9, fusion_mode=mode)
            lg = m.calculate_score(y, T(txt)).cpu().numpy() * (100.0 / m.model._logit_scale_exp)
            y = y.cpu().numpy()
            print(f"clip stress={stress} {mode} {prec}: feat max|err| {np.abs(y - ref).max():.3e} (max|ref| {np.abs(ref).max():.2f})  "
                  f"logit max|err| {np.abs(lg - rl).max():.3e}  argmax equal {np.array_equal(lg.argmax(0), rl.argmax(0))}  overflow {ops.split_overflow_count()}", flush=True)
            del m


if __name__ == "__main__":
    what = sys.argv[1:] or ["clip", "vith2", "vith"]
    if "clip" in what:
        clip(False)
        clip(True)
    if "vith2" in what:
        vith("vit_h_d2", True)
    if "vith" in what:
        vith("vit_h", False)
