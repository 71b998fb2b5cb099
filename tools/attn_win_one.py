#!/usr/bin/env python3
"""Run the SAM ViT-H encoder (two blocks: one windowed, one global) on a group of 16 images a few times, for rocprofv3
counter passes on the attention kernels (tools/pmc_run.sh <tag> attn_x3_kernel tools/attn_win_one.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from _diag import use_lib_from_env

use_lib_from_env()      # HGL_LIB_NAME=libhybridgl_diag.so: HGL_ATTN_PS_DBG is read
from hybridgl_amd import sam as hsam, weights
from hybridgl_amd.synth import synth_image

dev = torch.device("cuda:0")
cfg = weights.SAM_CONFIGS["vit_h_d2"]
m = hsam.Sam(weights.sam_state_dict("vit_h_d2", 0), cfg, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
imgs = [torch.from_numpy(synth_image(1024, 1024, 20 + i)).to(dev) for i in range(n)]
for _ in range(3):
    m.encode_batch(imgs)
torch.cuda.synchronize()
