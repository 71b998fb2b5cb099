"""CPU: register / scratch budgets of the kernels whose occupancy the design counts on, read from the compiler's own
assembly (hipcc cross-compiles gfx950 without a GPU).  A kernel that silently grows past its budget still computes the right
numbers -- at half the resident waves, or with spill reloads behind `s_waitcnt vmcnt(0)` in its inner loop."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hybridgl_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# file -> {kernel name fragment: (max VGPRs, max scratch bytes)}
BUDGETS = {
    "sam_decoder_t2i.hip": {
        "dec_i2t_fold_kernel": (128, 0),      # two workgroups of eight waves per CU
        "t2i_raw_attn_kernel": (256, 0),      # two workgroups of four waves per CU
    },
    "sam_decoder_fused.hip": {
        "dec_tail_kernel": (128, 0),          # sixteen waves per CU
        "dec_i2t_kernel": (128, 0),
    },
}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src", sorted(BUDGETS))
def test_kernel_register_and_scratch_budgets(src, tmp_path):
    out = tmp_path / (src + ".s")
    mk = open(os.path.join(CSRC, "Makefile")).read()      # the flags the library is built with
    flags = re.search(r"^CXXFLAGS\s*=\s*(.*)$", mk, flags=re.M).group(1).replace("$(ARCH)", "gfx950").replace("$(EXTRA)", "").split()
    extra = re.search(r"^build/%s:\s*EXTRA\s*\+=\s*(.*)$" % re.escape(src.replace(".hip", ".o")), mk, flags=re.M)
    flags = [f for f in flags if f != "-fPIC"] + (extra.group(1).split() if extra else [])
    subprocess.run([HIPCC] + flags + ["-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, src)], check=True,
                   capture_output=True, timeout=900)
    asm = out.read_text()
    for frag, (max_vgpr, max_scratch) in BUDGETS[src].items():
        v = re.findall(r"\.set (\S*%s\S*)\.num_vgpr, (\d+)" % frag, asm)
        s = re.findall(r"\.set (\S*%s\S*)\.private_seg_size, (\d+)" % frag, asm)
        assert v and s, f"{src}: no kernel matching {frag}"
        for name, n in v:
            assert int(n) <= max_vgpr, f"{name}: {n} VGPRs > {max_vgpr}"
        for name, n in s:
            assert int(n) <= max_scratch, f"{name}: {n} bytes of scratch"
