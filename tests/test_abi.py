"""CPU: the C-ABI library builds, loads and exports every symbol include/hybridgl.h declares
(no compute calls: there is no GPU here and no CPU path in the library)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "hybridgl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hgl_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from hybridgl_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_header_declares_something():
    syms = header_symbols()
    assert "hgl_clip_hybrid_forward" in syms and "hgl_gemm_f32" in syms and len(syms) >= 15


def test_library_exports_every_header_symbol(lib):
    raw = ctypes.CDLL(lib._name)
    for s in header_symbols():
        assert hasattr(raw, s), f"libhybridgl.so does not export {s}"


def test_binding_table_matches_header(lib):
    from hybridgl_amd import _lib
    assert sorted(_lib.PROTOTYPES) == header_symbols()


def test_abi_version_and_error_string(lib):
    from hybridgl_amd import _lib
    assert lib.hgl_abi_version() == _lib.ABI_VERSION == 7
    assert isinstance(lib.hgl_last_error(), bytes)


def test_no_cpu_fallback_without_device(lib):
    """Without a HIP device every compute entry must refuse (HGL_ENODEVICE), never compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    rc = lib.hgl_layernorm_f32(None, None, None, None, 1, 4, 1e-5, None)
    assert rc == -2
    assert b"no HIP device" in lib.hgl_last_error()
    from hybridgl_amd import ops, _lib as L
    with pytest.raises(L.HybridGLError):
        ops.layernorm(torch.zeros(2, 8), torch.ones(8), torch.zeros(8))


def test_product_code_never_imports_oracle():
    pkg = os.path.join(ROOT, "hybridgl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


# The environment switches of the product library (csrc/hgl_common.h): A/B switches between paths that give the same bits,
# each flipped by a -m gpu test.  Anything else is a diagnostic switch and must not be readable from the shipped library.
ENV_SWITCHES = {
    "HGL_ATTN_PP": "tests/test_gpu_primitives.py::test_ping_pong_attention_is_bit_identical_to_the_tile_kernel",
    "HGL_X3_TERMS": "tests/test_gpu_primitives.py::test_gemm_f16x3_fp16_valued_weights_drop_the_zero_products",
    "HGL_SAM_POST_SEP": "tests/test_gpu_sam.py::test_postprocess_shared_table_kernel_is_bit_identical",
    "HGL_ATTN_PS_CLIPBLOCKS": "tests/test_gpu_attention_ps.py::test_clip_hybrid_forward_presplit_equals_fp32_input_path",
}


def test_library_reads_only_the_listed_environment_switches(lib):
    """No knock-out or experiment switch in the shipped library: (1) getenv appears in one source file, behind two helpers;
    (2) every call of those helpers names a listed switch; (3) the NUL-terminated strings of the built library that look
    like a variable name are exactly the list; (4) every listed switch is exercised by the -m gpu test named beside it."""
    csrc = os.path.join(ROOT, "hybridgl_amd", "csrc")
    named = set()
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h", ".cpp")):
            continue
        src = open(os.path.join(csrc, f)).read()
        code = re.sub(r"//[^\n]*", "", src)
        if f != "api_core.hip":
            assert "getenv" not in code, f"{f} reads the environment directly"
        named |= set(re.findall(r"hgl_env_(?:int|str)\(\s*\"([A-Z0-9_]+)\"", code))
    assert named == set(ENV_SWITCHES), named ^ set(ENV_SWITCHES)
    blob = open(lib._name, "rb").read()
    in_lib = {s.decode() for s in re.findall(rb"(?<=\x00)(HGL_[A-Z0-9_]+)(?=\x00)", blob)}
    assert in_lib == set(ENV_SWITCHES), in_lib ^ set(ENV_SWITCHES)
    for name, where in ENV_SWITCHES.items():
        path, test = where.split("::")
        text = open(os.path.join(ROOT, path)).read()
        body = text.split("def " + test, 1)[1].split("\ndef ", 1)[0]
        assert name in body, f"{where} does not flip {name}"


def test_package_reads_no_undocumented_environment_variable():
    """the Python side: every HYBRIDGL_* variable the package reads is described in INTEGRATION.md"""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    pkg = os.path.join(ROOT, "hybridgl_amd")
    for f in sorted(os.listdir(pkg)):
        if f.endswith(".py"):
            for name in re.findall(r"environ(?:\.get)?[\(\[]\s*[\"']([A-Z0-9_]+)[\"']", open(os.path.join(pkg, f)).read()):
                if name.startswith(("HYBRIDGL_", "HGL_")):      # launcher variables (RANK, MASTER_ADDR, ...) are torch.distributed's
                    assert name in doc, f"hybridgl_amd/{f} reads {name}, which INTEGRATION.md does not describe"
