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
    assert lib.hgl_abi_version() == _lib.ABI_VERSION == 6
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
