import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# Both arithmetic modes of the library in the driver's GPU record: the module fixtures of the parity tests (CLIP tiny, SAM tiny,
# GEM tiny, the whole-ref world) are parametrised on this list, so every test that uses them runs once per mode.
PRECISIONS = ["f16x3", "f32"]


@pytest.fixture(autouse=True)
def _library_precision_is_the_default_at_test_entry(request):
    """The library keeps ONE current precision and every model re-asserts its own on entry (ops.use_precision); tests that
    call operators directly (ops.gemm_f16x3, ops.attention) follow the current mode, so each GPU test starts from the
    default whatever model ran last."""
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            from hybridgl_amd import ops
            ops.set_precision(ops.default_precision())
    yield


@pytest.fixture(scope="session")
def cuda():
    """cuda:0 or a hard failure: GPU tests must never silently pass on a CPU fallback."""
    import torch
    assert torch.cuda.is_available(), "GPU test selected but no HIP device is visible"
    return torch.device("cuda:0")
