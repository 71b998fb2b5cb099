"""Tools only: HGL_LIB_NAME=libhybridgl_diag.so (make -C hybridgl_amd/csrc diag) makes a tool bind the diagnostic twin,
in which the experiment switches of the sources (HGL_DIAG_SWITCH: HGL_X3_GM, HGL_ATTN_WIDE, HGL_ATTN_PS_DBG, ...) are read
from the environment.  The package itself never loads anything but libhybridgl.so."""
import os


def use_lib_from_env():
    from hybridgl_amd import _lib
    name = os.environ.get("HGL_LIB_NAME")
    if name:
        _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.__file__), name)
    return _lib.LIB_PATH
