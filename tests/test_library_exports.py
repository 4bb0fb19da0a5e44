"""CPU-side checks of the C-ABI boundary: the library builds, loads and exports every symbol that
include/maa.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from audiogpt_amd import build
    return build.build(verbose=False)


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "maa.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(maa_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    syms = _declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_binding_matches_header(lib_path):
    from audiogpt_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared_symbols()
    lib = _lib.load()
    assert b"gfx950" in lib.maa_version()


def test_error_path_without_gpu(lib_path):
    """No GPU in the build container: creating a context must fail cleanly with a message, not crash."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from audiogpt_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    st = lib.maa_ctx_create(0, None, ctypes.byref(h))
    assert st < 0
    assert len(lib.maa_last_error()) > 0
    from audiogpt_amd import backend
    with pytest.raises(_lib.MaaError):
        backend.Context("cuda:0")


def test_null_arguments_are_rejected(lib_path):
    from audiogpt_amd import _lib
    lib = _lib.load()
    assert lib.maa_ctx_synchronize(None) < 0
    assert lib.maa_unet_forward(None, None, None, None, 1, 1, 1, None) < 0
    assert lib.maa_vocoder_forward(None, None, None, 1, 1, None) < 0
    assert b"null" in lib.maa_last_error() or b"bad" in lib.maa_last_error()
