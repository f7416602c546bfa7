"""CPU-side checks of the drop-in boundary: libfpr_hip.so loads without a GPU and exports every
symbol include/fpr.h declares; the host mirror binds exactly that set."""
import ctypes
import os
import re

import pytest

import fpr_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "fpr.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fpr_[a-z0-9_A-Z]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    path = fpr_amd.pkg.lib_path()
    assert os.path.exists(path), "libfpr_hip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    syms = header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), "symbol %s declared in include/fpr.h is not exported" % s


def test_binding_covers_header():
    assert sorted(fpr_amd.pkg._lib.ALL_SYMBOLS) == header_symbols()


def test_version_and_error_paths_without_gpu():
    L = fpr_amd.pkg._lib.load_library()
    assert b"gfx950" in L.fpr_version()
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    h = ctypes.c_void_p()
    rc = L.fpr_ctx_create(ctypes.byref(h), 0, None, None)
    assert rc == -5 and not h.value  # FPR_ERR_NO_DEVICE: no silent CPU fallback
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        fpr_amd.pkg.Context(0)
    assert L.fpr_synchronize(None) == -1


def test_julia_shim_names_every_entry_point():
    """julia/FPRHip.jl (unexecuted here: no julia) must ccall every compute symbol of the header."""
    path = os.path.join(ROOT, "julia", "FPRHip.jl")
    if not os.path.exists(path):
        pytest.skip("julia shim not written yet")
    txt = open(path).read()
    for s in header_symbols():
        assert s in txt, "julia/FPRHip.jl does not bind %s" % s
