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


def test_three_iteration_kernel_is_built_without_register_spills():
    """k_diff3_march3's design IS its register budget (csrc/diffusion3d_fused3.hpp: twelve planes of 12 registers + the point update's
    temporaries in the 256 registers of two waves per SIMD): one spilled row costs a drain of every load in flight per iteration (measured:
    1.165 ms per launch with twelve spilled registers, 0.92 ms without).  The Makefile keeps the compiler's resource remarks of
    diffusion3d.hip beside the object; every instantiation of the kernel must show no spills and no scratch."""
    import re

    path = os.path.join(ROOT, "finalprojectrepo.jl_amd", "csrc", "build", "diffusion3d.remarks")
    if not os.path.exists(path):
        import pytest

        pytest.skip("library not built here (build/diffusion3d.remarks absent)")
    txt = open(path, errors="replace").read()
    blocks = re.split(r"remark: Function Name: ", txt)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if "k_diff3_march3" not in name:
            continue
        seen += 1
        spill = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
        assert spill == 0 and scratch == 0 and occ == 2, (name, spill, scratch, occ)
    assert seen == 4, seen          # NORM x WRES
