"""julia/FPRHip.jl against the surface the reference's seven hot-path files use -- no GPU, no julia.

tests/golden/reference_surface.json holds NAMES only (tools/make_reference_surface.py reads them off /root/reference in the
build container): every macro the files use, how often outside kernel-definition bodies, every kernel they define with
@parallel_indices / `@parallel function`, the macros / functions they define themselves and the ParallelStencil /
ImplicitGlobalGrid names they call.  The shim is unexecuted here (no julia in the image), so this is a static check that
nothing the files need is missing: with it, the only edit to a reference file is its `using` block (or none at all through
`include_reference`)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = open(os.path.join(ROOT, "julia", "FPRHip.jl"), encoding="utf-8").read()
SURF = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_surface.json"), encoding="utf-8"))

# macros of Base / the standard library (Printf, Logging) the files use outside kernel bodies
BASE_MACROS = {"views", "assert", "warn", "debug", "enum", "sprintf", "static", "isdefined", "__DIR__", "__FILE__"}


def shim_macros():
    return set(re.findall(r"^macro\s+([A-Za-z_][A-Za-z_0-9]*)", SHIM, flags=re.M))


def shim_exports():
    m = re.search(r"^export\s+(.*?)\n\n", SHIM, flags=re.M | re.S)
    return {t.strip() for t in m.group(1).replace("\n", " ").split(",") if t.strip()}


def shim_defined():
    names = set(re.findall(r"^function\s+([^\s(]+)\s*\(", SHIM, flags=re.M))
    names |= set(re.findall(r"^const\s+([^\s=]+)\s*=", SHIM, flags=re.M))
    names |= set(re.findall(r"^([A-Za-z_][^\s(=]*)\(.*\)\s*=", SHIM, flags=re.M))
    return names


def shim_kernel_set():
    m = re.search(r"const KERNELS = Set\{Symbol\}\(\[(.*?)\]\)", SHIM, flags=re.S)
    body = m.group(1)
    return set(re.findall(r'Symbol\("([^"]+)"\)', body)) | set(re.findall(r"(?<![A-Za-z_\"]):([A-Za-z_][^\s,\]]*)", body))


def test_every_kernel_definition_of_the_reference_is_swallowed_and_provided():
    kernels = {k["name"] for k in SURF["kernels"]}
    assert len(SURF["kernels"]) == 15 and len(kernels) == 15          # 14 definitions by @parallel_indices + 1 by `@parallel function`
    assert sum(k["form"] == "@parallel_indices" for k in SURF["kernels"]) == 14
    assert shim_kernel_set() == kernels, (shim_kernel_set() ^ kernels)
    defined, exported = shim_defined(), shim_exports()
    for k in kernels:
        assert k in defined, "FPRHip.jl swallows the definition of %s but does not define it" % k
        assert k in exported, "FPRHip.jl does not export %s" % k
    # the definition forms themselves
    assert {"parallel", "parallel_indices"} <= shim_macros()
    assert "swallow_kernel_definition(\"@parallel_indices\"" in SHIM and "swallow_kernel_definition(\"@parallel\"" in SHIM


def test_every_macro_used_outside_kernel_bodies_is_understood():
    mine = shim_macros()
    exported = shim_exports()
    for name, use in SURF["macros"].items():
        if use["outside_kernel_bodies"] == 0:
            continue        # only ever inside a swallowed kernel body (@threadIdx, @sharedMem, @all, @qx ...): never expanded
        if name in BASE_MACROS or name in SURF["macros_defined"]:
            continue
        assert name in mine, "the reference uses @%s outside kernel bodies; FPRHip.jl has no such macro" % name
        assert "@" + name in exported, "FPRHip.jl does not export @%s" % name
    # macros that exist only inside kernel bodies must really be unreachable: none of them is defined or needed
    body_only = {n for n, u in SURF["macros"].items() if u["outside_kernel_bodies"] == 0}
    assert {"threadIdx", "blockDim", "sharedMem", "sync_threads", "atomic", "all", "inn"} <= body_only


def test_parallelstencil_and_implicitglobalgrid_surface_is_exported():
    exported = shim_exports()
    for name in SURF["surface_calls"]:
        head = name.split(".")[0]
        assert head in exported, "the reference calls %s; FPRHip.jl does not export %s" % (name, head)
    assert "module Data" in SHIM and "const Array = ROCArray{Float64}" in SHIM


def test_include_reference_drops_only_what_the_shim_provides():
    ref_funcs = {f["name"] for f in SURF["functions_defined"]}
    m = re.search(r"const NATIVE_HOST = Set\{Tuple\{Symbol,Int\}\}\(\[(.*?)\]\)", SHIM, flags=re.S)
    native = set(re.findall(r'Symbol\("([^"]+)"\)', m.group(1))) | set(re.findall(r"\(:([A-Za-z_][A-Za-z_0-9]*),", m.group(1)))
    assert native, "NATIVE_HOST not found"
    defined = shim_defined()
    for n in native:
        assert n in ref_funcs, "NATIVE_HOST names %s, which the reference files do not define" % n
        assert n in defined, "include_reference(fast=true) drops %s but FPRHip.jl does not define it" % n
    # the files' own driver functions must survive the filter
    for keep in ("diffusion_3D_kernel_programming", "navier_stokes_2D", "init_local_gaussian", "linear_interpolate_3D", "compute_dt"):
        assert keep in ref_funcs and keep not in native
    for pkg in ("CUDA", "ParallelStencil", "ImplicitGlobalGrid"):
        assert ":" + pkg in SHIM.split("const ABSENT_PACKAGES")[1].split("\n")[0]
