"""Import shim: the package directory is named `finalprojectrepo.jl_amd` (a dot is not importable
as a plain module name), so this module loads it under the name `fpr_amd`.

    import fpr_amd
    F = fpr_amd.load()          # fails loudly if libfpr_hip.so or the GPU is missing
    F.part1.diffusion_3D_kernel_programming(nx=64, ny=64, nz=64)
"""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "finalprojectrepo.jl_amd")
_NAME = "finalprojectrepo_jl_amd"


def _import_pkg():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


pkg = _import_pkg()


def load(device=0):
    """Returns the package with a ready default context on `device` (HIP extension required)."""
    pkg.init(device)
    return pkg


def __getattr__(name):
    return getattr(pkg, name)
