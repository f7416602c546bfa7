"""finalprojectrepo.jl_amd -- MI355X-native stencil hot path of ntselepidis/FinalProjectRepo.jl.

Host-side mirror (Python; Julia is not available in this environment -- see DESIGN.md) of the
reference's script-level API for the hot path, on top of the C ABI of libfpr_hip.so (include/fpr.h):

    part1       scripts-part1/part1_kernel_programming.jl, part1_array_programming.jl, part1_utils.jl
    multigrid   scripts-part2/multigrid.jl, krylov.jl, part2_utils.jl
    part2       scripts-part2/part2.jl (NEXT row 8f-1)
    grid        ImplicitGlobalGrid's role: Cartesian decomposition + RCCL halo exchange

Julia's `f!` is spelled `f_` here; Unicode names (Hτ, dτ, diffusion_3D_step_τ) are kept.
There is NO CPU fallback: every compute entry point calls the HIP library and raises if it is missing.
"""
from . import _lib
from . import placement  # noqa: F401
from ._lib import Context, FprError, asdevice, fzeros, fones, tonumpy, lib_path  # noqa: F401

import threading as _threading

_default_ctx = None
_tls = _threading.local()      # a context bound to the calling THREAD (bind_context): several ranks in one process


def bind_context(c):
    """Make `c` the context every call of this package uses from the calling thread (None: back to the process default), and its
    compute stream torch's current stream of this thread.  One thread per rank, one context per thread: how eight ranks run their
    exchange code in ONE process on one card (tests/test_gpu_rccl.py; the GPU boxes admit six processes)."""
    _tls.ctx = c
    if c is not None:
        import torch

        torch.cuda.set_device(c.device)
        torch.cuda.set_stream(c.compute)


def init(device=0):
    """@init_parallel_stencil(...) / select_device(): create the default context on `device`."""
    global _default_ctx
    if _default_ctx is None or _default_ctx.device != device:
        _default_ctx = Context(device)
    return _default_ctx


def ctx():
    c = getattr(_tls, "ctx", None)
    if c is not None:
        return c
    if _default_ctx is None:
        return init(0)
    return _default_ctx


_second_ctx = None


def second_ctx():
    """A second context on the default context's device (own streams and multigrid arena): part2 runs the W solve on it
    beside the T solve of the same time step."""
    global _second_ctx
    c = ctx()
    if _second_ctx is None or _second_ctx.device != c.device or _second_ctx._closed:
        _second_ctx = Context(c.device, secondary=True)
        # the solve that runs beside another one must not depend on 16 workgroups being resident together: cg! in its
        # two-launch form there (a switch set on the default context, cg_fused included, still wins: copied below)
        _second_ctx.set_option("cg_fused", 2)
        _second_ctx.set_option("mg_jacobi_persist", 0)      # (the same for the persistent Jacobi coarse solve: its workgroups wait for their neighbours)
        for k, v in c.opts.items():
            _second_ctx.set_option(k, _lib.follower_value(k, v))
        c.followers = [_second_ctx]
    return _second_ctx


def reset():
    """@reset_parallel_stencil()"""
    global _default_ctx, _second_ctx
    if _second_ctx is not None:
        _second_ctx.close()
    _second_ctx = None
    if _default_ctx is not None:
        _default_ctx.close()
    _default_ctx = None


def synchronize():
    """@synchronize()"""
    ctx().synchronize()


from . import part1, multigrid, part2, grid, experiments  # noqa: E402,F401
