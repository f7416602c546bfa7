"""ctypes binding of libfpr_hip.so (include/fpr.h) + device-array helpers.

torch is used for device memory, streams and (in grid.py) torch.distributed -- plumbing only.
Field arrays are torch float64 CUDA tensors with Julia (column-major) strides: a tensor of Julia
shape (nx, ny, nz) has strides (1, nx, nx*ny); build them with fzeros()/asdevice().
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FPR_COARSE_JACOBI = 0
FPR_COARSE_CG = 1
ERRORS = {-1: "FPR_ERR_INVALID", -2: "FPR_ERR_HIP", -3: "FPR_ERR_NOT_POW2", -4: "FPR_ERR_ASSERT", -5: "FPR_ERR_NO_DEVICE",
          -6: "FPR_ERR_RCCL"}


class FprError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "FPR_ERR"), code, msg))
        self.code = code


def lib_path():
    # FPR_LIB_PATH: another build of the same library (A/B runs of kernel variants from tools/); no fallback either way
    return os.environ.get("FPR_LIB_PATH") or os.path.join(_HERE, "lib", "libfpr_hip.so")


def build(verbose=False):
    """Compile libfpr_hip.so for gfx950 with hipcc (csrc/Makefile)."""
    import subprocess

    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    return lib_path()


_d, _i, _l, _z, _vp = C.c_double, C.c_int, C.c_long, C.c_size_t, C.c_void_p
_dp = C.c_void_p  # device pointers travel as integers
_SIG = {
    "fpr_ctx_create": [C.POINTER(_vp), _i, _vp, _vp],
    "fpr_ctx_destroy": [_vp],
    "fpr_synchronize": [_vp],
    "fpr_set_option": [_vp, C.c_char_p, _l],
    "fpr_stream_wait": [_vp, _i, _i],
    "fpr_reserve_comm_cus": [_vp, _i],
    "fpr_stream_handle": [_vp, _i, C.POINTER(_vp)],
    "fpr_kernel_timer": [_vp, _i],
    "fpr_kernel_timer_read": [_vp, _i, C.POINTER(_d), C.POINTER(_l)],
    "fpr_diffusion3d_step": [_vp] + [_dp] * 4 + [_i] * 3 + [_d] * 8,
    "fpr_diffusion3d_step_norm": [_vp] + [_dp] * 4 + [_i] * 3 + [_d] * 8 + [_d, _dp],
    "fpr_diffusion3d_step_norm_host": [_vp] + [_dp] * 4 + [_i] * 3 + [_d] * 8 + [_d, C.POINTER(_d)],
    "fpr_diffusion3d_solve": [_vp] + [_dp] * 5 + [_i] * 3 + [_d] * 8 + [_d, _d, _i, _d, _l, _l, _i, C.POINTER(_l), C.POINTER(_d),
                              C.POINTER(_i)],
    "fpr_diffusion3d_step_box": [_vp] + [_dp] * 4 + [_i] * 3 + [_d] * 8 + [C.POINTER(_i), C.POINTER(_i), _d, _dp, _i],
    "fpr_diffusion3d_can_step2": [_vp] + [_dp] * 5 + [_i] * 3,
    "fpr_diffusion3d_step2": [_vp] + [_dp] * 5 + [_i] * 3 + [_d] * 8 + [_d, _dp],
    "fpr_diffusion3d_can_step3": [_vp] + [_dp] * 4 + [_i] * 3,
    "fpr_diffusion3d_step3": [_vp] + [_dp] * 4 + [_i] * 3 + [_d] * 8 + [_d, _dp],
    "fpr_diffusion3d_step2_box": [_vp] + [_dp] * 5 + [_i] * 3 + [_d] * 8 + [C.POINTER(_i), C.POINTER(_i), _d, _dp, _i],
    "fpr_diffusion3d_step2_box2": [_vp] + [_dp] * 5 + [_i] * 3 + [_d] * 8 + [C.POINTER(_i), C.POINTER(_i), _i, _i, _d, _dp, _i],
    "fpr_diffusion3d_step2_core": [_vp] + [_dp] * 5 + [_i] * 3 + [_d] * 8 + [C.POINTER(_i), C.POINTER(_i), _d, _dp, _i, _i, _i],
    "fpr_diffusion3d_step2_halo": [_vp] + [_dp] * 5 + [_i] * 3 + [_d] * 8 + [_d, _dp, _i],
    "fpr_diffusion3d_can_step3_halo": [_vp] + [_dp] * 4 + [_i] * 3,
    "fpr_diffusion3d_step3_halo": [_vp] + [_dp] * 4 + [_i] * 3 + [_d] * 8 + [_d, _dp, _i],
    "fpr_diffusion3d_join": [_vp],
    "fpr_diffusion3d_flux": [_vp] + [_dp] * 4 + [_i] * 3 + [_d] * 4,
    "fpr_diffusion3d_dHdtau": [_vp] + [_dp] * 6 + [_i] * 3 + [_d] * 4,
    "fpr_diffusion3d_update": [_vp] + [_dp] * 2 + [_i] * 3 + [_d],
    "fpr_sumsq_scaled_dev": [_vp, _dp, _z, _d, _dp],
    "fpr_sumsq_scaled": [_vp, _dp, _z, _d, C.POINTER(_d)],
    "fpr_dot": [_vp, _dp, _dp, _z, C.POINTER(_d)],
    "fpr_absmax": [_vp, _dp, _z, C.POINTER(_d)],
    "fpr_copy": [_vp, _dp, _dp, _z],
    "fpr_fill": [_vp, _dp, _d, _z],
    "fpr_fill_on": [_vp, _dp, _d, _z, _i],
    "fpr_add_on": [_vp, _dp, _dp, _z, _i],
    "fpr_init_gaussian3d": [_vp, _dp] + [_i] * 3 + [_d] * 6 + [_i] * 3,
    "fpr_halo_pack3d": [_vp, _dp, _i, _i, _i, _i, _dp, _i],
    "fpr_halo_unpack3d": [_vp, _dp, _i, _i, _i, _i, _dp, _i],
    "fpr_comm_init": [_vp, _i, _i, _vp],
    "fpr_comm_finalize": [_vp],
    "fpr_grid_init": [_vp] + [_i] * 9 + [C.POINTER(_i)] * 4,
    "fpr_grid_info": [_vp, C.POINTER(_i), C.POINTER(_i)],
    "fpr_halo_exchange3d": [_vp, _dp, _i, _i, _i],
    "fpr_halo_exchange3d_begin": [_vp, _dp, _i, _i, _i, _i],
    "fpr_halo_exchange3d_end": [_vp, _dp, _i, _i, _i, _i],
    "fpr_halo_exchange3d_comm": [_vp, _dp, _i, _i, _i, _i],
    "fpr_allreduce_sum_dev": [_vp, _dp, _i, _i],
    "fpr_allreduce_sum1": [_vp, C.POINTER(_d)],
    "fpr_gather3d": [_vp, _dp, _i, _i, _i, _vp],
    "fpr_residual2d": [_vp, _dp, _dp, _d, _d, _dp, _i, _i],
    "fpr_jacobi2d": [_vp, _dp, _dp, _d, _d, _dp, _i, _i, _d, C.POINTER(_d)],
    "fpr_restrict2d": [_vp, _dp, _dp, _i, _i, _i],
    "fpr_prolongate2d": [_vp, _dp, _dp, _i, _i, _i],
    "fpr_axmy2d": [_vp, _dp, _dp, _z],
    "fpr_laplace_apply2d": [_vp, _dp, _d, _d, _d, _dp, _i, _i],
    "fpr_bc_dirichlet2d": [_vp, _dp, _i, _i],
    "fpr_bc_neumann2d": [_vp, _dp, _i, _i],
    "fpr_bc2d": [_vp, _dp, _i, _i],
    "fpr_vcycle2d": [_vp, _dp, _dp, _d, _d, _d, _i, _i, _i, _i, _i, C.POINTER(_d)],
    "fpr_mgsolve2d": [_vp, _dp, _dp, _d, _d, _d, _i, _i, _i, _i, _i, _i, C.POINTER(_d), C.POINTER(_i), C.POINTER(_d),
                      C.POINTER(_d), C.POINTER(_i)],
    "fpr_cg2d": [_vp, _dp, _dp, _d, _d, _d, _d, _i, _i, _i, C.POINTER(_d), C.POINTER(_i)],
    "fpr_comm_init_hosted": [_vp, _i, _i, _vp, _vp, _vp, _vp],
    "fpr_placement_rank": [_vp, C.POINTER(_vp), _i, C.c_size_t, _i, C.POINTER(_i), _i, _vp, _vp, C.POINTER(_i), C.POINTER(_d)],
    "fpr_mg_arena_provide": [_vp, _i, _i, _vp, _vp],
    "fpr_mg_arena_provide_coarse": [_vp, _i, _i, _vp, _vp, _vp],
    "fpr_compute_velocity2d": [_vp, _dp, _d, _d, _dp, _dp, _i, _i],
    "fpr_compute_Ra_dTdx2d": [_vp, _d, _d, _dp, _dp, _i, _i],
    "fpr_compute_diffusion2d": [_vp, _dp, _d, _d, _d, _dp, _i, _i],
    "fpr_compute_advection2d_x": [_vp, _dp, _d, _dp, _dp, _i, _i],
    "fpr_compute_advection2d_y": [_vp, _dp, _d, _dp, _dp, _i, _i],
    "fpr_ns_velocity_max2d": [_vp, _dp, _d, _d, _dp, _dp, _i, _i, C.POINTER(_d)],
    "fpr_ns_rhs2d": [_vp, _dp, _dp, _dp, _d, _d, _i, _i, _d, _d, _d, _d, _d, _dp, _dp],
    "fpr_ns_step2d": [_vp, _vp, _dp, _dp, _dp, _dp, _dp, _i, _i, _d, _d, _d, _d, _d, _d, _d, _i, _i, _i, C.POINTER(_d), C.POINTER(_i)],
    "fpr_ns_run2d": [_vp, _vp, _dp, _dp, _dp, _dp, _dp, _i, _i, _d, _d, _d, _d, _d, _d, _d, _i, _i, _i, _d, _i, C.POINTER(_d),
                     C.POINTER(_i), C.POINTER(_d), C.POINTER(_i)],
}
# every symbol include/fpr.h declares (checked by tests/test_abi.py)
ALL_SYMBOLS = sorted(list(_SIG) + ["fpr_last_error", "fpr_version", "fpr_get_option", "fpr_last_coarse_iters",
                                   "fpr_comm_get_unique_id", "fpr_comm_rank", "fpr_comm_size", "fpr_comm_cus"])


def load_library():
    """dlopen libfpr_hip.so.  No fallback: a missing library is an error."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "libfpr_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path
        )
    L = C.CDLL(path)
    for name, args in _SIG.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = _i
    L.fpr_last_error.argtypes = [_vp]
    L.fpr_last_error.restype = C.c_char_p
    L.fpr_version.argtypes = []
    L.fpr_version.restype = C.c_char_p
    L.fpr_get_option.argtypes = [_vp, C.c_char_p]
    L.fpr_get_option.restype = _l
    L.fpr_last_coarse_iters.argtypes = [_vp]
    L.fpr_last_coarse_iters.restype = _l
    L.fpr_comm_get_unique_id.argtypes = [_vp]
    L.fpr_comm_get_unique_id.restype = _i
    for name in ("fpr_comm_rank", "fpr_comm_size", "fpr_comm_cus"):
        getattr(L, name).argtypes = [_vp]
        getattr(L, name).restype = _i
    _LIB = L
    return L


# ------------------------------------------------------------------------------------------------
# device arrays (Julia layout)
# ------------------------------------------------------------------------------------------------
def _torch():
    import torch

    return torch


def fzeros(*shape, device=None):
    """@zeros(nx, ny[, nz]) -- float64, column-major."""
    torch = _torch()
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    t = torch.zeros(tuple(reversed(shape)), dtype=torch.float64, device=dev)
    return t.permute(*reversed(range(len(shape))))


def fones(*shape, device=None):
    t = fzeros(*shape, device=device)
    t.fill_(1.0)
    return t


def asdevice(a, device=None):
    """Data.Array(host_array): numpy (any order) -> column-major device tensor."""
    torch = _torch()
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    a = np.asarray(a, dtype=np.float64)
    t = torch.from_numpy(np.array(a.T, dtype=np.float64, order="C", copy=True)).to(dev)
    return t.permute(*reversed(range(a.ndim)))


def tonumpy(t):
    """Array(device_array) -> Fortran-ordered numpy array."""
    nd = t.dim()
    return np.asfortranarray(t.permute(*reversed(range(nd))).contiguous().cpu().numpy().T)


def fptr(t, ndim=None):
    """Raw device pointer of a column-major float64 CUDA tensor (validated)."""
    torch = _torch()
    if not (isinstance(t, torch.Tensor) and t.dtype == torch.float64 and t.is_cuda):
        raise TypeError("expected a float64 CUDA tensor")
    if ndim is not None and t.dim() != ndim:
        raise ValueError("expected a %d-D array, got %d-D" % (ndim, t.dim()))
    st = 1
    for n, s in zip(t.shape, t.stride()):
        if n != 1 and s != st:
            raise ValueError("array is not column-major contiguous: shape %s strides %s" % (tuple(t.shape), t.stride()))
        st *= n
    return t.data_ptr()


def follower_value(key, value):
    """... except that cg! never runs as the persistent kernel (cg_fused = 3, the default) on a context whose solves run
    BESIDE another context's: its 16 workgroups synchronise through memory and must all be resident at once."""
    if key == "mg_jacobi_persist":
        return 0
    return 2 if (key == "cg_fused" and int(value) == 3) else int(value)


class Context:
    """fpr_ctx wrapper.  Two torch streams (compute, comm) are created and handed to the library so
    torch ops, torch.cuda.Event timing and the library's kernels share them."""

    def __init__(self, device=0, secondary=False):
        """secondary: a further context on the same device (own streams, own multigrid arena) for work that runs beside the
        default one; it does not become torch's current stream."""
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the fpr hot path needs an MI355X (there is no CPU fallback)")
        self.L = load_library()
        self.device = device
        torch.cuda.set_device(device)
        # One dedicated compute stream becomes torch's CURRENT stream for this process, so torch
        # allocations / fills / copies and the library's kernels are ordered on the same stream.
        torch.cuda.synchronize(device)
        self.compute = torch.cuda.Stream(device=device)
        self._comm0 = torch.cuda.Stream(device=device)
        self._comm, self._comm_handle = self._comm0, None
        self.h = None
        if not secondary:
            torch.cuda.set_stream(self.compute)
        h = _vp()
        rc = self.L.fpr_ctx_create(C.byref(h), device, _vp(self.compute.cuda_stream), _vp(self._comm0.cuda_stream))
        if rc != 0:
            raise FprError(rc, "fpr_ctx_create failed")
        self.h = h
        # scratch device scalars for fused norms
        self.scal = torch.zeros(16, dtype=torch.float64, device=torch.device("cuda", device))
        self._closed = False
        self.comm_ready = False   # fpr_comm_init done (grid.rccl_bootstrap)
        self.opts = {}            # options set through set_option (copied to contexts that follow this one)
        self.followers = []

    def call(self, name, *args):
        rc = getattr(self.L, name)(self.h, *args)
        if rc != 0:
            raise FprError(rc, self.L.fpr_last_error(self.h).decode())
        return rc

    def reserve_comm_cus(self, k):
        """fpr_reserve_comm_cus: the comm stream becomes a library-owned stream on k compute units (0: the torch stream
        given at creation again); self.comm follows, so torch work placed on it lands on the same stream."""
        # Always ask the library (it returns at once when nothing changes: it remembers what was ASKED, a share may be rounded up).
        self.call("fpr_reserve_comm_cus", int(k))

    @property
    def comm(self):
        """The library's CURRENT comm stream as a torch stream (fpr_stream_handle(1)): the torch stream given at creation, or the
        library-owned stream on the reserved compute units.  Looked up at every use -- the one-call pair
        (fpr_diffusion3d_step2_halo) re-splits the device by itself, and a wrapped handle of a stream the library has destroyed
        since must not survive here -- and re-wrapped only when the handle changed."""
        if self.h is None or self._closed:
            return self._comm0
        h = _vp()
        self.call("fpr_stream_handle", 1, C.byref(h))
        if h.value == self._comm0.cuda_stream or not h.value:
            self._comm, self._comm_handle = self._comm0, None
        elif self._comm_handle != h.value:
            torch = _torch()
            self._comm = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", self.device))
            self._comm_handle = h.value
        return self._comm

    def set_option(self, key, value):
        self.call("fpr_set_option", key.encode(), int(value))
        self.opts[key] = int(value)
        for follower in self.followers:      # a second context on the same device runs with the same switches ...
            if not follower._closed:
                follower.set_option(key, follower_value(key, value))

    def get_option(self, key):
        """fpr_get_option: a switch set by set_option, or a diagnostic the library recorded (e.g. diff3_last_bal, comm_units_found)."""
        return int(self.L.fpr_get_option(self.h, key.encode()))

    def synchronize(self):
        self.call("fpr_synchronize")

    def close(self):
        if not self._closed:
            self.L.fpr_ctx_destroy(self.h)
            self._closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
