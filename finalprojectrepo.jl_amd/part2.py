"""NEXT row 8f-1 -- stream-function / vorticity Navier-Stokes step around the V-cycle.
Host mirror of scripts-part2/part2.jl (buoyancy-driven convection, Rayleigh/Prandtl; the reference has
no lid-driven cavity).  Pointwise stencils run in libfpr_hip.so; the handful of whole-array
broadcasts of the reference's driver (part2.jl:193,220,225,229-230) are torch elementwise ops.
"""
import ctypes as C
import enum
import math
import time
from dataclasses import dataclass

import numpy as np

from . import multigrid as mg
from ._lib import asdevice, fptr, fzeros, tonumpy


def _second():
    from . import second_ctx

    return second_ctx()


def _ctx():
    from . import ctx

    return ctx()


class Init_t(enum.Enum):
    """part2.jl:23-27"""

    cosine = 0
    random = 1
    W_from_file = 2


cosine, random, W_from_file = Init_t.cosine, Init_t.random, Init_t.W_from_file


class SimIn_t:
    """part2.jl:30-46"""

    def __init__(self):
        self.k, self.Ra, self.Pr = 1.0, 1.0e6, 1.0e-3
        self.nx, self.ny = 257, 65
        self.ttot, self.beta, self.niters, self.tol = 0.1, 0.0, 50, 1.0e-3
        self.a_dif, self.a_adv = 0.15, 0.4
        self.T_init_strategy, self.W_init_strategy = cosine, random
        self.W_init_file = None  # path of Winit.bin for W_from_file
        self.seed = 1


@dataclass
class SimOut_t:
    """part2.jl:49-55"""

    T: np.ndarray
    W: np.ndarray
    S: np.ndarray
    t_elapsed: float
    timed_iters: float


def splitmix64_uniform(n, seed=1):
    """Counter-based U[0,1): identical on every platform."""
    z = (np.arange(n, dtype=np.uint64) + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def init_array_(M, scheme, h, width, opt=None):
    """part2.jl:58-73"""
    nx, ny = M.shape
    if scheme == cosine:
        i = np.arange(nx, dtype=np.float64)
        col = 0.5 * (1.0 + np.cos((3.0 * math.pi * i * h) / width))
        M.copy_(asdevice(np.repeat(col[:, None], ny, axis=1)))
    elif scheme == random:
        seed = opt.seed if opt is not None else 1
        M.copy_(asdevice(splitmix64_uniform(nx * ny, seed).reshape((nx, ny), order="F")))
    elif scheme == W_from_file:
        M.copy_(asdevice(mg.load(opt.W_init_file)))
    else:
        raise RuntimeError("unknown init scheme")


def compute_velocity_(S, hx, hy, vx, vy):
    """part2.jl:90-96"""
    _ctx().call("fpr_compute_velocity2d", fptr(S, 2), hx, hy, fptr(vx, 2), fptr(vy, 2), *S.shape)


def compute_Ra_dTdx_(Ra, hx, T, Ra_dTdx):
    """part2.jl:99-104"""
    _ctx().call("fpr_compute_Ra_dTdx2d", Ra, hx, fptr(T, 2), fptr(Ra_dTdx, 2), *T.shape)


def compute_diffusion2d_(T, hx, hy, k, dT2):
    """part2.jl:107-113"""
    _ctx().call("fpr_compute_diffusion2d", fptr(T, 2), hx, hy, k, fptr(dT2, 2), *T.shape)


def compute_advection2d_x_(T, hx, vx, dTx):
    """part2.jl:116-125"""
    _ctx().call("fpr_compute_advection2d_x", fptr(T, 2), hx, fptr(vx, 2), fptr(dTx, 2), *T.shape)


def compute_advection2d_y_(T, hy, vy, dTy):
    """part2.jl:128-137"""
    _ctx().call("fpr_compute_advection2d_y", fptr(T, 2), hy, fptr(vy, 2), fptr(dTy, 2), *T.shape)


def _absmax(x):
    out = C.c_double(0.0)
    _ctx().call("fpr_absmax", fptr(x), x.numel(), C.byref(out))
    return out.value


def compute_dt(v, vx, vy, dt_dif, a_dif, a_adv, h, beta):
    """part2.jl:76-87 (v >= 0, so maximum(v) == maximum(abs.(v)))"""
    v_max = _absmax(v)
    if v_max == 0:
        return dt_dif
    mx, my = _absmax(vx), _absmax(vy)   # (Julia: h / 0.0 = Inf)
    dt_adv = a_adv * min(h / mx if mx != 0 else math.inf, h / my if my != 0 else math.inf)
    return dt_adv if beta >= 0.5 else min(dt_dif, dt_adv)


def velocity_and_maxima(S, hx, hy, vx=None, vy=None):
    """Pass 1 of the fused step: compute_velocity! (part2.jl:190) folded into the three maxima compute_dt needs
    (:193-196, :76-87).  Returns (maximum(v), maximum(abs.(vx)), maximum(abs.(vy))); vx / vy are written on request."""
    out = (C.c_double * 3)()
    _ctx().call("fpr_ns_velocity_max2d", fptr(S, 2), hx, hy, fptr(vx, 2) if vx is not None else None,
                fptr(vy, 2) if vy is not None else None, *S.shape, out)
    return out[0], out[1], out[2]


def step_rhs_(T, W, S, hx, hy, Ra, Pr, k, beta, dt, T_out, W_out):
    """Pass 2 of the fused step (part2.jl:202-230): all pointwise terms and the right-hand sides of the two semi-implicit
    solves (beta > 0) or the explicit Euler update (beta == 0), bit-identical to the kernel-by-kernel path."""
    _ctx().call("fpr_ns_rhs2d", fptr(T, 2), fptr(W, 2), fptr(S, 2), hx, hy, *T.shape, Ra, Pr, k, beta, dt, fptr(T_out, 2),
                fptr(W_out, 2))


def navier_stokes_2D(opt=None, verbose=True, do_vis=False, testmode=False, max_steps=None, trace=None, fused=True,
                     timing=None, concurrent_solves=True, native_step=True):
    """part2.jl:140-262.  trace: optional list; one dict per time step is appended with the step's dt and the
    residual histories of its multigrid solves (diagnostics for the parity tests; costs nothing when None).
    fused (default): the step around the three multigrid solves runs as two passes (velocity_and_maxima, step_rhs_)
    instead of the reference's seven kernels, three maxima and four broadcasts -- same numbers, bit for bit.
    timing: optional dict; receives the seconds spent inside the multigrid solves ("mg_s") of the TIMED steps (from the
    fourth on, :182-184), with a stream synchronisation around every solve (diagnostic mode, slightly slower).
    concurrent_solves (default, fused path without trace / timing): the T solve (:221) and the W solve (:226) of a step do not
    depend on each other; W runs on a second context (own streams, own arena) from a worker thread beside T -- these solves
    are bound by launch latency, not by the GPU, so they overlap almost entirely.  Same results bit for bit.
    native_step (default, where concurrent_solves applies): the time loop itself inside the library (fpr_ns_run2d), called in
    pieces where this side has something to do between steps (the clock of :182-184, progress lines) -- the same solves on the
    same inputs, without Python between them, and software-pipelined: the next step's S solve (:187) needs W and nothing of T,
    so it runs behind the W solve on the second context beside the (longest) T solve.  Same T, W, S, dt bit for bit."""
    import torch

    opt = opt if opt is not None else SimIn_t()
    nx, ny = opt.nx, opt.ny
    h = 1.0 / (ny - 1.0)
    width = (nx - 1.0) / (ny - 1.0)
    dt_dif = (opt.a_dif * min(h, h) ** 2) / max(opt.k, opt.Pr)
    names = ("S T T_rhs W W_rhs" if fused else "S vx vy v T dT2 dTx dTy T_rhs W dW2 dWx dWy W_rhs Ra_dTdx").split()
    A = {n: fzeros(nx, ny) for n in names}
    S, T, W = A["S"], A["T"], A["W"]
    vx, vy = A.get("vx"), A.get("vy")
    init_array_(T, opt.T_init_strategy, h, width, opt)
    init_array_(W, opt.W_init_strategy, h, width, opt)
    prealloc = mg.preallocate_buffers(nx, ny)
    hx = hy = h
    ctx = _ctx()
    tic = time.time()
    sim_time, step = 0.0, 0
    mgopt = mg.MGOpt()
    import warnings

    pool, ctx2, ev_a, ev_b = None, None, None, None
    native = bool(native_step and fused and concurrent_solves and opt.beta > 0.0 and trace is None and timing is None)

    while sim_time < opt.ttot:
        if step == 3:
            ctx.synchronize()
            tic = time.time()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore" if not verbose else "default")
            rec = {} if trace is not None else None

            def solve(name, u, f, c, bcs):
                if timing is not None and step >= 3:
                    ctx.synchronize()
                    t_s = time.time()
                    mg.MGsolve_2DPoisson_(u, f, h, c, opt.tol, opt.niters, bcs, opt=mgopt, prealloc_dict=prealloc)
                    ctx.synchronize()
                    timing["mg_s"] = timing.get("mg_s", 0.0) + time.time() - t_s
                    return
                if rec is None:
                    mg.MGsolve_2DPoisson_(u, f, h, c, opt.tol, opt.niters, bcs, opt=mgopt, prealloc_dict=prealloc)
                else:
                    r, hist, frms, cit = mg.MGsolve_2DPoisson_(u, f, h, c, opt.tol, opt.niters, bcs, opt=mgopt,
                                                               prealloc_dict=prealloc, return_history=True)
                    rec[name] = {"r_rms": r, "history": hist, "f_rms": frms, "coarse_iters": cit, "c": c}

            if native:   # :182-249 inside the library, as many steps as nothing on this side has to see one by one
                if ctx2 is None:
                    ctx2 = _second()
                if testmode or verbose:
                    chunk = 1
                else:
                    chunk = (3 - step) if step < 3 else 1 << 30                      # (the clock starts before the fourth step, :182-184)
                    if max_steps is not None:
                        chunk = min(chunk, max_steps - step)
                t_c, n_c, dt_c, bad_c = C.c_double(sim_time), C.c_int(0), C.c_double(0.0), C.c_int(0)
                ctx.call("fpr_ns_run2d", ctx2.h, fptr(S, 2), fptr(T, 2), fptr(W, 2), fptr(A["T_rhs"], 2), fptr(A["W_rhs"], 2), nx, ny,
                         opt.Ra, opt.Pr, opt.k, opt.beta, opt.a_adv, dt_dif, opt.tol, int(opt.niters), int(mgopt.coarse_solve_size),
                         mg.CoarseSolver_t(mgopt.coarse_solver).value, float(opt.ttot), int(max(chunk, 1)), C.byref(t_c), C.byref(n_c),
                         C.byref(dt_c), C.byref(bad_c))
                dt = dt_c.value
                if verbose and bad_c.value:
                    warnings.warn("V-cycle multigrid failed to converge within %d iterations." % opt.niters)  # multigrid.jl:78-80
                sim_time, step = t_c.value, step + n_c.value     # (the library added every step's dt in order, :249)
                if n_c.value == 0:
                    break
            else:
                solve("S", S, W, 0.0, False)  # :187
            if native:
                pass
            elif fused:
                v_max, vx_max, vy_max = velocity_and_maxima(S, hx, hy)  # :190-193
                if v_max == 0:  # compute_dt, :76-87
                    dt = dt_dif
                else:
                    # (Julia: h / 0.0 = Inf; a flow along one axis only must not raise here)
                    dt_adv = opt.a_adv * min(h / vx_max if vx_max != 0 else math.inf, h / vy_max if vy_max != 0 else math.inf)
                    dt = dt_adv if opt.beta >= 0.5 else min(dt_dif, dt_adv)
                mg.apply_boundary_conditions_(T)  # :199
                step_rhs_(T, W, S, hx, hy, opt.Ra, opt.Pr, opt.k, opt.beta, dt, A["T_rhs"], A["W_rhs"])  # :202-230
                if opt.beta > 0.0 and concurrent_solves and rec is None and timing is None:
                    c = 1.0 / (opt.beta * dt)
                    if pool is None:
                        import concurrent.futures

                        pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
                        ctx2 = _second()
                        ev_a, ev_b = torch.cuda.Event(), torch.cuda.Event()
                    ev_a.record(ctx.compute)          # W, W_rhs (and what they were computed from) are complete for the other stream
                    ctx2.compute.wait_event(ev_a)
                    fut = pool.submit(mg.mgsolve_raw, ctx2, W, A["W_rhs"], h, c / opt.Pr, opt.tol, opt.niters, False, mgopt)  # :226
                    resT = mg.mgsolve_raw(ctx, T, A["T_rhs"], h, c, opt.tol, opt.niters, True, mgopt)                      # :221
                    resW = fut.result()
                    ev_b.record(ctx2.compute)         # the next step's kernels (default stream) read W
                    ctx.compute.wait_event(ev_b)
                    for nm, rs in (("T", resT), ("W", resW)):
                        if not rs[3] and verbose:
                            warnings.warn("V-cycle multigrid failed to converge within %d iterations." % opt.niters)  # multigrid.jl:78-80
                elif opt.beta > 0.0:
                    c = 1.0 / (opt.beta * dt)
                    solve("T", T, A["T_rhs"], c, True)  # :221
                    c = c / opt.Pr
                    solve("W", W, A["W_rhs"], c, False)  # :226
                else:  # the explicit update landed in the spare buffers: they become T and W (:229-230)
                    T, A["T_rhs"] = A["T_rhs"], T
                    W, A["W_rhs"] = A["W_rhs"], W
                    A["T"], A["W"] = T, W
            else:
                compute_velocity_(S, hx, hy, vx, vy)  # :190
                torch.sqrt(vx * vx + vy * vy, out=A["v"])  # :193
                dt = compute_dt(A["v"], vx, vy, dt_dif, opt.a_dif, opt.a_adv, h, opt.beta)  # :196
                mg.apply_boundary_conditions_(T)  # :199
                compute_Ra_dTdx_(opt.Ra, hx, T, A["Ra_dTdx"])  # :202
                if not math.isclose(opt.beta, 1.0):  # :205-208
                    compute_diffusion2d_(T, hx, hy, opt.k, A["dT2"])
                    compute_diffusion2d_(W, hx, hy, opt.Pr, A["dW2"])
                compute_advection2d_x_(T, hx, vx, A["dTx"])  # :211-214
                compute_advection2d_y_(T, hy, vy, A["dTy"])
                compute_advection2d_x_(W, hx, vx, A["dWx"])
                compute_advection2d_y_(W, hy, vy, A["dWy"])
                if opt.beta > 0.0:  # :217-226
                    c = 1.0 / (opt.beta * dt)
                    A["T_rhs"].copy_(-c * (T + dt * ((1.0 - opt.beta) * A["dT2"] - A["dTx"] - A["dTy"])))
                    solve("T", T, A["T_rhs"], c, True)  # :221
                    c = c / opt.Pr
                    A["W_rhs"].copy_(-c * (W + dt * ((1.0 - opt.beta) * A["dW2"] - A["dWx"] - A["dWy"] - opt.Pr * A["Ra_dTdx"])))
                    solve("W", W, A["W_rhs"], c, False)  # :226
                else:  # :229-230
                    T.copy_(T + dt * (A["dT2"] - A["dTx"] - A["dTy"]))
                    W.copy_(W + dt * (A["dW2"] - A["dWx"] - A["dWy"] - opt.Pr * A["Ra_dTdx"]))
        if rec is not None:
            rec["dt"] = dt
            trace.append(rec)
        if not native:
            sim_time += dt
            step += 1
        if verbose and (step - 1) % 20 == 0:
            print("time, step: %g %d" % (sim_time, step))
        if testmode or (max_steps is not None and step >= max_steps):
            break
    if pool is not None:
        pool.shutdown(wait=True)
    ctx.synchronize()
    t_elapsed = time.time() - tic
    out = SimOut_t(tonumpy(T), tonumpy(W), tonumpy(S), t_elapsed, step - 3)
    out.dt_last = dt
    out.steps = step
    return out
