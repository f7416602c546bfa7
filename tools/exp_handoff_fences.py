#!/usr/bin/env python3
"""What it costs to keep the neighbour hand-offs of the two persistent coarse solvers INSIDE the HIP memory model.
Five-level V-cycle (4097^2, coarse 257^2), Jacobi coarse solve (20 x 257 sweeps per cycle) and cg!:
    jacobi  tagged granules (default)            mg_jacp_tagged = 1, handoff_fences = 0
            flags, sc1 stores + drains           mg_jacp_tagged = 0, handoff_fences = 0
            flags + agent-scope release/acquire  handoff_fences = 1
    cg!     default / tagged edges / fences
Event time of the coarse-solver launches per sweep / per CG iteration, wall time per V-cycle, best of 3.   usage: exp_handoff_fences.py [cycles]"""
import ctypes as C
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd

F = fpr_amd.load(0)
mg = F.multigrid
ctx = F.ctx()
n = 4097
h = 1.0 / (n - 1)
b = F.asdevice(F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
x = F.fzeros(n, n)
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def run(solver, kind, opts):
    for k, v in opts.items():
        ctx.set_option(k, v)
    opt = mg.MGOpt()
    opt.coarse_solve_size, opt.coarse_solver = 257, solver
    best = None
    for _ in range(3):
        x.zero_()
        F.synchronize()
        ctx.call("fpr_kernel_timer", 1)
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, cycles, False, opt=opt, return_history=True)
        F.synchronize()
        dt = (time.perf_counter() - t0) / len(hist)
        tot, cnt = C.c_double(0.0), C.c_long(0)
        ctx.call("fpr_kernel_timer_read", kind, C.byref(tot), C.byref(cnt))
        ctx.call("fpr_kernel_timer", 0)
        if best is None or dt < best[0]:
            best = (dt, tot.value, cnt.value, cit, list(hist))
    return best


base = {}
for name, solver, kind, opts in (
        ("jacobi tagged granules (default)", mg.jacobi, 6, {"mg_jacp_tagged": 1, "handoff_fences": 0}),
        ("jacobi flags, no fences", mg.jacobi, 6, {"mg_jacp_tagged": 0, "handoff_fences": 0}),
        ("jacobi flags + release/acquire", mg.jacobi, 6, {"mg_jacp_tagged": 0, "handoff_fences": 1}),
        ("jacobi tagged asked, fences on", mg.jacobi, 6, {"mg_jacp_tagged": 1, "handoff_fences": 1}),
        ("cg default", mg.conjugate_gradient, 5, {"cg_tagged_edges": 0, "handoff_fences": 0}),
        ("cg tagged edges", mg.conjugate_gradient, 5, {"cg_tagged_edges": 1, "handoff_fences": 0}),
        ("cg + release/acquire", mg.conjugate_gradient, 5, {"cg_tagged_edges": 0, "handoff_fences": 1})):
    dt, tot, cnt, cit, hist = run(solver, kind, opts)
    key = name.split(" ")[0]
    base.setdefault(key, (dt, hist))
    dev = max(abs(a - c) / abs(c) for a, c in zip(hist, base[key][1]))
    print("%-36s %.3f ms per V-cycle (%+.1f %%), %.3f us per %s (%d), history vs first %.1e, timeouts jac %d cg %d"
          % (name, dt * 1e3, 100.0 * (dt / base[key][0] - 1.0), tot * 1e3 / max(cit, 1), "sweep" if key == "jacobi" else "CG iteration", cit, dev,
             ctx.get_option("mg_jacobi_persist_timeouts"), ctx.get_option("cg_persistent_timeouts")), flush=True)
ctx.set_option("mg_jacp_tagged", 1)
ctx.set_option("handoff_fences", 0)
ctx.set_option("cg_tagged_edges", 0)
