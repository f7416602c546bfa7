#!/usr/bin/env python3
"""Run MGsolve_2DPoisson! at n^2 (multigrid_bench.jl protocol) a few times -- target for rocprofv3.
usage: prof_mg.py [n] [coarse_solve_size] [jacobi|cg] [repeats]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
css = int(sys.argv[2]) if len(sys.argv) > 2 else 5
solver = sys.argv[3] if len(sys.argv) > 3 else "jacobi"
rep = int(sys.argv[4]) if len(sys.argv) > 4 else 3
F = fpr_amd.load(0)
mg = F.multigrid
for kv in filter(None, os.environ.get("FPR_OPTS", "").split(",")):   # e.g. FPR_OPTS=mg_wave_target=2048,mg_ahead=0
    k, v = kv.split("=")
    F.ctx().set_option(k, int(v))
b = F.asdevice(F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
x = F.fzeros(n, n)
if n == 4097 and os.environ.get("FPR_PLACE", "1") != "0":    # the arrays laid out as bench.py lays them out (benchlegs.place_vcycle_fields)
    rep_ = {}
    import importlib

    x, b = importlib.import_module(F.__name__ + ".benchlegs").place_vcycle_fields(F, n, b, rep_)
    print("placed: trial best %.3f ms, plain allocation %.3f ms, chosen %s" % (rep_.get("trial_ms_best", 0), rep_.get("plain_allocation_ms", 0), rep_.get("chosen")))
opt = mg.MGOpt()
opt.coarse_solve_size = css
opt.coarse_solver = mg.jacobi if solver == "jacobi" else mg.conjugate_gradient
for i in range(rep):
    x.zero_()
    F.synchronize()
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, 1.0 / (n - 1), 0.0, 1e-6, 100, False, opt=opt, return_history=True)
    F.synchronize()
    print("n=%d css=%d %s: %.3f ms, %d cycles, coarse iters %d, rel %.3e" % (n, css, solver, (time.time() - t0) * 1e3, len(hist), cit, r / frms))
