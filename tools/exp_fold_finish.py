#!/usr/bin/env python3
"""A/B of an on/off option of the V-cycle (argv[1], default mg_fold_finish; also: mg_mid4).  mg_fold_finish (the finish of cycle k in an extra workgroup row of cycle k+1's first pass below the finest level, against
a launch of its own) in one process: wall time of MGsolve at 4097^2 and 2049^2 (multigrid_bench.jl protocol), best and median of 15."""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fpr_amd

F = fpr_amd.load(0)
mg = F.multigrid
c = F.ctx()
OPT = sys.argv[1] if len(sys.argv) > 1 else "mg_fold_finish"
for n in (4097, 2049, 1025):
    h = 1.0 / (n - 1)
    b = F.asdevice(F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
    x = F.fzeros(n, n)
    res = {0: [], 1: []}
    for rep in range(16):
        for fold in (0, 1):
            c.set_option(OPT, fold)
            x.zero_()
            F.synchronize()
            t0 = time.perf_counter()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, h, 0.0, 1e-6, 100, False, opt=mg.MGOpt(), return_history=True)
            F.synchronize()
            if rep:
                res[fold].append((time.perf_counter() - t0) * 1e6 / len(hist))
    c.set_option(OPT, 1)
    print("%s %d^2, %d cycles: off %.1f / %.1f us per cycle (best / median), on %.1f / %.1f" % (
        OPT, n, len(hist), min(res[0]), float(np.median(res[0])), min(res[1]), float(np.median(res[1]))), flush=True)
