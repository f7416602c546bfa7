#!/usr/bin/env python3
"""NS step at 2049^2 (bench.py's ns_block protocol) with the T and W solves one after the other / side by side."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd

F = fpr_amd.load(0)
p2 = F.part2
for conc in (False, True, False, True):
    opt = p2.SimIn_t()
    opt.nx = opt.ny = 2049
    opt.beta, opt.tol, opt.Pr, opt.ttot = 0.5, 1.0e-7, 1.0, 1.0e9
    res = p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=23, fused=True, concurrent_solves=conc)
    print("concurrent_solves=%s: %.3f ms per step (%d timed steps)" % (conc, res.t_elapsed / max(res.timed_iters, 1) * 1e3, res.timed_iters))
