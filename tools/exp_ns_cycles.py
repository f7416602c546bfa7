#!/usr/bin/env python3
"""Cycles per multigrid solve in the Navier-Stokes step of bench.py's ns_block (2049^2, beta = 0.5, tol 1e-7)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd

F = fpr_amd.load(0)
p2 = F.part2
opt = p2.SimIn_t()
opt.nx = opt.ny = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
opt.beta, opt.tol, opt.Pr, opt.ttot = 0.5, 1.0e-7, 1.0, 1.0e9
tr = []
res = p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=9, fused=True, trace=tr)
for i, rec in enumerate(tr):
    print(i, {k: (len(v["history"]), v["coarse_iters"]) for k, v in rec.items() if isinstance(v, dict)}, "dt %.3e" % rec["dt"])
