#!/usr/bin/env python3
"""Does the Navier-Stokes step at 2049^2 depend on where its arrays and the library's arenas lie?  A fresh pair of contexts per round (spacer
allocations of changing size in between), 20 timed steps each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fpr_amd

F = fpr_amd.load(0)
p2 = F.part2
keep = []
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    if rnd:
        keep.append(torch.empty((1 << 30) + (97 << 20) * rnd, dtype=torch.uint8, device="cuda"))
        F.reset()
        F.ctx()
    opt = p2.SimIn_t()
    opt.nx = opt.ny = 2049
    opt.beta, opt.tol, opt.Pr, opt.ttot = 0.5, 1.0e-7, 1.0, 1.0e9
    p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=5, fused=True)
    res = p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=23, fused=True)
    print("round %d: %.4f ms per step (%d timed steps)" % (rnd, res.t_elapsed / max(res.timed_iters, 1) * 1e3, res.timed_iters), flush=True)
