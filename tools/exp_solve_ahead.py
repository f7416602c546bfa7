#!/usr/bin/env python3
"""fpr_diffusion3d_solve with the host waiting for every norm (diff3_ahead = 0) against pairs enqueued ahead of the host with
the exit test on the device: the reference's own benchmark protocol (part1_scaling_experiments.jl: n^3, tol 1e-6, ttot 2,
convergence check every iteration) at 128^3 and a shorter run at 512^3."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd

F = fpr_amd.load(0)
c = F.ctx()
for n, ttot in ((128, 2.0), (256, 0.4), (512, 0.2)):
    for ahead in (0, 2, 0, 2):
        c.set_option("diff3_ahead", ahead)
        F.synchronize()
        t0 = time.perf_counter()
        _, H, _, info = F.part1.diffusion_3D_kernel_programming(nx=n, ny=n, nz=n, ttot=ttot, tol=1e-6, verbose=False)
        F.synchronize()
        dt = time.perf_counter() - t0
        its = sum(info["iters"])
        print("n=%d ahead=%d: %.4f s, %d iterations, %.2f us per iteration" % (n, ahead, dt, its, dt / its * 1e6))
c.set_option("diff3_ahead", 2)
