#!/usr/bin/env python3
"""Soak of the two persistent coarse solvers (k_cg_persistent, k_jacobi_persist): the same solves over and over, every result compared bit
for bit with the first one and with the launch-per-iteration forms; a copy stream keeps the memory system busy meanwhile."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fpr_amd

F = fpr_amd.load(0)
mg = F.multigrid
c = F.ctx()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
side = torch.cuda.Stream()
big = [torch.empty(1 << 27, dtype=torch.float64, device="cuda") for _ in range(2)]


def noise():
    with torch.cuda.stream(side):
        for _ in range(4):
            big[0].copy_(big[1])


bad = 0
# ---- cg!: 257^2, consistent system ----
n = 257
b = F.part2.splitmix64_uniform(n * n, 5).reshape((n, n), order="F")
b[0, :] = b[-1, :] = 0.0
b[:, 0] = b[:, -1] = 0.0
gb = F.asdevice(b)
ref = None
t0 = time.time()
for form in (2, 3):        # two launches per iteration (the reference for the bits), then the persistent kernel
    c.set_option("cg_fused", form)
    for i in range(reps if form == 3 else 2):
        if i % 8 == 0:
            noise()
        x = F.fzeros(n, n)
        r, it = mg.cg_(x, gb, 1.0 / 256, 1.0 / 256, 0.0, 1e-10, 2000, return_iters=True)
        got = (r, it, F.tonumpy(x))
        if ref is None:
            ref = got
        elif not (got[0] == ref[0] and got[1] == ref[1] and np.array_equal(got[2], ref[2])):
            bad += 1
            print("cg mismatch: form %d rep %d: it %d vs %d" % (form, i, got[1], ref[1]), flush=True)
c.set_option("cg_fused", 3)
print("cg!: %d solves of %d iterations, %d mismatches, %d barrier time-outs, %.1f s" % (reps + 2, ref[1], bad, c.get_option("cg_persistent_timeouts"), time.time() - t0), flush=True)
# ---- Jacobi coarse solve directly (cap 5140 sweeps) and with an exit inside a launch: 257 x 129 (153 workgroups, groups of 8 sweeps) and
#      257 x 257 (225 workgroups, groups of 7): the data-tagged hand-off (k_jacobi_persist_tag) against plain launches ----
t0 = time.time()
nsolves = 0
for shape in ((257, 129), (257, 257)):
    u0 = F.part2.splitmix64_uniform(shape[0] * shape[1], 7).reshape(shape, order="F")
    f = F.part2.splitmix64_uniform(shape[0] * shape[1], 8).reshape(shape, order="F")
    f[0, :] = f[-1, :] = 0.0
    f[:, 0] = f[:, -1] = 0.0
    gf = F.asdevice(f)
    for tol in (1e-9, 0.05):
        ref = None
        for persist, n_rep in ((0, 1), (1, reps // 4)):
            c.set_option("mg_jacobi_persist", persist)
            for i in range(n_rep):
                if i % 4 == 0:
                    noise()
                gu = F.asdevice(u0)
                r = mg.Vcycle_2DPoisson_(gu, gf, 1.0 / 128, 0.0, tol, 257, mg.jacobi, mg.parallel_shmem, False)
                got = (r, F.tonumpy(gu))
                nsolves += persist
                if ref is None:
                    ref = got
                elif not (abs(got[0] - ref[0]) <= 1e-13 * abs(ref[0]) and np.array_equal(got[1], ref[1])):
                    bad += 1
                    print("jacobi mismatch: shape %s tol %g rep %d: r %.17g vs %.17g, %d cells differ"
                          % (shape, tol, i, got[0], ref[0], int((got[1] != ref[1]).sum())), flush=True)
c.set_option("mg_jacobi_persist", 1)
torch.cuda.synchronize()
print("jacobi: %d persistent solves, %d mismatches in all, %d hand-off time-outs, %.1f s" % (nsolves, bad, c.get_option("mg_jacobi_persist_timeouts"), time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
