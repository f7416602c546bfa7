#!/usr/bin/env python3
"""Where the persistent cg! kernel spends an iteration: wall_clock64 ticks (100 MHz) of workgroup 0 per section, summed over
all iterations of one solve at n^2 (option cg_prof = device address of 8 int64)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fpr_amd

F = fpr_amd.load(0)
mg = F.multigrid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 257
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 600
b = F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F")
b[0, :] = b[-1, :] = 0.0
b[:, 0] = b[:, -1] = 0.0
prof = torch.zeros(8, dtype=torch.int64, device="cuda")
F.ctx().set_option("cg_prof", prof.data_ptr())
names = ["beta/p/ring/LDS", "operator", "-", "sum+barrier 1", "update", "sum+barrier 2"]
for rep in range(9):
    F.ctx().set_option("cg_persistent_wgs", 16 if rep >= 6 else 64)
    F.ctx().set_option("cg_tagged_edges", 0 if 3 <= rep < 6 else 1)       # r edges as data-tagged granules (option) / sc1 + drain + flag form (default)
    if rep in (0, 3, 6):
        print("--- %d workgroups, %s ---" % (16 if rep >= 6 else 64, "edges as sc1 stores + drain (round 4)" if 3 <= rep < 6 else "edges as tagged granules"))
    prof.zero_()
    x = F.fzeros(n, n)
    r, it = mg.cg_(x, F.asdevice(b), 1.0 / (n - 1), 1.0 / (n - 1), 0.0, 1e-12, nmax, return_iters=True)
    F.synchronize()
    t = prof.cpu().numpy()[:6] / 100.0
    print("n=%d: %d iterations, %.2f us per iteration: %s" % (n, it, t.sum() / max(it, 1), ", ".join("%s %.2f" % (a, v / max(it, 1)) for a, v in zip(names, t))))
F.ctx().set_option("cg_prof", 0)
F.ctx().set_option("cg_persistent_wgs", 64)
F.ctx().set_option("cg_tagged_edges", 0)
