#!/usr/bin/env python3
"""Where k_pyr_down spends its time: wall_clock64 stamps (100 MHz) of thread 0 of the middle workgroup (option mg_pyr_prof)."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fpr_amd

F = fpr_amd.load(0)
mg = F.multigrid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
b = F.asdevice(F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F"))
x = F.fzeros(n, n)
prof = torch.zeros(32, dtype=torch.int64, device=x.device)
c = F.ctx()
c.set_option("mg_pyr_down", 1)
c.set_option("mg_pyr_prof", prof.data_ptr())
names = ["load"] + [s % l for l in "PABC" for s in ("%s sweeps", "%s store+residual", "%s halo")]
for rep in range(3):
    x.zero_()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mg.MGsolve_2DPoisson_(x, b, 1.0 / (n - 1), 0.0, 1e-6, 100, False, opt=mg.MGOpt(), return_history=False)
    F.synchronize()
    t = prof.cpu().numpy()
    k = int((t != 0).sum())
    d = [(t[i + 1] - t[i]) / 100.0 for i in range(k - 1)]
    print("n=%d: k_pyr_down %.2f us (last launch of the solve): %s" % (n, (t[k - 1] - t[0]) / 100.0, ", ".join("%s %.2f" % (names[i], v) for i, v in enumerate(d))))
    prof.zero_()
c.set_option("mg_pyr_prof", 0)
c.set_option("mg_pyr_down", 0)
