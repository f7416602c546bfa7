import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd
F = fpr_amd.load(0)
for rep in range(2):
    rows = F.experiments.part1_scaling_experiments(None, n=128, ttot=2.0, tol=1e-6, strong_scaling_modes=(True,), shared_memory_modes=(True,))
    r = rows[0]
    print("128^3 reference protocol: delta_t %.3f s, Work %.0f, %d iterations total -> %.1f us per timed iteration" % (r["delta_t"], r["Work"], r["_iters"], r["delta_t"] / (r["Work"] / (27 * 126 ** 3)) * 1e6))
