"""Balanced form of the fused kernel (fpr_diffusion3d_step2_core) against the static grid on ONE rank at n^3, no
neighbours: time per launch for several reserves (workgroups = 256 - reserve).  usage: exp_bal.py [n] [z-lo z-hi]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fpr_amd
F = fpr_amd.load(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dx = 10.0 / n
coef = (dx * dx / 8.1, 5.0, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx, 1 / dx)
Ht = F.fzeros(n, n, n); F.part1.init_local_gaussian((5., 5., 5.), dx, dx, dx, Ht)
A, O, C, R, sq = Ht.clone(), F.fzeros(n, n, n), Ht.clone(), F.fzeros(n, n, n), F.fzeros(2)
lo, hi = (1, 1, 2), (n - 1, n - 1, n - 2)
def run(reserve, reps):
    for _ in range(reps):
        if reserve is None:
            F.part1.diffusion_3D_step_τ2_box(Ht, A, O, C, R, *coef, lo, hi, 0.2, sq)
        else:
            F.part1.diffusion_3D_step_τ2_core(Ht, A, O, C, R, *coef, lo, hi, 0.2, sq, 0, reserve)
for rnd in range(2):
    for reserve in (None, 1, 4, 8, 12, 16, 32):
        run(reserve, 5); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(reserve, 30); torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 30
        print("round %d  %s: %.1f us per launch" % (rnd, "static grid (255 workgroups)" if reserve is None else "balanced, reserve %2d (%d workgroups)" % (reserve, 256 - reserve), t * 1e6))
