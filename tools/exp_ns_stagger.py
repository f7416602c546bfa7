"""NS step at 2049^2 against the start offset of the T solve beside the W + S side (ns_stagger_us).  usage: exp_ns_stagger.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd
F = fpr_amd.load(0)
p2 = F.part2
opt = p2.SimIn_t()
opt.nx = opt.ny = 2049
opt.beta, opt.tol, opt.Pr, opt.ttot = 0.5, 1.0e-7, 1.0, 1.0e9
def step(steps=43):
    res = p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=steps, fused=True)
    return res.t_elapsed / max(res.timed_iters, 1) * 1e3
step(8)
for us in (0, 10, 20, 30, 40, 50, 60, 70, 85, 100, 0):
    F.ctx().set_option("ns_stagger_us", us)
    print("ns_stagger_us=%3d: %.4f %.4f ms per step" % (us, step(), step()), flush=True)
