"""NS step at 2049^2 against the seam pass's chunking (mg_seam_wg_per_cu) and the wave target of the two-sweep passes (mg_wave_target).
usage: exp_ns_seam_chunks.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd
F = fpr_amd.load(0)
p2 = F.part2
opt = p2.SimIn_t()
opt.nx = opt.ny = 2049
opt.beta, opt.tol, opt.Pr, opt.ttot = 0.5, 1.0e-7, 1.0, 1.0e9
def step(steps=23):
    res = p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=steps, fused=True, concurrent_solves=True)
    return res.t_elapsed / max(res.timed_iters, 1) * 1e3
step(5)
print("default: %.3f %.3f ms" % (step(), step()))
for name, vals in (("mg_seam_wg_per_cu", (1, 2, 3, 4, 6, 8)), ("mg_wave_target", (1024, 2048, 4096, 8192, 16384)), ("mg_seam_rows_per_chunk", (2050, 1026, 684, 514, 342, 258, 130, 66))):
    for v in vals:
        F.ctx().set_option(name, v)
        F.synchronize()
        # a new arena geometry per option: the contexts rebuild theirs on the next solve
        print("%s=%d: %.3f %.3f ms" % (name, v, step(), step()))
    F.ctx().set_option(name, 0)
print("default again: %.3f ms" % step())
