"""A few Navier-Stokes steps at 2049^2 (bench.ns_block's configuration) for a dispatch timeline under rocprofv3 --kernel-trace.
usage: prof_ns.py [steps] [concurrent 0/1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd
F = fpr_amd.load(0)
p2 = F.part2
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
conc = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
opt = p2.SimIn_t()
opt.nx = opt.ny = 2049
opt.beta, opt.tol, opt.Pr, opt.ttot = 0.5, 1.0e-7, 1.0, 1.0e9
p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=5, fused=True, concurrent_solves=conc)
res = p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=steps, fused=True, concurrent_solves=conc)
F.synchronize()
print("NS step %.3f ms (%d timed steps, concurrent=%d)" % (res.t_elapsed / max(res.timed_iters, 1) * 1e3, res.timed_iters, conc))
