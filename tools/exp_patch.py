import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd
F = fpr_amd.load(0); mg = F.multigrid
n = 257
b = F.asdevice(F.part2.splitmix64_uniform(n*n, 1).reshape((n, n), order="F"))
for dbg in (0, 1, 2):
  F.ctx().set_option('mg_patch_dbg', dbg)
  print('dbg', dbg)
  for Sg in (1, 8):
    F.ctx().set_option("mg_group_sweeps", Sg)
    for rep in range(2):
        x = F.fzeros(n, n); F.synchronize(); t0 = time.time()
        r = mg.Vcycle_2DPoisson_(x, b, 1.0/(n-1), 0.0, 1e-12, 257, mg.jacobi, mg.parallel, False)
        F.synchronize(); dt = time.time() - t0
    print('  ', end=''); print("Sg=%d: %.2f ms for 5140 sweeps -> %.2f us/launch, %.3f us/sweep" % (Sg, dt*1e3, dt*1e6/((5140+Sg-1)//Sg), dt*1e6/5140))
