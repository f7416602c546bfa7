import os, sys, warnings
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import fpr_amd
from fixtures_io import splitmix64_uniform
from oracle.oracle import Oracle, asf, farr
F = fpr_amd.load(0); mg = F.multigrid; oracle = Oracle()
n = 4097; h = 1.0 / (n - 1)
b = asf(splitmix64_uniform(n * n, 1).reshape((n, n), order="F")); gb = F.asdevice(b)
opt = mg.MGOpt(); opt.coarse_solve_size = 257; opt.coarse_solver = mg.conjugate_gradient
for threads in (os.environ.get("OMP_NUM_THREADS", "default"),):
    xo = farr(n, n)
    _, hist_o, frms_o = oracle.mgsolve2d(xo, b, h, 0.0, 1e-6, 7, False, 257, 1)
    cit_o = oracle.last_coarse_iters()
    for form in (3, 2, 1):
        F.ctx().set_option("cg_fused", form)
        x = F.fzeros(n, n)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _, hist, frms, cit = mg.MGsolve_2DPoisson_(x, gb, h, 0.0, 1e-6, 7, False, opt=opt, return_history=True)
        dev = np.abs(np.array(hist) / np.array(hist_o[:len(hist)]) - 1)
        print("omp", threads, "cg_fused", form, "coarse iterations gpu/oracle", cit, cit_o, "cycles", len(hist), len(hist_o), "max rel hist dev", dev.max(), dev.tolist(),
              "field dev", float(np.abs(F.tonumpy(x) - xo).max() / np.abs(xo).max()))
