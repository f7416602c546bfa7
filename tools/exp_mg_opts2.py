"""MGsolve at 4097^2 (l = 2, Jacobi) under launch options of the fine-level passes.  usage: exp_mg_opts2.py"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fpr_amd
F = fpr_amd.load(0); mg = F.multigrid; c = F.ctx()
n = 4097
b = F.asdevice(F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F")); x = F.fzeros(n, n)
def run():
    ts = []
    for i in range(6):
        x.zero_(); F.synchronize(); t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, 1.0 / (n - 1), 0.0, 1e-6, 100, False, return_history=True)
        F.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts[1:])[2] * 1e3, len(hist), r / frms
base = dict(mg_rows_per_chunk=0, mg_wave_target=4096, mg_nt=0)
for opts in (dict(), dict(mg_rows_per_chunk=16), dict(mg_rows_per_chunk=32), dict(mg_rows_per_chunk=128), dict(mg_wave_target=8192),
             dict(mg_wave_target=16384), dict(mg_nt=1), dict(mg_nt=1, mg_rows_per_chunk=32), dict()):
    for k, v in {**base, **opts}.items(): c.set_option(k, v)
    ms, ncyc, rel = run()
    print("%-40s %.3f ms per solve  %d cycles  rel %.3e" % (opts, ms, ncyc, rel))
