"""Two levels in two launches (k_mid_down<2> / k_mid_up<2>, option mg_mid2) against one marching pass per level and direction:
MGsolve at 4097^2 (l = 2, Jacobi; per V-cycle) and the Navier-Stokes step at 2049^2.  usage: exp_mid2.py"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fpr_amd, bench
F = fpr_amd.load(0); mg = F.multigrid; c = F.ctx()
n = 4097
b = F.asdevice(F.part2.splitmix64_uniform(n * n, 1).reshape((n, n), order="F")); x = F.fzeros(n, n)
def run():
    ts = []
    for i in range(7):
        x.zero_(); F.synchronize(); t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r, hist, frms, cit = mg.MGsolve_2DPoisson_(x, b, 1.0 / (n - 1), 0.0, 1e-6, 100, False, return_history=True)
        F.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts[1:])[2] * 1e3, len(hist), r / frms, F.tonumpy(x)
ref = None
for rep in range(2):
    for opts in (dict(mg_mid2=0), dict(mg_mid2=1), dict(mg_mid2=1, mg_mid2_tile=1), dict(mg_mid2=2)):
        for k, v in {**dict(mg_mid2_tile=0), **opts}.items(): c.set_option(k, v)
        ms, ncyc, rel, xs = run()
        if ref is None: ref = xs
        print("%-34s %.3f ms per solve  %d cycles  %.4f ms per cycle  rel %.3e  field equal: %s" % (opts, ms, ncyc, ms / ncyc, rel, np.array_equal(xs, ref)), flush=True)
del b, x
for opts in (dict(mg_mid2=0), dict(mg_mid2=1), dict(mg_mid2=2), dict(mg_mid2=2, mg_mid2_tile=1), dict(mg_mid2=0)):
    for cc in (c, F.second_ctx()):
        for k, v in {**dict(mg_mid2_tile=0), **opts}.items(): cc.set_option(k, v)
    r = bench.ns_block(F)
    print("NS 2049^2 %-30s side by side %.3f ms, in sequence %.3f ms" % (opts, r["value"] * 1e3, r["solves_one_after_the_other_s_per_step"] * 1e3), flush=True)
