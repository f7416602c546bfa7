"""Soak of the Navier-Stokes time loop inside the library (fpr_ns_run2d: pipelined, two contexts, worker thread) against the loop
composed from Python with the solves one after the other: T, W, S, dt and step count bit for bit after N steps, R repeats, at several
sizes; also chunked calls (1, 2, 5 steps per call).  usage: soak_ns.py [steps] [repeats]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fpr_amd
F = fpr_amd.load(0)
p2 = F.part2
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bad = 0
for shape in ((513, 257), (1025, 1025), (257, 129), (2049, 513)):
    def run(**kw):
        opt = p2.SimIn_t()
        opt.nx, opt.ny, opt.beta, opt.tol, opt.Pr, opt.niters, opt.ttot = shape[0], shape[1], 0.5, 1.0e-7, 1.0, 30, 1e9
        opt.W_init_strategy = p2.random
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return p2.navier_stokes_2D(opt=opt, verbose=False, max_steps=N, fused=True, **kw)
    t0 = time.time()
    ref = run(concurrent_solves=False)
    t_ref = time.time() - t0
    mism = 0
    for r in range(R):
        t0 = time.time()
        got = run()
        t_nat = time.time() - t0
        ok = got.dt_last == ref.dt_last and got.steps == ref.steps and all(np.array_equal(getattr(got, n), getattr(ref, n)) for n in ("T", "W", "S"))
        mism += not ok
    bad += mism
    print("%dx%d: %d steps, %d repeats of the pipelined loop against the sequential composition: %d mismatching (%.2f s against %.2f s)"
          % (shape[0], shape[1], N, R, mism, t_nat, t_ref))
sys.exit(1 if bad else 0)
