#!/usr/bin/env python3
"""Oracle values for bench.py's N = 1 norm check: the CPU oracle (oracle/fpr_oracle.c, OpenMP build) runs bench.py's single-rank
problem -- n^3 cells, lx = ly = lz = 10, D = 1, dt = 0.2, Gaussian initial state, no physical time step taken -- for the number of
pseudo-iterations the driver's flags (--steps 20 --warmup 5: 1536 + 6 + 20 = 1562) and the default flags (1536 + 20 + 200 = 1756)
reach, and records sum((dHdtau * dt)^2) at those counts (and at 8 and 64) in tests/golden/scale_norms.json under
entries[n<N>_dims1,1,1].oracle_sumsq.  The GPU control values of the same entry (bench.py --golden-norms, tools/make_scale_norms.sh)
must agree with them to 1e-11 (tests/test_oracle_pins.py); the fields are bit-identical up to the 2 ulp of the Gaussian.

    python3 tools/make_n1_norm_pins.py [n] [file]        (n = 512: about 5 minutes on 8 cores)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 16)))
from oracle.oracle import Oracle, farr  # noqa: E402

COUNTS = (8, 64, 1562, 1756)


def oracle_sumsq(n, counts=COUNTS, progress=False):
    orc = Oracle(openmp=True)
    dx = 10.0 / n
    D, dt = 1.0, 0.2
    coef = (dx * dx / D / 8.1, 1.0 / dt, 1 / dx, 1 / dx, 1 / dx, D / dx, D / dx, D / dx)
    Ht = orc.init_gaussian((n, n, n), dx, dx, dx, (5.0, 5.0, 5.0))
    A, B, R = Ht.copy(order="F"), Ht.copy(order="F"), farr(n, n, n)
    out = {}
    t0 = time.time()
    for it in range(1, max(counts) + 1):
        orc.diffusion3d_step(Ht, A, B, R, *coef)
        A, B = B, A
        if it in counts:
            out[str(it)] = orc.sumsq_scaled(R, dt)
            if progress:
                print("n=%d iteration %d sumsq %.17g (%.0f s)" % (n, it, out[str(it)], time.time() - t0), flush=True)
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "scale_norms.json")
    pins = oracle_sumsq(n, progress=True)
    data = json.load(open(path))
    ent = data["entries"].setdefault("n%d_dims1,1,1" % n, {"n": n, "dims": [1, 1, 1], "global_grid": [n, n, n], "sumsq": []})
    ent["oracle_sumsq"] = pins
    ent["oracle_source"] = "tools/make_n1_norm_pins.py: oracle/fpr_oracle.c (OpenMP build), sum((dHdtau*dt)^2) after the given number of pseudo-iterations"
    with open(path, "w") as f:
        json.dump(data, f, indent=0)
        f.write("\n")
