#!/usr/bin/env python3
"""profiles/mg_traffic.json from the --pmc summaries of tools/profile_mg.sh (4097^2, l = 2, Jacobi).
usage: make_mg_traffic_json.py <tag>      (reads profiles/<tag>_mg_pmc_fetch.txt / _mg_pmc_write.txt)
HBM-side bytes per launch = FETCH_SIZE [KiB] x 1024 x 2 (gfx950: FETCH_SIZE reports half of a wide coalesced read,
MI355X_MICROARCH.md section HBM) + WRITE_SIZE [KiB] x 1024; the two counters come from separate passes."""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
n = 4097
N2 = float(n * n)


def mean(path, kernel, counter):
    for line in open(path):
        if kernel in line and counter in line:
            f = line.split()
            i = f.index(counter)
            return float(f[i + 4]), int(f[i + 1])     # max over dispatches: the finest level (coarser levels share the kernel)
    raise SystemExit("no %s for %s in %s" % (counter, kernel, path))


try:
    box = " | ".join(l.strip() for l in open(os.path.join(root, "profiles", tag + "_mg_box.txt")) if l.strip())
except OSError:
    box = None
entries = []
for name, kern, col_min in (("seam", "k_seam_march_v3<false, 6, 2>", 30.0 * N2), ("post", "k_smooth2_march_v2<true, true, false, false>", 26.0 * N2),
                            ("pre", "k_smooth2_march_v2<false, false, true, true>", 28.0 * N2)):
    fe, nf = mean(os.path.join(root, "profiles", tag + "_mg_pmc_fetch.txt"), kern, "FETCH_SIZE")
    wr, nw = mean(os.path.join(root, "profiles", tag + "_mg_pmc_write.txt"), kern, "WRITE_SIZE")
    traffic = fe * 1024 * 2 + wr * 1024
    entries.append({"n": n, "pass": name, "kernel": kern, "FETCH_SIZE_KiB": fe, "WRITE_SIZE_KiB": wr, "fetch_correction": 2.0,
                    "fetch_correction_calibration": "profiles/r5_pmc_calib_width.txt: FETCH_SIZE reports 0.500 of the bytes of an 8-byte-per-lane "
                                                    "streaming read, 0.535 in a row march of this shape; WRITE_SIZE 1.00-1.02", "box": box,
                    "traffic_bytes_per_launch": traffic, "min_bytes_per_launch": col_min, "traffic_over_min_bytes": traffic / col_min,
                    "source": "profiles/%s_mg_pmc_fetch.txt, profiles/%s_mg_pmc_write.txt: separate rocprofv3 --pmc passes of "
                              "`python3 tools/prof_mg.py 4097 5 jacobi 5` (tools/profile_mg.sh), the largest dispatch of the kernel "
                              "(= the finest level); FETCH_SIZE doubled per MI355X_MICROARCH.md; not measured in the run that "
                              "prints this line" % (tag, tag)})
json.dump({"entries": entries}, open(os.path.join(root, "profiles", "mg_traffic.json"), "w"), indent=1)
print(json.dumps(entries, indent=1))
