#!/usr/bin/env python3
"""profiles/diffusion_traffic.json from the --pmc summaries of tools/profile_bench.sh.
usage: make_traffic_json.py <tag>      (reads profiles/<tag>_pmc_fetch.txt / _pmc_write.txt)
HBM-side bytes per launch = FETCH_SIZE [KiB] x 1024 x 2 (gfx950: FETCH_SIZE reports half of a wide coalesced read,
MI355X_MICROARCH.md section HBM) + WRITE_SIZE [KiB] x 1024; the two counters come from separate passes."""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
n = 512
cells = (n - 2) ** 3


def mean(path, kernel, counter):
    for line in open(path):
        if kernel in line and counter in line:
            f = line.split()
            i = f.index(counter)
            return float(f[i + 2]), int(f[i + 1])
    raise SystemExit("no %s for %s in %s" % (counter, kernel, path))


try:
    box = " | ".join(l.strip() for l in open(os.path.join(root, "profiles", tag + "_box.txt")) if l.strip())
except OSError:
    box = None
entries = []
for fused, kern in ((3, "k_diff3_march3<true, true>"), (True, "k_diff3_march2<true, 8, true, false, false>"), (False, "k_diff3_march<")):
    fe, nf = mean(os.path.join(root, "profiles", tag + "_pmc_fetch.txt"), kern, "FETCH_SIZE")
    wr, nw = mean(os.path.join(root, "profiles", tag + "_pmc_write.txt"), kern, "WRITE_SIZE")
    traffic = fe * 1024 * 2 + wr * 1024
    entries.append({"n": n, "fuse2": fused is True, "depth": 3 if fused == 3 else (2 if fused else 1), "kernel": kern.rstrip("<") if kern.endswith("<") else kern, "FETCH_SIZE_KiB_mean": fe, "fetch_dispatches": nf,
                    "WRITE_SIZE_KiB_mean": wr, "write_dispatches": nw, "fetch_correction": 2.0, "box": box,
                    "traffic_bytes_per_launch": traffic, "min_bytes_per_launch": 32.0 * cells,
                    "traffic_over_min_bytes": traffic / (32.0 * cells),
                    "source": "profiles/%s_pmc_fetch.txt, profiles/%s_pmc_write.txt: separate rocprofv3 --pmc passes of "
                              "`python3 bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20` "
                              "(tools/profile_bench.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md; not measured in "
                              "the run that prints this line" % (tag, tag)})
json.dump({"entries": entries}, open(os.path.join(root, "profiles", "diffusion_traffic.json"), "w"), indent=1)
print(json.dumps(entries, indent=1))
