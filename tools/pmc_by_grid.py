#!/usr/bin/env python3
"""Mean of rocprofv3 --pmc counters per (kernel, grid size): tells the levels of a V-cycle apart (one kernel, one grid size per level).
usage: pmc_by_grid.py <dir> <kernel substr> [<kernel substr> ...]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
subs = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    if not any(s in k for s in subs): continue
    key = (k, int(r["Grid_Size"]), int(r.get("Workgroup_Size", 0) or 0))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
for key in sorted(acc, key=lambda q: (q[0], -q[1])):
    k, g, w = key
    print("%s  grid %d (%d workgroups of %d)" % (k, g, g // max(w, 1), w))
    for n, v in sorted(acc[key].items()): print("   %-22s %.6g (mean of %d dispatches)" % (n, v / cnt[(key, n)], cnt[(key, n)]))
