#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters, largest dispatches only. usage: pmc_kernels.py <dir> <substr> <min_grid>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
sub, ming = sys.argv[2], int(sys.argv[3])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    if sub not in k or int(r["Grid_Size"]) < ming: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()): print("   %-22s %.5g (mean of %d dispatches)" % (n, v / cnt[(k, n)], cnt[(k, n)]))
