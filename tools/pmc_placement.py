#!/usr/bin/env python3
"""Per-dispatch duration against per-dispatch counters (rocprofv3 --kernel-trace --pmc ... of tools/diffusion_tune f2place): does the time of
k_diff3_march2 on differently placed arrays follow a counter?  usage: pmc_placement.py <dir> <kernel substr>"""
import collections, csv, glob, sys
d, sub = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
rows = collections.OrderedDict()
for r in csv.DictReader(open(cc)):
    if sub not in r["Kernel_Name"]:
        continue
    e = rows.setdefault(r["Dispatch_Id"], {"ms": dur.get(r["Dispatch_Id"])})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if e["ms"] is None and "Start_Timestamp" in r:
        e["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
# groups of consecutive dispatches (one placement = 100 launches): print the mean per group of 100
keys = list(rows)
names = sorted(k for k in rows[keys[0]] if k != "ms")
print("dispatches %d; per group of 100: ms " % len(keys) + " ".join(names))
for g in range(0, len(keys), 100):
    grp = [rows[k] for k in keys[g:g + 100]]
    ms = [x["ms"] for x in grp if x["ms"]]
    print("%5d  %.4f ms  " % (g, sum(ms) / max(len(ms), 1)) + "  ".join("%.4g" % (sum(x.get(n, 0.0) for x in grp) / len(grp)) for n in names))
