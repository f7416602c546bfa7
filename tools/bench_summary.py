"""Headline numbers of a bench.py JSON line.  usage: bench_summary.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.1f GB/s, %.4f ms per step; %s %.4f ms = %.3f of peak" % (d["value"], d["ms_per_step"], r["kernel"].split(" ")[0], r["kernel_ms"], r["frac"]))
for k in ("vcycle", "vcycle_5levels", "ns_step"):
    v = d.get(k)
    if isinstance(v, dict) and "value" in v:
        rf = v.get("roofline") or {}
        print("%s: %.4f ms%s" % (k, v["value"] * 1e3, (", %s %.3f" % (str(rf.get("kernel", ""))[:24], rf["frac"])) if rf.get("frac") else ""))
        if k == "ns_step":
            print("   composed from Python %.4f ms, solves in sequence %.4f ms" % (v.get("composed_from_python_s_per_step", 0) * 1e3, v["solves_one_after_the_other_s_per_step"] * 1e3))
for k, v in (d.get("legs") or {}).items():
    if "over_plain_pair" in v:
        print("%s: pair %.1f us = %.3f x plain (comm units %s)" % (k, v["pair_ms"] * 1e3, v["over_plain_pair"], v.get("comm_units")))
print("cpu_baseline:", {k: v for k, v in (d.get("cpu_baseline") or {}).items() if k in ("value", "unit", "cores", "kind")})
