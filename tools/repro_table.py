#!/usr/bin/env python3
"""Table of `python3 bench.py --gpus 1 --steps 20 --warmup 5` runs, each the FIRST command of a fresh lease (the driver's condition):
usage: repro_table.py gpurun_out/r5/repro*.json > profiles/r5_driver_repro.txt"""
import json
import sys

cols = [("card", "gpu_unique_id"), ("kernel_ms", None), ("frac", None), ("value_GBs", None), ("steady_pair_ms", None), ("sclk", "steady_sclk_MHz"),
        ("P_W", "steady_power_W"), ("unplaced", "unplaced_kernel_ms"), ("trial_best", "placement_trial_ms_best"), ("trial_plain", "placement_trial_ms_plain"),
        ("trial_worst", "placement_trial_ms_worst"), ("pool_fast", "pool_fastest_pair_GBs"), ("single_ms", "single_kernel_ms"), ("fma_ms", "fma_kernel_ms"),
        ("fma_steady_pair", None), ("norm", "norm_check_ok"), ("vcycle_ms", None), ("seam_us", "vcycle_seam_us"), ("ns_ms", None)]
rows = []
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    c = d["config"]
    g = lambda k: c.get(k)
    if "gpu_unique_id" not in c:          # lines printed before the scalars were hoisted (round-4 bench.py)
        ds = (d.get("device_state") or {}).get("after_timed_region") or {}
        pp = (d.get("power_probe") or {}).get("fused_pairs") or {}
        fp = c.get("field_placement") or {}
        c = dict(c, gpu_unique_id=ds.get("unique_id"), steady_sclk_MHz=pp.get("sclk_MHz_avg"), steady_power_W=pp.get("power_W_avg"),
                 steady_ms_per_iteration=pp.get("ms_per_iteration"), placement_trial_ms_best=fp.get("trial_ms_best"),
                 placement_trial_ms_worst=fp.get("trial_ms_worst"), pool_fastest_pair_GBs=(fp.get("pair_copy_GBs_all") or {}).get("fastest"),
                 single_kernel_ms=(d.get("roofline_single") or {}).get("kernel_ms"), vcycle_s=(d.get("vcycle") or {}).get("value"),
                 vcycle_seam_us=1e3 * (((d.get("vcycle") or {}).get("roofline") or {}).get("kernels") or {}).get("finest_seam_pass", {}).get("ms", 0),
                 ns_step_s=(d.get("ns_step") or {}).get("value"))
    r = {"card": c.get("gpu_unique_id"), "kernel_ms": d["roofline"]["kernel_ms"], "frac": d["roofline"]["frac"], "value_GBs": d["value"],
         "steady_pair_ms": 2 * c["steady_ms_per_iteration"] if c.get("steady_ms_per_iteration") else None,
         "fma_steady_pair": 2 * c["fma_steady_ms_per_iteration"] if c.get("fma_steady_ms_per_iteration") else None,
         "vcycle_ms": 1e3 * c["vcycle_s"] if c.get("vcycle_s") else None, "ns_ms": 1e3 * c["ns_step_s"] if c.get("ns_step_s") else None}
    for name, key in cols:
        if key:
            r[name] = c.get(key)
    r["file"] = f
    rows.append(r)
fmt = lambda v: "-" if v is None else (("%.4g" % v) if isinstance(v, float) else str(v))
names = ["file"] + [n for n, _ in cols]
w = [max(len(n), max(len(fmt(r.get(n))) for r in rows)) for n in names]
print("  ".join(n.ljust(k) for n, k in zip(names, w)))
for r in rows:
    print("  ".join(fmt(r.get(n)).ljust(k) for n, k in zip(names, w)))
