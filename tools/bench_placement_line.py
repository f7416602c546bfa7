"""One line about the placement search of a bench.py JSON line.  usage: bench_placement_line.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fp = d["config"]["field_placement"]
print("card %s: kernel %.4f ms (frac %.3f); pool %s -> %s%s; trials %s: best %.4f first %.4f worst %.4f ms; pair copies %s" % (
    d["device_state"]["before_timed_region"].get("unique_id"), d["roofline"]["kernel_ms"], d["roofline"]["frac"], fp.get("pool_first"), fp.get("pool"),
    (" (extended: fastest pair %.0f GB/s)" % fp["pool_extended_because_fastest_pair_GBs"]) if fp.get("pool_extended_because_fastest_pair_GBs") else "",
    fp.get("trials"), fp.get("trial_ms_best", 0), fp.get("trial_ms_first", 0), fp.get("trial_ms_worst", 0),
    {k: round(v) for k, v in fp.get("pair_copy_GBs_all", {}).items()}))
