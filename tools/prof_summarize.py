#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output directories into small text files (the raw traces are too big to keep).

    prof_summarize.py stats  <dir> <out.txt>        kernel_stats.csv -> per-kernel calls / avg / total
    prof_summarize.py pmc    <dir> <out.txt>        counter_collection.csv -> per-kernel mean counter value
    prof_summarize.py timeline <dir> <out.txt> [n]  kernel_trace.csv -> the last n dispatches with durations and gaps
    prof_summarize.py overlap  <dir> <out.txt> [n]  kernel_trace.csv -> the last n dispatches with absolute start / end, queue, grid
    prof_summarize.py tailstats <dir> <out.txt> <marker> <n>   kernel_trace.csv -> per-kernel calls / avg over the dispatches from the n-th last
                                                    dispatch of a kernel whose name contains <marker> on (e.g. the last 5 solves of a run that
                                                    searched a placement first)
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def stats(d, out):
    f = find(d, "*kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    with open(out, "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats summary (%s)\n" % os.path.basename(f))
        o.write("%-70s %8s %14s %14s %8s\n" % ("kernel", "calls", "avg_ns", "total_ns", "pct"))
        for r in rows:
            o.write("%-70s %8s %14s %14s %8s\n" % (r["Name"][:70], r["Calls"], r.get("AverageNs", r.get("AverageDurationNs", "?")),
                                                  r.get("TotalDurationNs", r.get("Total(ns)", "?")), r.get("Percentage", "?")))


def timeline(d, out, last=40):
    """The last `last` dispatches of the trace in start order: duration and the idle gap since the previous one ended."""
    f = find(d, "*kernel_trace.csv")
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-last:]
    with open(out, "w") as o:
        o.write("# rocprofv3 --kernel-trace timeline (%s): last %d dispatches\n" % (os.path.basename(f), len(rows)))
        o.write("%-64s %12s %10s %10s\n" % ("kernel", "grid", "dur_us", "gap_us"))
        prev = None
        for r in rows:
            st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            grid = "%sx%s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", "?"))
            o.write("%-64s %12s %10.2f %10.2f\n" % (r["Kernel_Name"][:64], grid, (en - st) / 1e3, (st - prev) / 1e3 if prev else 0.0))
            prev = en


def overlap(d, out, last=40):
    """The last `last` dispatches in start order with absolute start / end (us since the first of them), queue, grid,
    workgroup size, VGPRs and LDS: shows which kernels ran side by side."""
    f = find(d, "*kernel_trace.csv")
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-last:]
    t0 = int(rows[0]["Start_Timestamp"])
    with open(out, "w") as o:
        o.write("# rocprofv3 --kernel-trace (%s): last %d dispatches, times in us since the first of them\n" % (os.path.basename(f), len(rows)))
        o.write("%-56s %5s %9s %5s %5s %7s %10s %10s %9s\n" % ("kernel", "queue", "grid", "wg", "vgpr", "lds", "start", "end", "dur"))
        for r in rows:
            st, en = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            o.write("%-56s %5s %9s %5s %5s %7s %10.1f %10.1f %9.1f\n" % (
                r["Kernel_Name"][:56], r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")),
                r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"), st / 1e3, en / 1e3, (en - st) / 1e3))


def tailstats(d, out, marker, n):
    f = find(d, "*kernel_trace.csv")
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    start = marks[-n] if len(marks) >= n else 0
    acc = defaultdict(list)
    for r in rows[start:]:
        acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in acc.values())
    with open(out, "w") as o:
        o.write("# rocprofv3 --kernel-trace (%s): the dispatches from the %d-th last `%s` on (%d of %d)\n" % (os.path.basename(f), n, marker, len(rows) - start, len(rows)))
        o.write("%-70s %8s %14s %14s %8s\n" % ("kernel", "calls", "avg_ns", "total_ns", "pct"))
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            o.write("%-70s %8d %14.1f %14d %8.2f\n" % (k[:70], len(v), sum(v) / len(v), sum(v), 100.0 * sum(v) / tot))


def pmc(d, out):
    f = find(d, "*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(out, "w") as o:
        o.write("# rocprofv3 --pmc summary (%s): mean counter value per dispatch\n" % os.path.basename(f))
        o.write("%-70s %-14s %8s %18s %18s %18s\n" % ("kernel", "counter", "n", "mean", "min", "max"))
        for k, cs in sorted(acc.items()):
            for c, v in sorted(cs.items()):
                o.write("%-70s %-14s %8d %18.1f %18.1f %18.1f\n" % (k[:70], c, len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    if sys.argv[1] == "timeline":
        timeline(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 40)
    elif sys.argv[1] == "tailstats":
        tailstats(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]))
    elif sys.argv[1] == "overlap":
        overlap(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 40)
    else:
        {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
