#!/bin/bash
# durations of the fused kernel through a whole default bench run, in dispatch order (which phase of the run sees which kernel time?)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $R/gpurun_out/phases_bench.json 2> /tmp/pp.err || { tail -5 /tmp/pp.err; exit 1; }
python3 - <<'PY'
import csv, glob, os
f = glob.glob('/tmp/pp/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    if 'k_diff3_march2' in r['Kernel_Name'] or 'k_diff3_march<' in r['Kernel_Name']:
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:48]))
rows.sort()
t0 = rows[0][0]
out = open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/phases.txt', 'w')
i = 0
while i < len(rows):
    name = rows[i][2]
    j = i
    while j < len(rows) and rows[j][2] == name and j - i < 40 and (j == i or rows[j][0] - rows[j-1][1] < 5e6): j += 1
    d = [(e - s) / 1e3 for s, e, _ in rows[i:j]]
    out.write("t=%8.1f ms  %-48s n=%3d  avg %.1f  min %.1f  max %.1f us\n" % ((rows[i][0] - t0) / 1e6, name, j - i, sum(d) / len(d), min(d), max(d)))
    i = j
out.close()
PY
