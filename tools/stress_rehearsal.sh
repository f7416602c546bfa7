#!/bin/bash
# Repeat the four-rank shared-card rehearsal of bench.py for each decomposition and print the norm rank 0 reports with all
# digits: every run of one decomposition must print the same number (the kernels and the reduction orders are
# deterministic), and the three decompositions agree to rounding.  A deviation would mean a missing dependency between the
# streams / processes of the shell-core choreography.
R=${GRAFT_REPO_ROOT:-.}
N=${1:-6}
for d in 2,2,1 2,1,2 1,2,2 1,1,4; do
  for i in $(seq 1 $N); do
    timeout -k 10 120 python $R/bench.py --gpus 4 --rehearse-shared-gpu --dims $d --n 128 --steps 12 --warmup 4 --no-cpu-baseline --no-secondary --no-single-leg --prewarm-ms 0 2>/tmp/stress.err | python3 -c "
import sys, json
ok = False
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$d', repr(d['config']['last_err'])); ok = True
if not ok: print('$d', 'NO JSON LINE')
" || { echo "run failed"; tail -5 /tmp/stress.err; }
  done
done
