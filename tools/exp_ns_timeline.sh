#!/bin/bash
# dispatch timeline of the last Navier-Stokes steps at 2049^2 under rocprofv3 --kernel-trace
# usage: tools/exp_ns_timeline.sh <out.txt> <steps> <concurrent 0/1> <last dispatches>
R=$GRAFT_REPO_ROOT; OUT=$R/$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $R/tools/prof_ns.py $2 $3 > $OUT 2>/tmp/tl.err || { tail -20 /tmp/tl.err; exit 1; }
python3 $R/tools/prof_summarize.py overlap /tmp/tl /tmp/tl.txt $4 && cat /tmp/tl.txt >> $OUT
