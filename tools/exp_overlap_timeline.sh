#!/bin/bash
# dispatch timeline (absolute start / end per kernel) of the fused-pair choreography with neighbours, one rank that is
# its own periodic neighbour over the library's RCCL transport
# usage: tools/exp_overlap_timeline.sh <out.txt> <n> <periods> "<opts1>" "<opts2>" ...   (opts: k=v,k=v or "none")
R=$GRAFT_REPO_ROOT; OUT=$R/$1; N=$2; PER=$3; shift 3
cd /tmp && export TMPDIR=/tmp
: > $OUT
for O in "$@"; do
  rm -rf /tmp/tl
  echo "=== options: $O" >> $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $R/tools/exp_overlap.py $N $PER 6 "$O" >> $OUT 2>/tmp/tl.err || { tail -20 /tmp/tl.err; exit 1; }
  python3 $R/tools/prof_summarize.py overlap /tmp/tl /tmp/tl.txt 28 && cat /tmp/tl.txt >> $OUT
done
