#!/bin/bash
# dispatch timeline of the last V-cycles of an MGsolve under rocprofv3 --kernel-trace, for a list of option sets
# usage: tools/exp_mg_timeline.sh <out.txt> <n> <css> "<opts1>" "<opts2>" ...   (opts: k=v,k=v or "none")
R=$GRAFT_REPO_ROOT; OUT=$R/$1; N=$2; CSS=$3; shift 3
cd /tmp && export TMPDIR=/tmp
: > $OUT
for O in "$@"; do
  rm -rf /tmp/tl
  if [ "$O" = "none" ]; then export FPR_OPTS=""; else export FPR_OPTS="$O"; fi
  echo "=== options: $O" >> $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $R/tools/prof_mg.py $N $CSS jacobi 3 >> $OUT 2>/tmp/tl.err || exit 1
  python3 $R/tools/prof_summarize.py timeline /tmp/tl /tmp/tl.txt 32 && cat /tmp/tl.txt >> $OUT
done
