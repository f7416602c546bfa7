#!/bin/bash
# per-kernel counter means for the MGsolve target (tools/prof_mg.py), one --pmc pass per counter group
# usage: tools/pmc_mg.sh <out.txt> "<counters pass 1>" "<counters pass 2>" ...
R=$GRAFT_REPO_ROOT; OUT=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
: > $OUT
for C in "$@"; do
  rm -rf /tmp/pmc_mg
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_mg -- python3 $R/tools/prof_mg.py 4097 5 jacobi 2 > /dev/null 2>/tmp/pmc_mg.err || { tail -5 /tmp/pmc_mg.err; exit 1; }
  python3 $R/tools/prof_summarize.py pmc /tmp/pmc_mg /tmp/pmc_mg.txt && grep -E "^# |^kernel|k_seam_march|k_smooth2_march<true|k_smooth2_march<false, false, true" /tmp/pmc_mg.txt >> $OUT
done
cat $OUT
