#!/bin/bash
# A/B of library builds on one box: alternating runs of `python tools/exp_bal.py 512` (static 255-workgroup fused launch = first line)
# usage: tools/exp_ab_lib.sh <libA.so|default> <libB.so> [rounds]
R=$GRAFT_REPO_ROOT; A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    if [ "$L" = default ]; then unset FPR_LIB_PATH; else export FPR_LIB_PATH=$R/$L; fi
    echo -n "$L: "; python3 $R/tools/exp_bal.py 512 2>/dev/null | grep "round 1  static" | cut -c1-70
  done
done
