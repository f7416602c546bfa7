#!/bin/bash
# Generic rocprofv3 --pmc runner (one pass per counter set, no tracing) for the diffusion tuning harness.
# usage (on the GPU box): tools/pmc_sets.sh <harness filter> <kernel substr> <out.txt> "<set 1>" "<set 2>" ...
R=$GRAFT_REPO_ROOT; F=$1; K=$2; OUT=$3; shift 3
cd /tmp && export TMPDIR=/tmp
: > $OUT
i=0
for C in "$@"; do
    i=$((i+1)); rm -rf /tmp/pmcs$i
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcs$i -- $R/tools/diffusion_tune 512 3 $F > /tmp/ps$i.log 2>&1
    rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i timed out" >> $OUT; exit $rc; fi
    python3 $R/tools/pmc_kernels.py /tmp/pmcs$i $K 100000 >> $OUT 2>&1 || tail -5 /tmp/ps$i.log >> $OUT
done
cat $OUT
