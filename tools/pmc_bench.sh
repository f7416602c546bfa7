#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (+ optional SQ set) of bench.py's diffusion kernels: separate rocprofv3 --pmc passes, no tracing.
# usage (GPU box, repo root): tools/pmc_bench.sh <out.txt> [bench.py args...]
R=$GRAFT_REPO_ROOT; OUT=$R/$1; shift
ARGS="--no-secondary --no-cpu-baseline --no-single-leg --steps 100 --warmup 20 $*"
cd /tmp && export TMPDIR=/tmp
: > $OUT
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
    i=$((i+1)); rm -rf /tmp/pb$i
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pb$i -- python3 $R/bench.py $ARGS > /dev/null 2> /tmp/pb$i.err
    rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i timed out" >> $OUT; exit $rc; fi
    python3 $R/tools/pmc_kernels.py /tmp/pb$i diff3 1000 >> $OUT 2>&1
done
cat $OUT
