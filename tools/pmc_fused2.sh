#!/bin/bash
# SQ / HBM counters of the fused two-step diffusion kernel (separate rocprofv3 --pmc passes, no tracing).
# usage (on the GPU box): tools/pmc_fused2.sh <filter> <out.txt>
R=$GRAFT_REPO_ROOT; F=${1:-f2-xcd0-zc0-n0}; OUT=${2:-$R/gpurun_out/f2_pmc.txt}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES" \
         "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc$i -- $R/tools/diffusion_tune 512 3 $F > /tmp/p$i.log 2>&1
    rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i timed out"; exit $rc; fi
done
(for d in /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 /tmp/pmc4; do python3 $R/tools/pmc_kernels.py $d march2 100000; done) > $OUT 2>&1
cat $OUT
