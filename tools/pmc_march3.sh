#!/bin/bash
# SQ counters of k_diff3_march3 beside k_diff3_march2 (tools/diffusion_tune3, 512^3): separate rocprofv3 --pmc passes, no tracing.
# usage (GPU box, repo root): tools/pmc_march3.sh <harness binary> <out.txt>
R=$GRAFT_REPO_ROOT; B=${1:-tools/diffusion_tune3}; OUT=${2:-gpurun_out/pmc_march3.txt}
cd /tmp && export TMPDIR=/tmp
{ echo "# host $(hostname) $(rocm-smi --showuniqueid 2>/dev/null | grep -i 'unique id' | head -1) utc $(date -u +%Y-%m-%dT%H:%M:%SZ)"; echo "# $B 512 512 512 6"; } > $R/$OUT
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); rm -rf /tmp/pm3_$i
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pm3_$i -- $R/$B 512 512 512 6 > /tmp/pm3_$i.log 2>&1
    rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i timed out" >> $R/$OUT; exit $rc; fi
    echo "## pass $i: $C" >> $R/$OUT
    python3 $R/tools/pmc_kernels.py /tmp/pm3_$i k_diff3_march 100000 >> $R/$OUT 2>&1 || tail -5 /tmp/pm3_$i.log >> $R/$OUT
done
cat $R/$OUT
