#!/bin/bash
# Memory-side counters of bench.py's two diffusion kernels side by side -- k_diff3_march2 (two iterations per launch) and
# k_diff3_march (one) run in the same process on the same box: where each waits (L1 / L2 / fabric), request latencies, queue levels.
# Separate rocprofv3 --pmc passes, no tracing.  usage (GPU box, repo root): tools/pmc_memside.sh <out.txt>
R=$GRAFT_REPO_ROOT; OUT=$R/$1; shift
ARGS="--no-secondary --no-cpu-baseline --no-neighbour-leg --steps 100 --warmup 20 $*"
cd /tmp && export TMPDIR=/tmp
{ echo "# host $(hostname) $(rocm-smi --showuniqueid 2>/dev/null | grep -i 'unique id' | head -1) utc $(date -u +%Y-%m-%dT%H:%M:%SZ)"; echo "# python3 bench.py $ARGS"; } > $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum" \
         "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_HIT_sum TCC_MISS_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TD_TC_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1)); rm -rf /tmp/pm$i
    echo "## pass $i: $C" >> $OUT
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pm$i -- python3 $R/bench.py $ARGS > /dev/null 2> /tmp/pm$i.err
    rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i timed out" >> $OUT; exit $rc; fi
    python3 $R/tools/pmc_kernels.py /tmp/pm$i diff3_march 1000 >> $OUT 2>&1 || tail -3 /tmp/pm$i.err >> $OUT
done
cat $OUT
