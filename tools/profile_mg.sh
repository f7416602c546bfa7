#!/bin/bash
# rocprofv3 evidence for the V-cycle block of bench.py (run on the GPU box from the repo root):
#   --kernel-trace --stats of tools/prof_mg.py (MGsolve at 4097^2, l=2 Jacobi; multigrid_bench.jl protocol), then
#   separate --pmc passes for FETCH_SIZE / WRITE_SIZE.  Summaries land in gpurun_out/<tag>_mg_*.txt.
# usage: tools/profile_mg.sh <tag> [n] [coarse_solve_size] [jacobi|cg]
R=$GRAFT_REPO_ROOT; TAG=${1:-prof}; N=${2:-4097}; CSS=${3:-5}; SOLVER=${4:-jacobi}
mkdir -p $R/gpurun_out
# the box all passes of this call run on (kernel statistics and counters come from ONE lease)
{ echo "host $(hostname)"; (rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -2) ; date -u +"utc %Y-%m-%dT%H:%M:%SZ"; } > $R/gpurun_out/${TAG}_mg_box.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm_stats /tmp/pm_fetch /tmp/pm_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm_stats -- python3 $R/tools/prof_mg.py $N $CSS $SOLVER 5 > $R/gpurun_out/${TAG}_mg_run.txt 2> /tmp/pm1.err
rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "stats pass timed out"; exit $rc; fi
python3 $R/tools/prof_summarize.py stats /tmp/pm_stats $R/gpurun_out/${TAG}_mg_kernel_stats.txt
# the five solves the run prints (on the arrays it has placed; the statistics above include the placement's own timed solves and copies)
python3 $R/tools/prof_summarize.py tailstats /tmp/pm_stats $R/gpurun_out/${TAG}_mg_kernel_stats_solves.txt k_cycle_init 5
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pm_fetch -- python3 $R/tools/prof_mg.py $N $CSS $SOLVER 5 > /dev/null 2> /tmp/pm2.err
rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "fetch pass timed out"; exit $rc; fi
python3 $R/tools/prof_summarize.py pmc /tmp/pm_fetch $R/gpurun_out/${TAG}_mg_pmc_fetch.txt
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pm_write -- python3 $R/tools/prof_mg.py $N $CSS $SOLVER 5 > /dev/null 2> /tmp/pm3.err
rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "write pass timed out"; exit $rc; fi
python3 $R/tools/prof_summarize.py pmc /tmp/pm_write $R/gpurun_out/${TAG}_mg_pmc_write.txt
head -14 $R/gpurun_out/${TAG}_mg_kernel_stats_solves.txt
cat $R/gpurun_out/${TAG}_mg_run.txt
