#!/bin/bash
# rocprofv3 evidence for bench.py's dominant kernel (run on the GPU box from the repo root):
#   1. --kernel-trace --stats      -> per-kernel average duration
#   2. --pmc FETCH_SIZE            -> L2 -> fabric read traffic   (separate passes, no tracing alongside counters)
#   3. --pmc WRITE_SIZE            -> write traffic
# Raw traces stay in /tmp; summaries land in gpurun_out/<tag>_*.txt (copy them to profiles/).
# usage: tools/profile_bench.sh <tag> [bench.py args...]
R=$GRAFT_REPO_ROOT; TAG=${1:-prof}; shift
ARGS="--no-secondary --no-cpu-baseline --no-neighbour-leg --steps 200 --warmup 20 $*"
mkdir -p $R/gpurun_out
# the box all passes of this call run on (kernel statistics and counters come from ONE lease)
{ echo "host $(hostname)"; (rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -2) ; date -u +"utc %Y-%m-%dT%H:%M:%SZ"; } > $R/gpurun_out/${TAG}_box.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb_stats /tmp/pb_fetch /tmp/pb_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> /tmp/pb1.err
rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "stats pass timed out"; exit $rc; fi
python3 $R/tools/prof_summarize.py stats /tmp/pb_stats $R/gpurun_out/${TAG}_kernel_stats.txt
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pb_fetch -- python3 $R/bench.py $ARGS > /dev/null 2> /tmp/pb2.err
rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "fetch pass timed out"; exit $rc; fi
python3 $R/tools/prof_summarize.py pmc /tmp/pb_fetch $R/gpurun_out/${TAG}_pmc_fetch.txt
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pb_write -- python3 $R/bench.py $ARGS > /dev/null 2> /tmp/pb3.err
rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "write pass timed out"; exit $rc; fi
python3 $R/tools/prof_summarize.py pmc /tmp/pb_write $R/gpurun_out/${TAG}_pmc_write.txt
head -12 $R/gpurun_out/${TAG}_kernel_stats.txt
grep -E "diff3" $R/gpurun_out/${TAG}_pmc_fetch.txt $R/gpurun_out/${TAG}_pmc_write.txt
cut -c1-300 $R/gpurun_out/${TAG}_bench_under_rocprof.json
