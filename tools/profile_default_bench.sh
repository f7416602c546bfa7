#!/bin/bash
# rocprofv3 --kernel-trace --stats of the driver's own command (python3 bench.py --gpus 1 --steps 20 --warmup 5): per-kernel averages of everything
# the line reports (diffusion legs, V-cycle blocks, Navier-Stokes).  usage: tools/profile_default_bench.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=${1:-default}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pd_stats
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd_stats -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> /tmp/pd.err
rc=$?; if [ $rc -ne 0 ]; then tail -5 /tmp/pd.err; exit $rc; fi
python3 $R/tools/prof_summarize.py stats /tmp/pd_stats $R/gpurun_out/${TAG}_kernel_stats.txt
head -40 $R/gpurun_out/${TAG}_kernel_stats.txt
