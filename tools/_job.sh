mkdir -p gpurun_out/r6
run() { t=$1; shift; timeout -k 10 $t "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT rc=$rc: $*"; exit $rc; fi; return 0; }
run 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r6/bench_d.json 2> gpurun_out/r6/bench_d.err; tail -c 2500 gpurun_out/r6/bench_d.json; cp bench_detail.json gpurun_out/r6/bench_d_detail.json
run 700 python -m pytest tests/test_gpu_rccl.py -q -x -k "two_ranks_rehearsal_on_one_card or watchdog" > gpurun_out/r6/t_rehearse.log 2>&1; tail -25 gpurun_out/r6/t_rehearse.log
