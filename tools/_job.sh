mkdir -p gpurun_out/r6
run() { t=$1; shift; timeout -k 10 $t "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT rc=$rc: $*"; exit $rc; fi; return 0; }
for tol in 2e-4 3e-5; do run 200 python tests/thread_ranks_worker.py 1,1,4 solver 128 0 $tol 2>&1 | tail -6; done > gpurun_out/r6/thr_solver.txt 2>&1; cat gpurun_out/r6/thr_solver.txt
run 900 python -m pytest tests -q -x -m gpu --ignore=tests/test_gpu_rccl.py > gpurun_out/r6/gputests_rest.log 2>&1; tail -8 gpurun_out/r6/gputests_rest.log
