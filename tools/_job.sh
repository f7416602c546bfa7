mkdir -p gpurun_out/r6
timeout -k 10 1120 python -m pytest tests/test_gpu_rccl.py -q -x -m gpu > gpurun_out/r6/gputests_rccl.log 2>&1; echo rc=$?; tail -12 gpurun_out/r6/gputests_rccl.log
