mkdir -p gpurun_out/r6
timeout -k 10 900 python -m pytest tests/test_gpu_rccl.py -q -x -k "triples_chained or config4_512" > gpurun_out/r6/t_triples2.log 2>&1; echo rc=$?; tail -15 gpurun_out/r6/t_triples2.log
