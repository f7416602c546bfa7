mkdir -p gpurun_out/r6
timeout -k 10 300 python3 tools/probe_slab_windows.py > gpurun_out/r6/slab_windows.txt 2>&1; cat gpurun_out/r6/slab_windows.txt | tail -70
