#!/bin/bash
# Control runs for bench.py's norm check at N > 1 (tests/golden/scale_norms.json): every process grid the scaling bench can
# run, as ONE rank on one MI355X (the largest, 2,2,2 at n = 512, is a 1022^3 grid: 5 arrays = 43 GB).
#   gpurun -- 'bash tools/make_scale_norms.sh'     ->  gpurun_out/scale_norms.json  (copy to tests/golden/)
set -e
mkdir -p gpurun_out
cp tests/golden/scale_norms.json gpurun_out/scale_norms.json 2>/dev/null || true
python3 bench.py --gpus 1 --n 128 --golden-norms gpurun_out/scale_norms.json --golden-iters 160 --golden-dims "1,1,2;1,1,4;2,1,1;2,2,1;2,1,2;1,2,2;2,2,2"
python3 bench.py --gpus 1 --n 512 --golden-norms gpurun_out/scale_norms.json --golden-iters 320
# N = 1 (round 5): the single-rank problem itself, far enough for the default flags (1536 + 20 + 200 iterations); its values at
# 8 / 64 / 1562 / 1756 iterations are pinned on the CPU oracle by tools/make_n1_norm_pins.py
python3 bench.py --gpus 1 --n 128 --golden-norms gpurun_out/scale_norms.json --golden-iters 1800 --golden-dims "1,1,1"
python3 bench.py --gpus 1 --n 512 --golden-norms gpurun_out/scale_norms.json --golden-iters 1800 --golden-dims "1,1,1"
