#!/bin/bash
# Evidence for finalprojectrepo.jl_amd/placement.py (run on the GPU box from the repo root; writes gpurun_out/r4_placement_*.txt):
#   1. tools/place_probe: 12 separate 1 GiB allocations, every array alone, then the copy rate between every pair
#   2. tools/diffusion_tune f2place: the fused launch on arrays carved out of fresh slabs (same virtual layout, other physical pages)
#   3. tools/diffusion_tune f2class: the fused launch for every assignment of two copy-classes to its four streams
#   4. bench.py with and without the placement search, alternating, three times
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out; mkdir -p $O
hdr() { echo "# host $(hostname) $(rocm-smi --showuniqueid 2>/dev/null | grep -i 'unique id' | head -1) utc $(date -u +%Y-%m-%dT%H:%M:%SZ)"; }
{ hdr; echo "# tools/place_probe 12"; timeout -k 5 200 $R/tools/place_probe 12; } > $O/r4_placement_probe.txt 2>&1 || exit 1
{ hdr; echo "# tools/diffusion_tune 512 20 f2place"; timeout -k 5 200 $R/tools/diffusion_tune 512 20 f2place | grep f2place; } > $O/r4_placement_f2place.txt 2>&1 || exit 1
{ hdr; echo "# tools/diffusion_tune 512 20 f2class"; timeout -k 5 200 $R/tools/diffusion_tune 512 20 f2class | grep f2class; } > $O/r4_placement_f2class.txt 2>&1 || exit 1
{ hdr; echo "# python3 bench.py [--no-placement] --no-cpu-baseline --no-secondary --no-neighbour-leg --no-power-probe --steps 20 --warmup 5, alternating"
  for i in 1 2 3; do
    for f in "" "--no-placement"; do
      timeout -k 5 200 python3 $R/bench.py $f --no-cpu-baseline --no-secondary --no-neighbour-leg --no-power-probe --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); fp=d['config']['field_placement']
        print('%-15s ms_per_step %.4f  k_diff3_march2 %.4f ms  k_diff3_march %.4f ms  %s' % ('$f' or 'placement', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline_single']['kernel_ms'], {k: fp.get(k) for k in ('pool','trials','trial_ms_best','trial_ms_first','trial_ms_worst','chosen')} if fp.get('selected') else ''))"
    done
  done; } > $O/r4_placement_bench_ab.txt 2>&1
cat $O/r4_placement_bench_ab.txt
