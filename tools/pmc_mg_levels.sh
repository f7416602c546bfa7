#!/bin/bash
# what do the passes over the 2049^2 and 1025^2 levels of a 4097^2 V-cycle wait for?  Per level (= per grid size of
# k_smooth2_march_v2): HBM-side traffic (FETCH_SIZE x 2 + WRITE_SIZE: is the level really served from the Infinity Cache?), instruction counts and
# wait shares, from separate rocprofv3 --pmc passes of tools/prof_mg.py; beside it the kernel trace for the durations per level.
# usage (GPU box, repo root): tools/pmc_mg_levels.sh <out.txt>
R=$GRAFT_REPO_ROOT; OUT=${1:-gpurun_out/mg_levels.txt}
cd /tmp && export TMPDIR=/tmp
{ echo "# host $(hostname) $(rocm-smi --showuniqueid 2>/dev/null | grep -i 'unique id' | head -1) utc $(date -u +%Y-%m-%dT%H:%M:%SZ)"; echo "# python3 tools/prof_mg.py 4097 5 jacobi 5"; } > $R/$OUT
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    i=$((i+1)); rm -rf /tmp/pml_$i
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pml_$i -- python3 $R/tools/prof_mg.py 4097 5 jacobi 5 > /tmp/pml_$i.log 2>&1
    rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pass $i timed out" >> $R/$OUT; exit $rc; fi
    echo "## pass $i: $C" >> $R/$OUT
    python3 $R/tools/pmc_by_grid.py /tmp/pml_$i k_smooth2_march_v2 k_seam_march k_mid_down k_mid_up k_mg_small >> $R/$OUT 2>&1 || tail -5 /tmp/pml_$i.log >> $R/$OUT
done
rm -rf /tmp/pml_t
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pml_t -- python3 $R/tools/prof_mg.py 4097 5 jacobi 5 > /tmp/pml_t.log 2>&1
echo "## kernel trace: duration per (kernel, grid size), mean of the last 30 dispatches of each" >> $R/$OUT
python3 - >> $R/$OUT <<PY
import csv, glob, collections
f = glob.glob("/tmp/pml_t/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    if not any(s in k for s in ("k_smooth2_march_v2", "k_seam_march", "k_mid_down", "k_mid_up", "k_mg_small", "k_cycle")): continue
    g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    d[(k, g)].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for (k, g), v in sorted(d.items(), key=lambda q: (q[0][0], -q[0][1])):
    v = v[-30:]
    print("%s  grid %d: %.2f us (n=%d)" % (k, g, sum(e - s for s, e in v) / len(v) / 1e3, len(v)))
PY
cat $R/$OUT
