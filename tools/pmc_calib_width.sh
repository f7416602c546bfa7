#!/bin/bash
# What FETCH_SIZE / WRITE_SIZE report for a KNOWN byte count by access width (MI355X_MICROARCH: FETCH_SIZE reads half the bytes of a
# 16-byte-per-lane streaming read; "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern"):
# tools/bw_probe (out[i] = a[i] + b[i] over 4097^2 doubles: 268.6 MB read, 134.3 MB written per launch) with 8-byte lanes (k_v8), 16-byte lanes
# (k_v16) and the row march of the multigrid passes (k_rows8), one --pmc pass per counter.
#   gpurun -- 'bash tools/pmc_calib_width.sh gpurun_out/r5/pmc_calib_width.txt'
R=$GRAFT_REPO_ROOT; OUT=$R/$1
cd /tmp && export TMPDIR=/tmp
: > $OUT
echo "# bytes per launch: read 268566544 (2 x 4097^2 x 8), written 134283272; counters in KiB (rocprofv3 --pmc, mean per dispatch)" >> $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_cal
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_cal -- $R/tools/bw_probe > /dev/null 2>/tmp/pmc_cal.err || { tail -5 /tmp/pmc_cal.err; exit 1; }
  python3 $R/tools/prof_summarize.py pmc /tmp/pmc_cal /tmp/pmc_cal.txt && grep -E "^kernel|k_v8|k_v16|k_rows" /tmp/pmc_cal.txt >> $OUT
done
python3 - $OUT <<'PY' >> $OUT
import sys
rows = [l.split() for l in open(sys.argv[1]) if l.startswith("k_") or l.startswith("void k_")]
rd, wr = 268566544 / 1024.0, 134283272 / 1024.0
print("# kernel counter mean_KiB  bytes_counted / bytes_moved")
for r in rows:
    name = " ".join(r[:-5]); cnt = r[-5]; mean = float(r[-3])
    print("# %-40s %-10s %12.0f  %.3f" % (name[:40], cnt, mean, mean / (rd if cnt == "FETCH_SIZE" else wr)))
PY
cat $OUT
