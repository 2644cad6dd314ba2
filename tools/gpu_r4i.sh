#!/bin/bash
# Round 4, GPU visit I: what the box sustains for the hot path's access patterns (csrc/diag.hip), and FETCH_SIZE / WRITE_SIZE calibrated against their known byte counts
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O
t0=$(date +%s)
lap() { echo "## $1: $(( $(date +%s) - t0 )) s since start"; }
timeout 300 python tools/pmc_calibrate.py run --sizes 800,3200 --reps 5 --out $O/diag_run.json 2> $O/run.log | tee $O/diag_run.txt; tail -3 $O/run.log
lap run
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$ctr; mkdir -p $O/pmc_$ctr
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d /root/repo/$O/pmc_$ctr -o r --output-format csv -- python /root/repo/tools/pmc_calibrate.py run --sizes 800,3200 --reps 2 > /root/repo/$O/pmc_$ctr/run.txt 2> /root/repo/$O/pmc_$ctr/run.log); echo "pmc $ctr rc=$?"
done
python tools/pmc_calibrate.py pmc $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/diag_run.json --out $O/pmc_calibration.json | tee $O/pmc_calibration.txt
find $O -name "*.csv" -size +1M -delete 2>/dev/null
lap pmc
