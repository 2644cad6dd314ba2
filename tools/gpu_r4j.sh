#!/bin/bash
# Round 4, GPU visit J: does a gather whose fetches stay inside a window of 32 .. 512 MB run from the Infinity Cache?  (csrc/diag.hip, 6.4 GB of records = BASELINE configs[4])
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4j; mkdir -p $O
timeout 300 python tools/pmc_calibrate.py run --sizes 6400 --reps 4 --patterns gather32+cols,gather32+cols_win32MB,gather32+cols_win64MB,gather32+cols_win128MB,gather32+cols_win256MB,gather32+cols_win512MB,copy16,read8_of32+write8,scatter_runs32 --out $O/diag_windows.json 2> $O/run.log | tee $O/diag_windows.txt; tail -3 $O/run.log
for ctr in FETCH_SIZE; do
  rm -rf $O/pmc_$ctr; mkdir -p $O/pmc_$ctr
  (cd /tmp && timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d /root/repo/$O/pmc_$ctr -o r --output-format csv -- python /root/repo/tools/pmc_calibrate.py run --sizes 6400 --reps 1 --patterns gather32+cols,gather32+cols_win32MB,gather32+cols_win64MB,gather32+cols_win128MB,gather32+cols_win256MB,gather32+cols_win512MB > /root/repo/$O/pmc_$ctr/run.txt 2> /root/repo/$O/pmc_$ctr/run.log); echo "pmc $ctr rc=$?"
  python3 - <<'PY'
import csv, glob
for fn in glob.glob("gpurun_out/r4j/pmc_FETCH_SIZE/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if row.get("Counter_Name") == "FETCH_SIZE" and "gather32" in row.get("Kernel_Name", ""):
            print("   FETCH_SIZE %.2f GB (x2 = %.2f GB)  dispatch %s" % (float(row["Counter_Value"]) * 1024 / 1e9, float(row["Counter_Value"]) * 2048 / 1e9, row.get("Dispatch_Id")))
PY
done
find $O -name "*.csv" -size +1M -delete 2>/dev/null
