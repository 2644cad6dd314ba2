#!/bin/bash
# Round 4, GPU visit P: per-kernel times of the hit chain at a tenth of the scale (BASELINE configs[1], 20 M hits): which launches carry the part of a pass that does not shrink with the input
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4p; mkdir -p $O
rm -rf $O/prof; mkdir -p $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof -o r --output-format csv -- python /root/repo/bench.py --reads 200000 --lines 10000000 --seed 1 --steps 10 --warmup 2 --no-cpu --no-legs --no-text --prof-steps 0 > /root/repo/$O/prof/bench.json 2> /root/repo/$O/prof/bench.log); echo "rocprof cfg2 rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_cfg2.csv
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r4p/rocprofv3_kernel_stats_cfg2.csv")))
for r in rows[:26]:
    print("%-60s calls %4s avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
find $O/prof -name "*.csv" ! -name "*stats*" -delete 2>/dev/null
