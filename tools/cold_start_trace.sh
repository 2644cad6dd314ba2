#!/bin/bash
# what a cold command-line run spends in the HIP runtime: rocprofv3 --hip-trace --stats of `miniasm <configs[3] text>` (MA_CLEAN_EXIT=1: the fast exit leaves the profiler without its summary)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
P=/tmp/cfg4.paf
[ -f $P ] || miniasm_amd/bin/pafgen -r 2000000 -n 100000000 -s 2 -o $P 2>/dev/null
rm -rf gpurun_out/cold; mkdir -p gpurun_out/cold
(cd /tmp && MA_CLEAN_EXIT=1 MA_PIPE_TIMING=1 timeout 600 rocprofv3 --hip-trace --stats -d /root/repo/gpurun_out/cold -o r --output-format csv -- /root/repo/miniasm_amd/bin/miniasm $P > /dev/null 2> /root/repo/gpurun_out/cold/run.log); echo "rc=$?"
grep -E "T::|Real time" gpurun_out/cold/run.log | head -12
f=$(find gpurun_out/cold -name "*hip_api_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-150
python3 - <<'PY'
import csv, glob
fs = glob.glob('/root/repo/gpurun_out/cold/*hip_api_trace.csv')
if fs:
    rows = list(csv.DictReader(open(fs[0])))
    rows.sort(key=lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp']), reverse=True)
    t0 = min(int(r['Start_Timestamp']) for r in rows)
    print("longest calls (ms since the first call, duration ms, function):")
    for r in rows[:30]:
        print("  %9.2f %8.2f  %s" % ((int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, r['Function']))
PY
find gpurun_out/cold -name "*trace*.csv" -size +4M -delete
