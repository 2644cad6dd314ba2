#!/bin/bash
# the 50 M-line realistic input (pafgen -j 30 -b 0.1 -t) through the command line under rocprofv3 --kernel-trace --stats: what the device does on an input where the record sort,
# the tie census and both walks run (MA_CLEAN_EXIT=1: the fast exit leaves the profiler without its summary) -> gpurun_out/real_prof/
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
miniasm_amd/bin/pafgen -r 1000000 -n 50000000 -s 7 -j 30 -b 0.1 -t -d 0.2 -x 0.03 -o /tmp/real50.paf 2>/dev/null
rm -rf gpurun_out/real_prof; mkdir -p gpurun_out/real_prof
(cd /tmp && MA_CLEAN_EXIT=1 MA_PIPE_TIMING=1 timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/real_prof -o r --output-format csv -- /root/repo/miniasm_amd/bin/miniasm /tmp/real50.paf 2> /root/repo/gpurun_out/real_prof/run.log | md5sum); echo "rc=$?"
grep -E "T::ties\]|Real time|T::pipeline" gpurun_out/real_prof/run.log | head -5
f=$(find gpurun_out/real_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 tools/kstats.py "$f" 25
find gpurun_out/real_prof -name "*trace*.csv" -size +4M -delete
rm -f /tmp/real50.paf
