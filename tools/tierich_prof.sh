#!/bin/bash
# the tie-rich 5 M-overlap input (bench.py legs.tie_rich) through the command line: laps (MA_PIPE_TIMING=2) and rocprofv3 kernel stats
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
miniasm_amd/bin/pafgen -r 250000 -n 5000000 -s 5 -q 16 -L uniform -d 0.3 -x 0.03 -o /tmp/twrich.paf 2>/dev/null
MA_PIPE_TIMING=2 MA_REFSORT_TIMING=1 timeout 600 miniasm_amd/bin/miniasm /tmp/twrich.paf 2> gpurun_out/tierich.log | md5sum
grep -E "T::ties|T::refsort|T::head|T::tail|T::clean|T::pipeline|Real time" gpurun_out/tierich.log | head -60
rm -rf gpurun_out/tierich_prof; mkdir -p gpurun_out/tierich_prof
(cd /tmp && MA_CLEAN_EXIT=1 timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/tierich_prof -o r --output-format csv -- /root/repo/miniasm_amd/bin/miniasm /tmp/twrich.paf > /dev/null 2> /root/repo/gpurun_out/tierich_prof/run.log); echo "rc=$?"
f=$(find gpurun_out/tierich_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 tools/kstats.py "$f" 30
find gpurun_out/tierich_prof -name "*trace*.csv" -size +4M -delete
