#!/bin/bash
# BASELINE configs[4] through the command line under a list of environment settings, on ONE box, a pause between the runs (the driver clears a finished process's
# memory in the background: a 30 GB allocation right behind a 200 GB process waits seconds for it).  usage: VARIANTS="X=1 MA_THREADS=128 ..." tools/cfg5_ab.sh
cd "$(dirname "$0")/.." || exit 1
P=/tmp/cfg5.paf
[ -f $P ] || miniasm_amd/bin/pafgen -r 5000000 -n 500000000 -s 3 -L uniform -d 0.35 -x 0.03 -o $P 2>/dev/null
lscpu | grep -i "numa\|^CPU(s)\|Thread\|Socket"; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' '; echo
for v in ${VARIANTS:-X=1}; do
  sleep 8
  echo "## [$v]"
  t0=$(date +%s.%N); env $(echo $v | tr '+' ' ') MA_REFSORT_TIMING=1 MA_PIPE_TIMING=1 timeout 900 miniasm_amd/bin/miniasm $P 2> gpurun_out/cfg5_ab.log | md5sum; t1=$(date +%s.%N)
  python3 -c "print('wall %.3f s' % ($t1 - $t0))"
  grep -E "hipMalloc of the text|T::refsort\]  (top|buckets|tasks)|walk: |stable order|hit ranks|push order|T::ties\] [0-9]|Real time" gpurun_out/cfg5_ab.log | head -24
done
echo "(reference md5 of this input: fa9c76984d44526d1a9a9e70132d01da)"
