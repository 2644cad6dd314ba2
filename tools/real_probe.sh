#!/bin/bash
# the realistic inputs (pafgen -j -b -t) through the command line: digest against tests/golden/big.json, laps of the tie repair
cd "$(dirname "$0")/.." || exit 1
for cfg in "real10 250000 10000000 6 -L uniform" "real50 1000000 50000000 7"; do
  set -- $cfg; name=$1; r=$2; n=$3; s=$4; shift 4
  miniasm_amd/bin/pafgen -r $r -n $n -s $s -j 30 -b 0.1 -t "$@" -d 0.2 -x 0.03 -o /tmp/$name.paf 2>/dev/null
  for k in 1 2; do
    t0=$(date +%s.%N); MA_REFSORT_TIMING=1 MA_PIPE_TIMING=2 timeout 900 miniasm_amd/bin/miniasm /tmp/$name.paf 2> gpurun_out/$name.log | md5sum; t1=$(date +%s.%N)
    python3 -c "print('$name run $k: %.3f s wall' % ($t1 - $t0))"
  done
  grep -E "T::ties|T::refsort|T::head\] (sort|sg_gen)|T::ingest|walk:|Real time|T::pipeline" gpurun_out/$name.log | head -40
  python3 -c "
import json; g=json.load(open('tests/golden/big.json'))['inputs']['$name']; print('recorded reference md5', g['gfa_md5'], 'wall', g['reference_wall_s'])"
  rm -f /tmp/$name.paf
done
