#!/bin/bash
# BASELINE configs[4] (500 M overlaps, 5 M reads, noisy) through the command line with every lap stamp on: where its seconds go (MA_PIPE_TIMING=2 adds a device wait per lap)
cd "$(dirname "$0")/.." || exit 1
P=/tmp/cfg5.paf
[ -f $P ] || miniasm_amd/bin/pafgen -r 5000000 -n 500000000 -s 3 -L uniform -d 0.35 -x 0.03 -o $P 2>/dev/null
ls -l $P
for v in "$@"; do
  echo "## [${v:-default}]"
  t0=$(date +%s.%N); env ${v:-X=1} MA_REFSORT_TIMING=1 MA_PIPE_TIMING=2 timeout 900 miniasm_amd/bin/miniasm $P 2> gpurun_out/cfg5_probe.log | md5sum; t1=$(date +%s.%N)
  python3 -c "print('wall %.3f s' % ($t1 - $t0))"
  grep -E "^\[T::" gpurun_out/cfg5_probe.log | head -80
done
echo "(reference md5 of this input: fa9c76984d44526d1a9a9e70132d01da)"
