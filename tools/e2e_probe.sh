#!/bin/bash
# end-to-end wall time of the CLI on the cfg2 text (GPU box); extra args = env assignments
cd "$(dirname "$0")/.." || exit 1
P=/tmp/cfg2.paf
miniasm_amd/bin/pafgen -r 200000 -n 10000000 -s 1 -o $P 2>/dev/null
ls -l $P | awk '{print "paf bytes", $5}'
cat $P > /dev/null
for i in 1 2 3; do
  t0=$(date +%s.%N)
  env MA_PIPE_TIMING=1 "$@" miniasm_amd/bin/miniasm $P 2> /tmp/e2e.log > /tmp/e2e.gfa
  t1=$(date +%s.%N)
  grep -E "T::|Real time" /tmp/e2e.log
  echo "wall $(echo "$t1 - $t0" | bc -l 2>/dev/null || python3 -c "print($t1-$t0)") s"
done
md5sum /tmp/e2e.gfa
