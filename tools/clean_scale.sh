#!/bin/bash
# sequential vs speculative cleaners at scale (GPU box): usage tools/clean_scale.sh reads lines
cd "$(dirname "$0")/.." || exit 1
P=/tmp/cs.paf
miniasm_amd/bin/pafgen -r $1 -n $2 -s 3 -L uniform -d 0.35 -x 0.03 -o $P 2>/dev/null
for mode in 1000000000 0; do
  for th in 16 32; do
    [ $mode = 1000000000 ] && [ $th = 32 ] && continue
    MA_CLEAN_PAR_MIN=$mode MA_THREADS=$th MA_PIPE_TIMING=1 miniasm_amd/bin/miniasm $P 2> /tmp/cs.log > /tmp/cs_$mode.gfa
    echo "== MA_CLEAN_PAR_MIN=$mode threads=$th: $(grep -E 'T::tail' /tmp/cs.log)"
    grep -E "T::pop_bubble" /tmp/cs.log | head -12
  done
done
md5sum /tmp/cs_0.gfa /tmp/cs_1000000000.gfa
