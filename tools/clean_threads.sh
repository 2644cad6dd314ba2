#!/bin/bash
# cleaner thread count at cfg5 scale (GPU box)
cd "$(dirname "$0")/.." || exit 1
P=/tmp/big.paf
miniasm_amd/bin/pafgen -r 5000000 -n 500000000 -s 3 -L uniform -d 0.35 -x 0.03 -o $P 2>/dev/null
for th in 8 16 32 64; do
  MA_THREADS=$th MA_PIPE_TIMING=1 miniasm_amd/bin/miniasm $P 2> /tmp/ct.log > /tmp/ct.gfa
  echo "== MA_THREADS=$th: $(grep -E 'T::cleaners' /tmp/ct.log) | $(grep -E 'Real time' /tmp/ct.log) | md5 $(md5sum < /tmp/ct.gfa | cut -c1-8)"
  grep -E "T::ingest_mt|T::ingest_gpu\] parse|T::tail" /tmp/ct.log | cut -c1-200
done
