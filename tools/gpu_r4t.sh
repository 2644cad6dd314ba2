#!/bin/bash
# Round 4, GPU visit T (the round's last 2 GPU-minutes): k_paf_parse from aligned LDS words (MA_PARSE_WORDS=1) against the byte-wise form, CLI on the configs[3] text
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4t; mkdir -p $O
miniasm_amd/bin/pafgen -r 2000000 -n 100000000 -s 2 -o /tmp/t.paf 2>/dev/null
for m in 0 1 0 1; do
  MA_PARSE_WORDS=$m MA_PIPE_TIMING=1 timeout 30 miniasm_amd/bin/miniasm /tmp/t.paf 2> $O/cli_$m.log | md5sum | cut -c1-12 > $O/cli_$m.md5
  echo "words=$m md5 $(cat $O/cli_$m.md5) $(grep -E 'T::ingest_gpu\] parse' $O/cli_$m.log | head -1)"
done
for m in 0 1; do
  rm -rf $O/prof_$m; mkdir -p $O/prof_$m
  (cd /tmp && MA_PARSE_WORDS=$m timeout 40 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_$m -o r --output-format csv -- /root/repo/miniasm_amd/bin/miniasm /tmp/t.paf > /dev/null 2> /root/repo/$O/prof_$m/log)
  f=$(find $O/prof_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "k_paf_parse|k_dict_insert|k_paf_nl|k_paf_ids|k_paf_emit" $f | cut -d, -f1-4 | cut -c1-120
  find $O/prof_$m -name "*.csv" ! -name "*stats*" -delete 2>/dev/null
done
