#!/bin/bash
# the graph-heavy input (200 M arcs): step time with the host's big blocks on huge pages (MA_HOST_THP=1) or not, alternating on ONE box
cd "$(dirname "$0")/.." || exit 1
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null
GH="--reads 2000000 --lines 100000000 --seed 4 --model fixed --no-cpu --no-legs --no-text --steps 12 --warmup 3 --prof-steps 0"
for v in 0 1 0 1 0 1; do
  MA_HOST_THP=$v timeout 900 python bench.py $GH > gpurun_out/gh_ab.json 2> gpurun_out/gh_ab.log; echo "[MA_HOST_THP=$v] rc=$?"
  python3 -c "
import json; d=json.load(open('gpurun_out/gh_ab.json')); p=d['phases']; print('  step %.3f ms  head %.2f tail %.2f ' % (d['ms_per_step'], p['head_wall_ms'], p['tail_wall_ms']), p['tail_last_pass_ms'])"
done
