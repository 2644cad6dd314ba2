#!/bin/bash
# the graph-heavy input (200 M arcs, 87 MB of GFA): step time with the tail of a batch on one worker thread (round 5: device part, then text) or in two stages on two
# threads (device part of batch k+1 beside the text of batch k), alternating on ONE box.  VARIANTS="..." overrides the list (bench.py flags or VAR=1 settings per entry)
cd "$(dirname "$0")/.." || exit 1
GH="--reads 2000000 --lines 100000000 --seed 4 --model fixed --no-cpu --no-legs --no-text --steps 12 --warmup 3 --prof-steps 0"
IFS='|' read -ra VARS <<< "${VARIANTS:---tail-one-stage||--tail-one-stage|}"
for v in "${VARS[@]}"; do
  envs=""; flags=""
  for w in $v; do case $w in *=*) envs="$envs $w";; *) flags="$flags $w";; esac; done
  env $envs timeout 900 python bench.py $GH $flags > gpurun_out/gh_ab.json 2> gpurun_out/gh_ab.log; echo "[${v:-default}] rc=$?"
  python3 -c "
import json; d=json.load(open('gpurun_out/gh_ab.json')); p=d['phases']; print('  step %.3f ms  head %.2f tail %.2f (device stage %s, text stage %s) ' % (d['ms_per_step'], p['head_wall_ms'], p['tail_wall_ms'], p.get('tail_device_stage_ms'), p.get('tail_text_stage_ms')), p['tail_last_pass_ms'])"
done
