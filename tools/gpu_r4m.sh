#!/bin/bash
# Round 4, GPU visit M: the coverage kernels' grid on an 8-rank shard (MA_SUB_BLOCKS: 4096 blocks for 250 k reads leave a wave 15 reads; fewer blocks = longer software pipelines per wave)
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4m; mkdir -p $O
for b in 4096 2048 1024 512; do
  MA_SUB_BLOCKS=$b timeout 300 python tools/shard_projection.py --ranks 8 --steps 4 --per-n-timeout 150 --out $O/proj_$b.json > $O/proj_$b.log 2>&1; echo "MA_SUB_BLOCKS=$b"; grep -E "^N=|failed|Error" $O/proj_$b.log | head -3
  python3 - $b <<'PY'
import json, sys
d = json.load(open("gpurun_out/r4m/proj_%s.json" % sys.argv[1]))
for r in d["runs"]:
    print("   " + "  ".join("%s %.3f" % (k, v[0]) for k, v in r["phase_ms_max_min"].items() if not k.startswith("x:")))
PY
done
