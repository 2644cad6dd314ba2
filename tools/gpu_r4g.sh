#!/bin/bash
# Round 4, GPU visit G: the larger size classes' streams at the device's highest priority (MA_SUB_PRIO=1, the default) against plain side streams
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4g; mkdir -p $O
for v in "MA_SUB_PRIO=0" "MA_SUB_PRIO=1" "MA_SUB_PRIO=0" "MA_SUB_PRIO=1"; do
  env $v timeout 600 python bench.py --no-cpu --no-legs --no-text --steps 10 --warmup 3 > $O/b.json 2> $O/b.log; echo "$v rc=$?"
  python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4g/b.json")); ks = {k["name"]: k for k in d["kernels"]}
print("   step %.3f ms | " % d["ms_per_step"] + "  ".join("%s %.3f" % (n, ks[n]["avg_ms"]) for n in ("k_hit_sub<gather>", "k_hit_sub<cut+flt>", "k_hit_cut_contained", "k_hit_keys") if n in ks))
PY
done
