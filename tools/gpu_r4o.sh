#!/bin/bash
# Round 4, GPU visit O: what the side streams of the coverage passes cost on small inputs (MA_SUB_FORK_MIN: below it the three size-class launches run back to back)
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4o; mkdir -p $O
for v in "MA_SUB_FORK_MIN=0" "MA_SUB_FORK_MIN=1000000000" "MA_SUB_FORK_MIN=0" "MA_SUB_FORK_MIN=1000000000"; do
  env $v timeout 300 python bench.py --reads 200000 --lines 10000000 --seed 1 --no-cpu --no-legs --no-text --steps 20 --warmup 4 > $O/c2.json 2> $O/c2.log; echo "cfg2 $v rc=$?"
  python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4o/c2.json")); ks = {k["name"]: k for k in d["kernels"]}
print("   step %.3f ms | " % d["ms_per_step"] + "  ".join("%s %.3f" % (n, ks[n]["avg_ms"]) for n in ("k_hit_sub<gather>", "k_hit_sub<cut+flt>", "k_hit_cut_contained", "k_hit_keys", "k_radix_scatter", "k_radix_hist") if n in ks))
PY
done
for v in "MA_SUB_FORK_MIN=0" "MA_SUB_FORK_MIN=1000000000"; do
  env $v timeout 300 python tools/shard_projection.py --ranks 8 --steps 4 --per-n-timeout 100 --out $O/proj_${v#*=}.json > $O/proj_${v#*=}.log 2>&1; echo "projection $v"; grep -E "^N=|failed|Error" $O/proj_${v#*=}.log | head -3
done
