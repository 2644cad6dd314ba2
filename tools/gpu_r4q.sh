#!/bin/bash
# Round 4, GPU visit Q: the larger size classes of the coverage passes end after the first tier (visit P: 0.65 against 0.44 ms at configs[1], 6.2 against 4.9 at configs[3]):
# their launches first (MA_SUB_ORDER=1) and / or on a smaller grid (MA_SUB_BLOCKS_HI) so that they run beside the first tier from its start
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
run() { # $1 = bench args tag, rest = env
  tag=$1; shift
  case $tag in cfg4) A="--steps 8 --warmup 2";; cfg2) A="--reads 200000 --lines 10000000 --seed 1 --steps 20 --warmup 4";; esac
  env "$@" timeout 300 python bench.py $A --no-cpu --no-legs --no-text > $O/b.json 2> $O/b.log; rc=$?
  python3 - "$tag $*" <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/r4q/b.json")); ks = {k["name"]: k for k in d["kernels"]}
    print("%-44s step %.3f ms | " % (sys.argv[1], d["ms_per_step"]) + "  ".join("%s %.3f" % (n, ks[n]["avg_ms"]) for n in ("k_hit_sub<gather>", "k_hit_sub<cut+flt>", "k_hit_cut_contained") if n in ks))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
run cfg4 MA_X=0
run cfg4 MA_SUB_ORDER=1
run cfg4 MA_SUB_BLOCKS_HI=1024
run cfg4 MA_SUB_BLOCKS_HI=512
run cfg4 MA_SUB_BLOCKS_HI=256
run cfg4 MA_SUB_ORDER=1 MA_SUB_BLOCKS_HI=1024
run cfg4 MA_SUB_ORDER=1 MA_SUB_BLOCKS_HI=512
run cfg4 MA_SUB_ORDER=1 MA_SUB_BLOCKS_HI=256
run cfg4 MA_X=0
run cfg2 MA_X=0
run cfg2 MA_SUB_ORDER=1 MA_SUB_BLOCKS_HI=512
run cfg2 MA_SUB_ORDER=1 MA_SUB_BLOCKS_HI=256
run cfg2 MA_SUB_BLOCKS_HI=512
