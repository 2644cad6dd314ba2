#!/bin/bash
# Round 4, GPU visit H: the sort's streaming kernels as walking blocks with the next tile in flight (k_radix_hist, k_radix_scatter, k_hit_keys_tiled,
# k_hit_goff) against HEAD, tile / occupancy variants of them, and the ticket dispenser of the coverage kernels (MA_SUB_TICKET) at cfg2 and on 8 shards
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O
t0=$(date +%s)
lap() { echo "## $1: $(( $(date +%s) - t0 )) s since start"; }
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/tests.log 2>&1; echo "parity rc=$?"; tail -2 $O/tests.log
lap parity
tools/variants.sh run head new wpe4 i8 i8w6 new+MA_RS_GRID=1024 new+MA_RS_GRID=4096 i8+MA_RS_GRID=4096 new+MA_SUB_TICKET=1 head 2>&1 | tee $O/variants.txt
lap variants
for v in "MA_SUB_TICKET=0" "MA_SUB_TICKET=1" "MA_SUB_TICKET=0" "MA_SUB_TICKET=1"; do
  env $v timeout 300 python bench.py --reads 200000 --lines 10000000 --seed 1 --no-cpu --no-legs --no-text --steps 20 --warmup 4 > $O/c2.json 2> $O/c2.log; echo "cfg2 $v rc=$?"
  python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4h/c2.json")); ks = {k["name"]: k for k in d["kernels"]}
print("   step %.3f ms | " % d["ms_per_step"] + "  ".join("%s %.3f" % (n, ks[n]["avg_ms"]) for n in ("k_hit_sub<gather>", "k_hit_sub<cut+flt>", "k_hit_cut_contained", "k_hit_keys", "k_radix_scatter", "k_radix_hist", "k_hit_goff") if n in ks))
PY
done
lap cfg2
for v in "MA_SUB_TICKET=0" "MA_SUB_TICKET=1"; do
  env $v timeout 400 python tools/shard_projection.py --ranks 1,8 --steps 4 --per-n-timeout 150 --out $O/shard_projection_${v#*=}.json > $O/projection_${v#*=}.log 2>&1; echo "projection $v"; grep -E "^N=|failed|Error" $O/projection_${v#*=}.log | head
done
lap projection
