#!/bin/bash
# Round 4, GPU visit Y: SQ counters (instruction mix, wait cycles) of the hit chain's kernels and of the text parser at the round's last kernel commit, and the default bench line again
# (bench.py's pmc lookup learned the k_group_close scope after the final evidence visit: roofline.sort_group.traffic)
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4y; mkdir -p $O
bash tools/gpu_round.sh sq > $O/sq.log 2>&1; grep "sq set" $O/sq.log
for f in k_hit_sub k_radix k_hit_cut_contained k_hit_keys k_paf_parse k_dict_insert; do python tools/pmc_generic.py gpurun_out/sq_1 gpurun_out/sq_2 gpurun_out/sq_3 --filter $f; done > $O/sq_counters.txt 2>&1
wc -l $O/sq_counters.txt
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log; echo "bench rc=$?"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4y/bench_default.json"))
r = d["roofline"]
print("ms_per_step %.3f  value %.4g  gfa_identical %s  latency %s" % (d["ms_per_step"], d["value"], d["gfa_identical"], d.get("latency") and d["latency"]["ms"]))
print("roofline: %s %.3f ms frac %.3f frac_counter %s | sort_group %.3f ms frac %.3f traffic %s | hit_chain %.3f" % (r["kernel"], r["avg_launch_ms"], r["frac"], r.get("frac_counter"), r["sort_group"]["ms_per_step"], r["sort_group"]["frac"], r["sort_group"].get("traffic"), r["hit_chain"]["frac"]))
for n, l in d["legs"].items(): print("leg %-12s %.3f ms/step  identical %s" % (n, l["ms_per_step"], l.get("gfa_identical")))
PY
