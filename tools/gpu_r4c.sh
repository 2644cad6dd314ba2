#!/bin/bash
# Round 4, GPU visit C: pipelined reduction + arc sort (rows a read ahead, bounds from the emit pass), arcs_clean; shard projection at HEAD
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
t0=$(date +%s)
lap() { echo "## $1: $(( $(date +%s) - t0 )) s since start"; }
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph_api.py tests/test_gpu_graph_fuzz.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"
grep -vE "^\[M::|^\[pafgen" $O/tests.log | tail -4
lap tests
for k in 1 2; do
timeout 900 python bench.py --no-cpu --no-text --legs graph_heavy --steps 10 --warmup 3 > $O/head_$k.json 2> $O/head_$k.log; echo "bench rc=$?"
python3 - $O/head_$k.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); g = d["legs"]["graph_heavy"]; r = g["reduce_group"]
    ks = {k["name"]: k for k in d["kernels"]}
    print("   cfg4 step %.3f ms | " % d["ms_per_step"] + "  ".join("%s %.3f" % (n, ks[n]["avg_ms"]) for n in ("k_hit_sub<gather>", "k_hit_sub<cut+flt>", "k_hit_cut_contained", "k_hit_keys", "k_radix_scatter") if n in ks))
    print("   graph-heavy step %.2f ms  reduce_group %.2f ms frac %.3f | " % (g["ms_per_step"], r["ms_per_step"], r["frac"]) + "  ".join("%s %.3f" % (k["name"], k["avg_ms"]) for k in r["kernels"]))
    print("   graph-heavy: " + "  ".join("%s x%g %.3f" % (k["name"], k["launches_per_step"], k["avg_ms"]) for k in g["kernels"][:12]))
except Exception as e:
    print("   failed:", e)
PY
done
lap bench
timeout 1200 python tools/shard_projection.py --ranks 1,2,4,8 --steps 3 --out $O/shard_projection.json 2>&1 | grep -E "^N=" 
lap projection
