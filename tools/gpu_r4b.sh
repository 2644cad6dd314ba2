#!/bin/bash
# Round 4, GPU visit B: v_med3 sorting network, lazy CSR prefetch in the reduction, pipelined arc group sort, bubble witnesses on the device.
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
t0=$(date +%s)
lap() { echo "## $1: $(( $(date +%s) - t0 )) s since start"; }
timeout 900 python -m pytest tests/test_gpu_graph_api.py -m gpu -q -x --tb=short -p no:cacheprovider -k "revives or sequential_bubble or handmade or circular" > $O/tests_new.log 2>&1; echo "new tests rc=$?"
grep -vE "^\[M::|^\[pafgen" $O/tests_new.log | tail -4
MINIASM_AMD_LIB=$PWD/build/variants/med3/libminiasm_amd.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/tests_med3.log 2>&1; echo "parity on the med3 library rc=$?"
grep -vE "^\[M::|^\[pafgen" $O/tests_med3.log | tail -3
lap tests
for v in base med3 trpre4 trpre1 base; do
  lib=$PWD/build/variants/$v/libminiasm_amd.so
  MINIASM_AMD_LIB=$lib timeout 900 python bench.py --no-cpu --no-text --legs graph_heavy --steps 10 --warmup 3 > $O/v_$v.json 2> $O/v_$v.log; echo "variant $v rc=$?"
  python3 - $O/v_$v.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); g = d["legs"]["graph_heavy"]; r = g["reduce_group"]
    ks = {k["name"]: k for k in d["kernels"]}
    print("   cfg4 step %.3f ms | " % d["ms_per_step"] + "  ".join("%s %.3f" % (n, ks[n]["avg_ms"]) for n in ("k_hit_sub<gather>", "k_hit_sub<cut+flt>", "k_hit_cut_contained", "k_hit_keys", "k_radix_scatter") if n in ks))
    print("   graph-heavy step %.2f ms  reduce_group %.2f ms frac %.3f | " % (g["ms_per_step"], r["ms_per_step"], r["frac"]) + "  ".join("%s %.3f" % (k["name"], k["avg_ms"]) for k in r["kernels"]))
    gk = {k["name"]: k for k in g["kernels"]}
    print("   graph-heavy: " + "  ".join("%s %.3f" % (n, gk[n]["avg_ms"]) for n in ("k_hit_sub<gather>", "k_hit_sub<cut+flt>", "k_sg_emit", "k_sg_arcs") if n in gk))
except Exception as e:
    print("   failed:", e)
PY
done
lap variants
