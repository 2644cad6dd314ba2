#!/bin/bash
# Kernel-variant experiments: build libminiasm_amd.so with other compile-time parameters next to the product library
# (build/variants/<name>/libminiasm_amd.so, found by MINIASM_AMD_LIB), then time them in ONE GPU visit:
#   tools/variants.sh build  base:"" items8:"-DRS_ITEMS=8" ...        (here, hipcc cross-compiles)
#   tools/variants.sh run    base items8 ...                             (on the GPU box: bench.py --no-cpu --no-legs --no-text per variant)
cd "$(dirname "$0")/.." || exit 1
cmd=$1; shift
case $cmd in
build)
  make lib > /dev/null || exit 1
  for spec in "$@"; do
    name=${spec%%:*}; extra=${spec#*:}
    d=build/variants/$name; mkdir -p $d/obj
    cp -p build/obj/*.o $d/obj/
    rm -f $d/obj/radix.hip.o $d/obj/hits.hip.o $d/obj/graph.hip.o $d/obj/paf.hip.o
    make lib B=$d/obj LIB=$d/libminiasm_amd.so EXTRA="$extra" 2>&1 | grep -E "error|warning" ; ls -la $d/libminiasm_amd.so
  done ;;
run)
  mkdir -p gpurun_out/variants
  for spec in "$@"; do # name, or name+ENV=VALUE (the variant's library with an environment switch on)
    name=${spec//+/_}; vlib=${spec%%+*}; venv=""; [ "$spec" != "$vlib" ] && venv=${spec#*+}
    lib=$PWD/build/variants/$vlib/libminiasm_amd.so
    [ -f $lib ] || { echo "$name: not built"; continue; }
    env $venv MINIASM_AMD_LIB=$lib timeout 600 python bench.py --no-cpu --no-legs --no-text --steps 8 --warmup 2 > gpurun_out/variants/$name.json 2> gpurun_out/variants/$name.log
    python3 - $name <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open("gpurun_out/variants/%s.json" % n))
    ks = {k["name"]: (k["launches_per_step"], k["avg_ms"]) for k in d["kernels"]}
    print("%-12s step %.3f ms | " % (n, d["ms_per_step"]) + "  ".join("%s %gx%.3f" % (k, v[0], v[1]) for k, v in ks.items() if k in ("k_hit_sub<gather>", "k_radix_scatter", "k_radix_hist", "k_hit_keys", "k_runs_expand", "k_hit_sub<cut+flt>", "k_hit_cut_contained")) + ("  identical %s" % d.get("gfa_identical")))
except Exception as e:
    print(n, "failed:", e)
PY
  done ;;
esac
