#!/bin/bash
# Round 4, GPU visit A: quick parity, the graph-heavy leg (new arc path vs the radix / three-launch forms), sort-group variants, rocprof of a
# graph-heavy CLI run, then the default bench line.  Everything lands in gpurun_out/r4a/.
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
rocminfo 2>/dev/null | grep -m1 -E "gfx9" > $O/gpu.txt
t0=$(date +%s)
lap() { echo "## $1: $(( $(date +%s) - t0 )) s since start"; }

timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph_api.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
grep -vE "^\[M::|^\[pafgen" $O/tests.log | tail -5
lap tests

for tag in new radix radix_oldrm; do
  case $tag in new) v="";; radix) v="MA_ARC_RADIX=1";; radix_oldrm) v="MA_ARC_RADIX=1 MA_ARC_RM_OLD=1";; esac
  env $v timeout 900 python bench.py --reads 200000 --lines 10000000 --seed 1 --no-cpu --no-text --legs graph_heavy --steps 10 --warmup 2 > $O/gh_$tag.json 2> $O/gh_$tag.log; echo "gh_$tag rc=$?"
  python3 - $O/gh_$tag.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); g = d["legs"]["graph_heavy"]; r = g["reduce_group"]
    print("   step %.2f ms  arcs %s  I %s  reduce_group %.2f ms  frac %.3f (%s B/arc)" % (g["ms_per_step"], g["arcs"], r["inner_iterations_I"], r["ms_per_step"], r["frac"], r["bytes_per_arc"]))
    for k in g["kernels"][:14]: print("     %-24s x%-4g %8.3f ms  alg %s  design %s" % (k["name"], k["launches_per_step"], k["avg_ms"], k["alg_GBs"], k["design_GBs"]))
except Exception as e:
    print("   failed:", e)
PY
done
lap graph_heavy

tools/variants.sh run base nt base+MA_GATHER_APART=1 nt+MA_GATHER_APART=1 base+MA_NO_GATHER_FUSE=1 2>&1 | tee $O/variants.txt
lap variants

GH=$(ls /tmp/ma_bench/leg_graph_heavy_*.paf 2>/dev/null | head -1)
if [ -n "$GH" ]; then
  rm -rf $O/prof_gh; mkdir -p $O/prof_gh
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_gh -o r --output-format csv -- /root/repo/miniasm_amd/bin/miniasm $GH > /tmp/gh_cli.gfa 2> /root/repo/$O/prof_gh/run.log); echo "rocprof gh rc=$?"
  f=$(find $O/prof_gh -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_cli_graph_heavy.csv && head -16 $f | cut -c1-140
  find $O/prof_gh -name "*.csv" ! -name "*stats*" -delete 2>/dev/null
fi
lap rocprof

timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log; echo "bench rc=$?"
python3 - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r4a/bench_default.json"))
    print("ms_per_step %.3f  value %.4g  gfa_identical %s  latency %s  e2e %s  from_text %s" % (d["ms_per_step"], d["value"], d["gfa_identical"], d.get("latency") and d["latency"]["ms"], d.get("e2e") and round(d["e2e"]["wall_s"], 3), d.get("from_text") and round(d["from_text"]["ms_per_step"], 2)))
    r = d["roofline"]; print("roofline: %s %.3f ms frac %.3f | sort_group %.3f ms frac %.3f | hit_chain %.3f" % (r["kernel"], r["avg_launch_ms"], r["frac"], r["sort_group"]["ms_per_step"], r["sort_group"]["frac"], r["hit_chain"]["frac"]))
    rg = r.get("reduce_group"); print("reduce_group:", rg and (rg["ms_per_step"], rg["frac"], rg["slowest_by_8d"]))
    for n, l in d["legs"].items(): print("leg %-12s %.3f ms/step  identical %s  cpu %s" % (n, l["ms_per_step"], l.get("gfa_identical"), l.get("cpu_overlaps_per_s")))
    for k in d["kernels"]: print("   %-24s x%-4g %8.3f ms" % (k["name"], k["launches_per_step"], k["avg_ms"]))
except Exception as e:
    print("bench summary failed:", e)
PY
lap bench
