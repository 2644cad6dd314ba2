#!/bin/bash
# Round 4, final GPU visit (evidence for profiles/ at the round's last kernel commit): rocprofv3 kernel stats and PMC traffic of the bench command at BASELINE configs[3] and on the graph-heavy
# input (main workload = pafgen -L fixed), the default bench line.  usage: tools/gpu_r4z.sh [tests] (with "tests": the whole -m gpu suite first)
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4z; mkdir -p $O
t0=$(date +%s)
lap() { echo "## $1: $(( $(date +%s) - t0 )) s since start"; }
if [ "$1" = "tests" ]; then
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/tests_all.log 2>&1; echo "all gpu tests rc=$?"
  grep -vE "^\[M::|^\[pafgen" $O/tests_all.log | tail -4
  lap tests
fi
GH="--reads 2000000 --lines 100000000 --seed 4 --model fixed"
for w in cfg4 gh; do
  case $w in cfg4) A="";; gh) A="$GH";; esac
  rm -rf $O/prof_$w; mkdir -p $O/prof_$w
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_$w -o r --output-format csv -- python /root/repo/bench.py $A --steps 5 --warmup 1 --no-cpu --no-legs --no-text --prof-steps 0 > /root/repo/$O/prof_$w/bench.json 2> /root/repo/$O/prof_$w/bench.log); echo "rocprof $w rc=$?"
  f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_$w.csv
  find $O/prof_$w -name "*.csv" ! -name "*stats*" -delete 2>/dev/null
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${w}_$ctr; mkdir -p $O/pmc_${w}_$ctr
    (cd /tmp && timeout 900 rocprofv3 --pmc $ctr --kernel-trace -d /root/repo/$O/pmc_${w}_$ctr -o r --output-format csv -- python /root/repo/bench.py $A --steps 2 --warmup 1 --no-cpu --no-legs --no-text --prof-steps 0 > /root/repo/$O/pmc_${w}_$ctr/bench.json 2> /root/repo/$O/pmc_${w}_$ctr/bench.log); echo "pmc $w $ctr rc=$?"
  done
  python tools/pmc_summary.py $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE > $O/pmc_traffic_$w.json 2> $O/pmc_$w.log; tail -1 $O/pmc_$w.log
  find $O/pmc_${w}_FETCH_SIZE $O/pmc_${w}_WRITE_SIZE -name "*.csv" -size +1M -delete 2>/dev/null
  lap "profiles $w"
done
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.log; echo "bench rc=$?"
python3 - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r4z/bench_default.json"))
    print("ms_per_step %.3f  value %.4g  gfa_identical %s  latency %s  e2e %s  from_text %s" % (d["ms_per_step"], d["value"], d["gfa_identical"], d.get("latency") and d["latency"]["ms"], d.get("e2e") and round(d["e2e"]["wall_s"], 3), d.get("from_text") and round(d["from_text"]["ms_per_step"], 2)))
    r = d["roofline"]; print("roofline: %s %.3f ms frac %.3f | sort_group %.3f ms frac %.3f | hit_chain %.3f" % (r["kernel"], r["avg_launch_ms"], r["frac"], r["sort_group"]["ms_per_step"], r["sort_group"]["frac"], r["hit_chain"]["frac"]))
    rg = r.get("reduce_group"); print("reduce_group:", rg and (rg["ms_per_step"], rg["frac"], rg["slowest_by_8d"]))
    for n, l in d["legs"].items(): print("leg %-12s %.3f ms/step  identical %s  cpu %s" % (n, l["ms_per_step"], l.get("gfa_identical"), l.get("cpu_overlaps_per_s")))
    print("cpu_baseline", d["cpu_baseline"] and d["cpu_baseline"]["value"])
except Exception as e:
    print("bench summary failed:", e)
PY
lap bench
