#!/bin/bash
# Round 4, GPU visit S: the radix passes' digit counts as a row per tile + a scan down the columns (variant rows) against digit-major counts + scan_exclusive_u32 (variant gf = commit 926884d)
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4s; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/tests.log 2>&1; echo "parity rc=$?"; tail -2 $O/tests.log
MINIASM_AMD_LIB=$PWD/build/variants/rows/libminiasm_amd.so timeout 150 python bench.py --no-legs --no-text --steps 4 --warmup 1 > $O/first.json 2> $O/first.log; echo "first rows run rc=$?"
python3 -c "import json; d=json.load(open('$O/first.json')); print('   step %.3f ms identical %s' % (d['ms_per_step'], d.get('gfa_identical')))" || { tail -5 $O/first.log; exit 1; }
tools/variants.sh run gf rows gf rows 2>&1 | tee $O/variants.txt
python3 - <<'PY'
import json
for f in ("gf", "rows"):
    d = json.load(open("gpurun_out/variants/%s.json" % f)); ks = {k["name"]: k for k in d["kernels"]}
    print("   %-6s " % f + "  ".join("%s %gx%.3f" % (n, ks[n]["launches_per_step"], ks[n]["avg_ms"]) for n in ("k_hit_keys", "k_radix_hist", "k_radix_scatter", "k_radix_colscan", "scan_exclusive_u32", "k_group_close") if n in ks))
PY
