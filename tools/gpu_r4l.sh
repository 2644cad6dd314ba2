#!/bin/bash
# Round 4, GPU visit L: radix passes 2 and 3 as chained tiles on digit totals counted by the keys kernel (no k_radix_hist sweeps, no scans) against the two-phase passes
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4l; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/tests.log 2>&1; echo "parity rc=$?"; tail -2 $O/tests.log
MINIASM_AMD_LIB=$PWD/build/variants/chain/libminiasm_amd.so timeout 150 python bench.py --no-legs --no-text --steps 4 --warmup 1 > $O/first.json 2> $O/first.log; rc=$?; echo "first chained run rc=$rc"
python3 -c "import json; d=json.load(open('$O/first.json')); print('   step %.3f ms identical %s' % (d['ms_per_step'], d.get('gfa_identical')))" || { tail -5 $O/first.log; exit 1; }
tools/variants.sh run nb7 chain chain+MA_RADIX_CHAIN=0 nb7 chain 2>&1 | tee $O/variants.txt
env MINIASM_AMD_LIB=$PWD/build/variants/chain/libminiasm_amd.so timeout 200 python bench.py --reads 200000 --lines 10000000 --seed 1 --no-cpu --no-legs --no-text --steps 20 --warmup 4 > $O/c2.json 2> $O/c2.log; echo "cfg2 chain rc=$?"
env MINIASM_AMD_LIB=$PWD/build/variants/nb7/libminiasm_amd.so timeout 200 python bench.py --reads 200000 --lines 10000000 --seed 1 --no-cpu --no-legs --no-text --steps 20 --warmup 4 > $O/c2b.json 2> $O/c2b.log; echo "cfg2 nb7 rc=$?"
python3 - <<'PY'
import json
for f in ("c2", "c2b"):
    d = json.load(open("gpurun_out/r4l/%s.json" % f)); ks = {k["name"]: k for k in d["kernels"]}
    print("   %s step %.3f ms | " % (f, d["ms_per_step"]) + "  ".join("%s %gx%.3f" % (n, ks[n]["launches_per_step"], ks[n]["avg_ms"]) for n in ("k_hit_keys", "k_radix_scatter", "k_radix_hist", "k_hit_goff") if n in ks))
PY
