#!/bin/bash
# Round 4, GPU visit N: the group offsets out of the sort's last pass (RsGroups + two small launches) against the sweep over the sorted keys (k_hit_goff)
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4n; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/tests.log 2>&1; echo "parity rc=$?"; tail -2 $O/tests.log
MINIASM_AMD_LIB=$PWD/build/variants/gf/libminiasm_amd.so timeout 150 python bench.py --no-legs --no-text --steps 4 --warmup 1 > $O/first.json 2> $O/first.log; echo "first fused run rc=$?"
python3 -c "import json; d=json.load(open('$O/first.json')); print('   step %.3f ms identical %s' % (d['ms_per_step'], d.get('gfa_identical')))" || { tail -5 $O/first.log; exit 1; }
tools/variants.sh run gf+MA_GOFF_FUSE=0 gf gf+MA_GOFF_FUSE=0 gf 2>&1 | tee $O/variants.txt
python3 - <<'PY'
import json
for f in ("gf_MA_GOFF_FUSE=0", "gf"):
    d = json.load(open("gpurun_out/variants/%s.json" % f)); ks = {k["name"]: k for k in d["kernels"]}
    print("   %-20s " % f + "  ".join("%s %gx%.3f" % (n, ks[n]["launches_per_step"], ks[n]["avg_ms"]) for n in ("k_radix_scatter", "k_hit_goff", "k_group_close") if n in ks))
PY
