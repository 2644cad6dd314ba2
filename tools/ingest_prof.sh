#!/bin/bash
# the ingest kernels alone under rocprofv3 --stats, one variant per environment setting: usage tools/ingest_prof.sh "VAR=1" "OTHER=2" ...  ("" = default)
# workload: BASELINE configs[3]'s text (100 M lines, 6 GB); each variant parses it three times (file -> HBM -> records + dictionary)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
P=/tmp/ingest_prof.paf
[ -f $P ] || miniasm_amd/bin/pafgen -r ${READS:-2000000} -n ${LINES:-100000000} -s 2 $GENEXTRA -o $P 2>/dev/null
cat > /tmp/ingest_prof.py <<'PY'
import sys, time
sys.path.insert(0, '/root/repo')
import miniasm_amd as ma
ctx = ma.Ctx(0)
for k in range(3):
    t0 = time.perf_counter(); g = ma.GpuIngest(ctx, sys.argv[1]); dt = time.perf_counter() - t0
    print("ingest %d: %.3f s, %d records, %d reads" % (k, dt, g.n, g.n_seq), flush=True); g.close()
ctx.close()
PY
for v in "$@"; do
  tag=$(echo "${v:-default}" | tr -c 'A-Za-z0-9=_\n' '_')
  rm -rf gpurun_out/iprof_$tag; mkdir -p gpurun_out/iprof_$tag
  (cd /tmp && env ${v:-X=1} timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/iprof_$tag -o r --output-format csv -- python /tmp/ingest_prof.py $P > /root/repo/gpurun_out/iprof_$tag/run.log 2>&1); echo "[${v:-default}] rc=$? $(grep -c ingest gpurun_out/iprof_$tag/run.log) runs: $(grep 'ingest 2' gpurun_out/iprof_$tag/run.log)"
  f=$(find gpurun_out/iprof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 tools/kstats.py "$f" 8
  find gpurun_out/iprof_$tag -name "*trace*.csv" -size +2M -delete
done
