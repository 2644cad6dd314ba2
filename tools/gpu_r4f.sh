#!/bin/bash
# Round 4, GPU visit F: occupancy variants of the gather / fused tiers on the batched-gather code, sharded ingest at cfg4 on 2 ranks (chunked shm exchange), projection N = 8 with a time limit
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O
t0=$(date +%s)
lap() { echo "## $1: $(( $(date +%s) - t0 )) s since start"; }
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/tests.log 2>&1; echo "parity rc=$?"; tail -2 $O/tests.log
tools/variants.sh run base g5 g7 f6 f8 base 2>&1 | tee $O/variants.txt
lap variants
P=/tmp/ma_bench/w_lognormal_r2000000_n100000000_s2.paf
for mode in one ranges2; do
  case $mode in one) e="";; ranges2) e="MA_GPUS=2 MA_COMM=shm";; esac
  ts=$(date +%s.%N)
  env $e MA_PIPE_TIMING=1 timeout 300 miniasm_amd/bin/miniasm $P 2> $O/cli_$mode.log | md5sum | cut -c1-32 > $O/cli_$mode.md5
  te=$(date +%s.%N)
  echo "cli $mode: $(cat $O/cli_$mode.md5)  wall $(python3 -c "print('%.3f' % ($te - $ts))") s"
  grep -E "T::ingest_gpu|E::" $O/cli_$mode.log | sed 's/^/      /' | cut -c1-230 | head -4
done
lap cli
timeout 400 python tools/shard_projection.py --ranks 1,8 --steps 4 --per-n-timeout 150 --out $O/shard_projection.json > $O/projection.log 2>&1; grep -E "^N=|failed|Error" $O/projection.log | head; tail -3 $O/projection.log | cut -c1-300
lap projection
