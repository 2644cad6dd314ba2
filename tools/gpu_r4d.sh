#!/bin/bash
# Round 4, GPU visit D: sharded ingest on the device (N ranks on the one GPU through the shared-memory double), projection with the best-of-steps statistic
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O
t0=$(date +%s)
lap() { echo "## $1: $(( $(date +%s) - t0 )) s since start"; }
timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_ingest.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?"
grep -vE "^\[M::|^\[pafgen" $O/tests.log | tail -4
lap tests
P=/tmp/ma_bench/w_lognormal_r2000000_n100000000_s2.paf
mkdir -p /tmp/ma_bench; [ -f $P ] || miniasm_amd/bin/pafgen -r 2000000 -n 100000000 -s 2 -o $P 2>/dev/null
cat $P > /dev/null
for mode in one ranges2 whole2 ranges4 ranges8; do
  case $mode in one) e="";; ranges2) e="MA_GPUS=2 MA_COMM=shm";; whole2) e="MA_GPUS=2 MA_COMM=shm MA_INGEST_WHOLE=1";; ranges4) e="MA_GPUS=4 MA_COMM=shm";; ranges8) e="MA_GPUS=8 MA_COMM=shm";; esac
  ts=$(date +%s.%N)
  env $e MA_PIPE_TIMING=1 timeout 600 miniasm_amd/bin/miniasm $P 2> $O/cli_$mode.log | md5sum | cut -c1-32 > $O/cli_$mode.md5
  te=$(date +%s.%N)
  echo "cli $mode: $(cat $O/cli_$mode.md5)  wall $(python3 -c "print('%.3f' % ($te - $ts))") s"
  grep -E "T::ingest_gpu" $O/cli_$mode.log | sed 's/^/      /' | cut -c1-230 | head -8
done
lap cli
timeout 1200 python tools/shard_projection.py --ranks 1,2,4,8 --steps 4 --out $O/shard_projection.json 2>&1 | grep -E "^N="
lap projection
