#!/bin/bash
# large single-GPU run from PAF text (BASELINE configs[4] scale): our CLI only, optionally the reference
# usage: tools/e2e_big.sh reads lines seed "<pafgen extra>" [ref]
cd "$(dirname "$0")/.." || exit 1
R=$1; N=$2; S=$3; EXTRA=$4; REF=$5
P=/tmp/big.paf
out=gpurun_out/e2e_big.txt
mkdir -p gpurun_out
{
echo "host: $(nproc) cores, $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2); free RAM $(free -g | awk '/Mem:/{print $7}') GB"
t0=$(date +%s.%N); miniasm_amd/bin/pafgen -r $R -n $N -s $S $EXTRA -o $P 2>/dev/null; t1=$(date +%s.%N)
echo "pafgen -r $R -n $N -s $S $EXTRA: $(python3 -c "print('%.1f' % ($t1-$t0))") s, $(stat -c %s $P) bytes"
for mode in default exact; do
  [ $mode = exact ] && export MA_EXACT_TIES=1
  t0=$(date +%s.%N); MA_PIPE_TIMING=1 timeout 600 miniasm_amd/bin/miniasm $P > /tmp/big_$mode.gfa 2> /tmp/big_$mode.log; rc=$?; t1=$(date +%s.%N)
  echo "gpu ($mode ties): rc=$rc wall $(python3 -c "print('%.3f' % ($t1-$t0))") s"
  grep -E "T::|Real time|ma_hit_read|ma_hit_contained|ma_sg_gen|E::" /tmp/big_$mode.log
  unset MA_EXACT_TIES
done
rocm-smi --showmemuse 2>/dev/null | grep -i "GPU\[0\]" | head -2
if [ -n "$REF" ]; then
  t0=$(date +%s.%N); timeout 1500 taskset -c 2 oracle/_ref/miniasm_ref $P > /tmp/big_ref.gfa 2> /tmp/big_ref.log; rc=$?; t1=$(date +%s.%N)
  echo "reference: rc=$rc wall $(python3 -c "print('%.3f' % ($t1-$t0))") s"
  grep -E "Real time|ma_hit_read|ma_hit_contained|ma_sg_gen" /tmp/big_ref.log
  echo "raw md5:        default $(md5sum < /tmp/big_default.gfa | cut -c1-32) exact $(md5sum < /tmp/big_exact.gfa | cut -c1-32) ref $(md5sum < /tmp/big_ref.gfa | cut -c1-32)"
  echo "normalised md5: default $(LC_ALL=C sort /tmp/big_default.gfa | md5sum | cut -c1-32) exact $(LC_ALL=C sort /tmp/big_exact.gfa | md5sum | cut -c1-32) ref $(LC_ALL=C sort /tmp/big_ref.gfa | md5sum | cut -c1-32)"
else
  echo "md5: default $(md5sum < /tmp/big_default.gfa | cut -c1-32) exact $(md5sum < /tmp/big_exact.gfa | cut -c1-32); bytes $(stat -c %s /tmp/big_default.gfa)"
fi
} > $out 2>&1
cat $out
