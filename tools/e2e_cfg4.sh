#!/bin/bash
# BASELINE configs[3] at 1 GPU, end to end from PAF text: this CLI vs the unmodified reference binary (1 thread), same file.
# usage: tools/e2e_cfg4.sh [reads] [lines] [seed]
cd "$(dirname "$0")/.." || exit 1
R=${1:-2000000}; N=${2:-100000000}; S=${3:-2}
P=/tmp/cfg4.paf
out=gpurun_out/e2e_cfg4.txt
mkdir -p gpurun_out
{
echo "host: $(nproc) cores, $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)"
t0=$(date +%s.%N); miniasm_amd/bin/pafgen -r $R -n $N -s $S -o $P 2>/dev/null; t1=$(date +%s.%N)
echo "pafgen: $(python3 -c "print('%.1f' % ($t1-$t0))") s, $(stat -c %s $P) bytes, $(wc -l < $P) lines"
cat $P > /dev/null
for i in 1 2 3; do
  [ $i = 3 ] && export MA_NO_NUMA_PIN=1
  t0=$(date +%s.%N); MA_PIPE_TIMING=${TIMING:-1} miniasm_amd/bin/miniasm $P > /tmp/cfg4_gpu.gfa 2> /tmp/cfg4_gpu.log; rc=$?; t1=$(date +%s.%N)
  echo "gpu run $i${MA_NO_NUMA_PIN:+ (no NUMA pin)}: rc=$rc wall $(python3 -c "print('%.3f' % ($t1-$t0))") s"
  grep -E "T::|Real time|ma_hit_read|ma_hit_contained|ma_sg_gen" /tmp/cfg4_gpu.log
done
unset MA_NO_NUMA_PIN
[ -n "$NOREF" ] && { echo "raw md5: gpu $(md5sum < /tmp/cfg4_gpu.gfa | cut -c1-32) (reference not run)"; exit 0; }
t0=$(date +%s.%N); timeout 900 taskset -c 2 oracle/_ref/miniasm_ref $P > /tmp/cfg4_ref.gfa 2> /tmp/cfg4_ref.log; rc=$?; t1=$(date +%s.%N)
echo "reference: rc=$rc wall $(python3 -c "print('%.3f' % ($t1-$t0))") s"
grep -E "Real time|ma_hit_read|ma_hit_contained|ma_sg_gen" /tmp/cfg4_ref.log
echo "gfa bytes: gpu $(stat -c %s /tmp/cfg4_gpu.gfa) ref $(stat -c %s /tmp/cfg4_ref.gfa)"
echo "raw md5:        gpu $(md5sum < /tmp/cfg4_gpu.gfa | cut -c1-32) ref $(md5sum < /tmp/cfg4_ref.gfa | cut -c1-32)"
echo "normalised md5: gpu $(LC_ALL=C sort /tmp/cfg4_gpu.gfa | md5sum | cut -c1-32) ref $(LC_ALL=C sort /tmp/cfg4_ref.gfa | md5sum | cut -c1-32)"
echo "arc tie groups in the reference's -S5 graph are not checked here (pafgen keeps coordinates distinct)"
} > $out 2>&1
cat $out
