#!/bin/bash
# BASELINE configs[4] (500 M overlaps, high-repeat), the configs[2] stand-in (40 M) and the 50 M noisy input through the CLI at HEAD, each against
# the unmodified reference on the same file (raw + normalised md5).  The reference runs (7 min for cfg5) go to the background on their own cores
# while the GPU works; everything lands in gpurun_out/e2e_<tag>.txt.   usage: tools/e2e_three.sh [tags...]   (default: cfg3 noisy50 cfg5)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
declare -A SPEC=( [cfg3]="1200000 40000000 4" [noisy50]="1000000 50000000 3 -L uniform -d 0.35 -x 0.03" [cfg5]="5000000 500000000 3 -L uniform -d 0.35 -x 0.03" )
tags="${*:-cfg3 noisy50 cfg5}"
BIN=${MA_BIN:-miniasm_amd/bin/miniasm}; DIV=${MA_E2E_DIV:-1} # (a dry run of this script on the CPU build: MA_BIN=tests/emu/_build/miniasm MA_E2E_DIV=1000)
now() { date +%s.%N; }
el() { python3 -c "print('%.3f' % ($2-$1))"; }
core=2
for t in $tags; do # generators + references in the background, one core each
  ( set -- ${SPEC[$t]}; R=$(($1/DIV)); N=$(($2/DIV)); S=$3; shift 3
    t0=$(now); miniasm_amd/bin/pafgen -r $R -n $N -s $S "$@" -o /tmp/e2e_$t.paf 2>/dev/null; t1=$(now)
    echo "pafgen -r $R -n $N -s $S $*: $(el $t0 $t1) s, $(stat -c %s /tmp/e2e_$t.paf) bytes" > /tmp/e2e_$t.gen
    touch /tmp/e2e_$t.ready
    t0=$(now); timeout 2400 taskset -c $core oracle/_ref/miniasm_ref /tmp/e2e_$t.paf > /tmp/e2e_$t.ref.gfa 2> /tmp/e2e_$t.ref.log; rc=$?; t1=$(now)
    echo "reference (1 thread, core $core): rc=$rc wall $(el $t0 $t1) s" > /tmp/e2e_$t.ref.txt
    touch /tmp/e2e_$t.refdone ) &
  core=$((core+1))
done
for t in $tags; do
  while [ ! -f /tmp/e2e_$t.ready ]; do sleep 1; done
  out=gpurun_out/e2e_$t.txt
  { echo "host: $(nproc) cores, $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2); free RAM $(free -g | awk '/Mem:/{print $7}') GB"
    cat /tmp/e2e_$t.gen
    modes="default"; [ $t != cfg5 ] && modes="default exact"
    for mode in $modes; do
      [ $mode = exact ] && export MA_EXACT_TIES=1
      t0=$(now); MA_PIPE_TIMING=2 timeout 900 $BIN /tmp/e2e_$t.paf > /tmp/e2e_$t.$mode.gfa 2> /tmp/e2e_$t.$mode.log; rc=$?; t1=$(now)
      echo "gpu ($mode ties): rc=$rc wall $(el $t0 $t1) s"
      grep -E "T::|Real time|ma_hit_read|ma_hit_contained|ma_sg_gen|asg_pop_bubble|asg_cut|E::" /tmp/e2e_$t.$mode.log
      unset MA_EXACT_TIES
    done
    rocm-smi --showmemuse 2>/dev/null | grep -i "GPU\[0\]" | head -2
  } > $out 2>&1
  tail -4 $out
done
for t in $tags; do # the references: wait, compare
  while [ ! -f /tmp/e2e_$t.refdone ]; do sleep 2; done
  out=gpurun_out/e2e_$t.txt
  { cat /tmp/e2e_$t.ref.txt
    grep -E "Real time|ma_hit_read|ma_hit_contained|ma_sg_gen" /tmp/e2e_$t.ref.log
    for mode in default exact; do
      [ -f /tmp/e2e_$t.$mode.gfa ] || continue
      echo "raw md5:        $mode $(md5sum < /tmp/e2e_$t.$mode.gfa | cut -c1-32) ref $(md5sum < /tmp/e2e_$t.ref.gfa | cut -c1-32)   ($(stat -c %s /tmp/e2e_$t.$mode.gfa) vs $(stat -c %s /tmp/e2e_$t.ref.gfa) bytes)"
      echo "normalised md5: $mode $(LC_ALL=C sort /tmp/e2e_$t.$mode.gfa | md5sum | cut -c1-32) ref $(LC_ALL=C sort /tmp/e2e_$t.ref.gfa | md5sum | cut -c1-32)"
    done
  } >> $out 2>&1
  echo "== $t"; tail -6 $out
done
wait
