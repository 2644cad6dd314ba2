#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
miniasm_amd/bin/pafgen -r 3000 -n 80000 -s 41 -o /tmp/ds.paf 2>/dev/null
fail=0
for i in $(seq 1 40); do
  for a in "-p paf" "-p bed" "-p ug"; do
    "$@" oracle/_ref/miniasm_dropin $a /tmp/ds.paf > /tmp/ds.out 2> /tmp/ds.err; rc=$?
    if [ $rc -ne 0 ]; then fail=$((fail+1)); echo "run $i [$a] rc=$rc"; tail -2 /tmp/ds.err; fi
  done
done
echo "failures: $fail / 120"
which gdb valgrind 2>/dev/null
