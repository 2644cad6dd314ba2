#!/bin/bash
# one input through the command line N times: every run must give the same digest (default: case 141 of tools/fuzz_emu.py --seed 60603, whose digest is the reference's ba2bfcd5...)
# usage: [GEN="pafgen options"] [ARGS="miniasm options"] [VARIANTS="X=1 MA_EXACT_TIES=0"] tools/flake_loop.sh [N]
cd "$(dirname "$0")/.." || exit 1
N=${1:-300}
GEN=${GEN:--r 4000 -n 50000 -s 165948710 -L uniform -d 0.05 -x 0.100 -i 0.10}
ARGS=${ARGS:--m 500 -s 1000 -i 0.00 -h 5000 -p paf -S3}
miniasm_amd/bin/pafgen $GEN -o /tmp/flake.paf 2>/dev/null
[ -x oracle/_ref/miniasm_ref ] && echo "reference: $(oracle/_ref/miniasm_ref $ARGS /tmp/flake.paf 2>/dev/null | md5sum)"
for v in ${VARIANTS:-X=1}; do
  echo "## [$v] pafgen $GEN | miniasm $ARGS"
  for k in $(seq 1 $N); do env $v miniasm_amd/bin/miniasm $ARGS /tmp/flake.paf 2>/dev/null | md5sum; done | sort | uniq -c
done
