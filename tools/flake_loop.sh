#!/bin/bash
# one fuzz case (tools/fuzz_emu.py --seed 60603, case 141) through the command line N times: the digests must all be the reference's (ba2bfcd5...)
cd "$(dirname "$0")/.." || exit 1
N=${1:-300}
miniasm_amd/bin/pafgen -r 4000 -n 50000 -s 165948710 -L uniform -d 0.05 -x 0.100 -i 0.10 -o /tmp/m.paf 2>/dev/null
A="-m 500 -s 1000 -i 0.00 -h 5000 -p paf -S3"
for v in ${VARIANTS:-X=1}; do
  echo "## [$v]"
  for k in $(seq 1 $N); do env $v miniasm_amd/bin/miniasm $A /tmp/m.paf 2>/dev/null | md5sum; done | sort | uniq -c
done
