#!/bin/bash
# Round 4, GPU visit K: k_radix_scatter with the digit width at compile time (unrolled ballot steps, 128-bin LDS tables: 4 blocks per CU), k_radix_hist with a table per wave
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
O=gpurun_out/r4k; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/tests.log 2>&1; echo "parity rc=$?"; tail -2 $O/tests.log
tools/variants.sh run head nb7 nb7hw head nb7 nb7hw 2>&1 | tee $O/variants.txt
