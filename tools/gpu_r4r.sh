#!/bin/bash
# Round 4, GPU visit R: is it the LDS atomics or the digit-major write-out (128 words a tile-count apart) that costs k_radix_hist a quarter of a plain read's rate?
cd "$(dirname "$0")/.." || exit 1
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/r4r
timeout 200 python tools/pmc_calibrate.py run --sizes 1600,3200 --reps 5 --patterns read16_x8,read16_x8+lds_atomics,read16_x8+lds_atomics+rows_out,read16_x8+lds_atomics_x2 2>&1 | tee gpurun_out/r4r/hist_out.txt
