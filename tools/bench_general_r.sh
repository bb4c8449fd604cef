#!/bin/bash
# Block sizes other than 2, 4, 8 at C2 size: row-part tile (16, 32), strip tile (3, 5, 6, 7, 10, 12), generic kernels
# (Y % 4 != 0, odd slab length); DFX_NO_GENERAL_TILE=1 sends the strip-tile sizes to the generic kernels for comparison.
set -o pipefail
OUT=gpurun_out/general_r_bench.txt; : > $OUT
for W in c2r3 c2r5 c2r6 c2r7 c2r10 c2r12 c2r16 c2r32 c2gen c2odd; do
  echo "## bench.py --workload $W --no-cpu --steps 20" >> $OUT
  timeout -k 10 200 python3 bench.py --workload $W --no-cpu --steps 20 2>/dev/null | python3 tools/_bench_line.py >> $OUT || exit 1
done
for W in c2r3 c2r5 c2r6 c2r7 c2r10 c2r12; do
  echo "## DFX_NO_GENERAL_TILE=1 bench.py --workload $W --single --no-cpu --steps 20   (generic kernels)" >> $OUT
  DFX_NO_GENERAL_TILE=1 timeout -k 10 200 python3 bench.py --workload $W --single --no-cpu --steps 20 2>/dev/null | python3 tools/_bench_line.py >> $OUT || exit 1
done
cat $OUT
