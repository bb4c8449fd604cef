#!/bin/bash
# Block sizes other than 2, 4, 8 and plane widths that are not multiples of 32, at C2 size: row-part tile (16, 32), flat map
# for any r (3, 5, 6, 7, 10, 12), flat map (r = 2, 4, 8 on 120 / 100 / 200-wide planes), generic kernels (Y % 4 != 0, odd
# slab length).  For comparison: DFX_NO_GENERAL_TILE=1 sends the other block sizes to the generic kernels,
# DFX_NT_ANY_WIDTH=1 keeps r = 4 on the non-temporal row map whatever the width, DFX_NO_FLAT_MAP=1 sends the power-of-two
# flat map's shapes to the flat map for any r, DFX_NO_ROWPART_TILE=1 sends 16 / 32 to the flat map, DFX_FLAT_ANY_WIDTH=1
# runs the flat map on aligned rows too.
set -o pipefail
OUT=gpurun_out/general_r_bench.txt; : > $OUT
line() { python3 tools/_bench_line.py; }
for W in c2r3 c2r5 c2r6 c2r7 c2r10 c2r12 c2r16 c2r32 c2y120 c2y100 c2y200 c2r2y120 c2r8y120 c2r3y96 c2r3y192 c2gen c2odd; do
  echo "## bench.py --workload $W --no-cpu --steps 20" >> $OUT
  timeout -k 10 200 python3 bench.py --workload $W --no-cpu --steps 20 2>/dev/null | line >> $OUT || exit 1
done
for W in c2r3 c2r6 c2r12; do
  echo "## DFX_NO_GENERAL_TILE=1 bench.py --workload $W --single --no-cpu --steps 20   (generic kernels)" >> $OUT
  DFX_NO_GENERAL_TILE=1 timeout -k 10 200 python3 bench.py --workload $W --single --no-cpu --steps 20 2>/dev/null | line >> $OUT || exit 1
done
echo "## DFX_NT_ANY_WIDTH=1 bench.py --workload c2y120 --single --no-cpu --steps 20   (non-temporal r = 4 map on 480-byte rows)" >> $OUT
DFX_NT_ANY_WIDTH=1 timeout -k 10 200 python3 bench.py --workload c2y120 --single --no-cpu --steps 20 2>/dev/null | line >> $OUT || exit 1
for W in c2y120 c2y100 c2y200 c2r8y120; do
  echo "## DFX_NO_FLAT_MAP=1 bench.py --workload $W --single --no-cpu --steps 20   (flat map for any r)" >> $OUT
  DFX_NO_FLAT_MAP=1 timeout -k 10 200 python3 bench.py --workload $W --single --no-cpu --steps 20 2>/dev/null | line >> $OUT || exit 1
done
echo "## DFX_FLAT_ANY_WIDTH=1 bench.py --workload c2 --single --no-cpu --steps 20   (flat map on aligned rows)" >> $OUT
DFX_FLAT_ANY_WIDTH=1 timeout -k 10 200 python3 bench.py --workload c2 --single --no-cpu --steps 20 2>/dev/null | line >> $OUT || exit 1
for W in c2r16 c2r32; do
  echo "## DFX_NO_ROWPART_TILE=1 bench.py --workload $W --single --no-cpu --steps 20   (flat map)" >> $OUT
  DFX_NO_ROWPART_TILE=1 timeout -k 10 200 python3 bench.py --workload $W --single --no-cpu --steps 20 2>/dev/null | line >> $OUT || exit 1
done
for W in c2r3 c2r16 c2y120 c2gen; do
  for M in "adam" "adam --moment1 explicit" "adam-autograd"; do
    echo "## bench.py --workload $W --method $M --no-cpu --steps 10" >> $OUT
    timeout -k 10 200 python3 bench.py --workload $W --method $M --no-cpu --steps 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   it/s %.1f  ms/it %.3f  kernels %s' % (d['value'], d['ms_per_step'], ', '.join('%s %.3f' % kv for kv in d['kernel_ms'].items())))" >> $OUT || exit 1
  done
done
cat $OUT
