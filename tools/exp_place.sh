#!/bin/bash
set -o pipefail
hostname; rocm-smi --showserial --showclocks 2>/dev/null | grep -i "serial\|sclk\|mclk\|fclk" | head -8
for i in 1 2 3; do
for W in c2 c2y96 c2y160; do
  printf "## $W run $i "; timeout -k 10 200 python3 bench.py --workload $W --single --no-cpu --steps 20 2>/dev/null | python3 tools/_bench_line.py | cut -c1-120
done
done
rocm-smi --showmemuse --showpower 2>/dev/null | grep -i "power\|memory" | head -6
