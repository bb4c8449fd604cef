#!/bin/bash
# planes per z-chunk (DFX_PLAN_ZS) against the planner's own choice, sweep of the factored CG iteration
set -o pipefail
for W in ${WORKLOADS:-f128m1k f128z8 f128z32 f64z16 f256z16}; do
  for Z in default 2 4; do
    if [ $Z = default ]; then unset DFX_PLAN_ZS; else export DFX_PLAN_ZS=$Z; fi
    printf "## $W ZS=$Z: "; timeout -k 10 200 python3 bench.py --workload $W --single --no-cpu --steps 20 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); pl=d['config']['plan']; print('%.1f it/s sweep %.3f ms frac %.3f  plan zs? zc %d nzc %d blocks %d cpb %d' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], pl['zc'], pl['n_zchunks'], pl['n_blocks'], pl['cpb']))"
  done
done
