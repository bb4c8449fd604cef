#!/bin/bash
# experiment: one workgroup per CU through a dynamic LDS request (DFX_SOLO_LDS bytes; default 96 KiB, 0 = off)
set -o pipefail
python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print("shared_memory_per_block", p.shared_memory_per_block, "optin", getattr(p, "shared_memory_per_block_optin", None), "per CU", getattr(p, "shared_memory_per_multiprocessor", None))
PY
for W in p512s p512m c4g8; do
for D in 0 default; do
  echo "## $W DFX_SOLO_LDS=$D"
  if [ $D = default ]; then unset DFX_SOLO_LDS; else export DFX_SOLO_LDS=$D; fi
  timeout -k 10 240 python bench.py --workload $W --single --no-cpu --steps 20 --warmup 3 2>/dev/null | python tools/_bench_line.py || exit 1
  for M in "adam" "adam-autograd"; do
  timeout -k 10 200 python bench.py --workload $W --method $M --no-cpu --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', '$M', round(d[\"value\"],1), round(d[\"ms_per_step\"],3), {k: round(v,3) for k,v in d[\"kernel_ms\"].items()})" || exit 1
  done
done
done
