"""stdin: one bench.py JSON line -> a short human-readable summary (used by tools/bench_*.sh)."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
s = (f"   it/s {d['iters_per_s']:.2f}  ms/it {d['ms_per_step']:.3f} (blocks " + ", ".join(f"{b:.3f}" for b in d["timed_blocks"]["ms_per_step"])
     + f")  sweep {r['avg_launch_ms']:.3f} ms x {d['config']['sweep_launches_per_iteration']} launches  frac {r['frac']:.3f}")
km = d.get("kernel_ms", {})
s += "  kernels " + ", ".join(f"{k} {v:.3f}" for k, v in km.items())
c = d.get("comm")
if c:
    s += (f"\n   comm: slices {c['sweep_slices']}, exposed waits " + ", ".join(f"{k} {v * 1e3:.0f} us" for k, v in c["exposed_wait_ms_per_iteration"].items())
          + f", probe {c['allreduce_probe']}")
print(s)
