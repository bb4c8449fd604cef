"""Workload for the counter passes of tools/placement_counters.sh: ONE arena, x in its first block, then the factored
XP sweep with p at each of --offsets-gib behind x, `reps` launches per offset, in that order - so the dispatches of
k_sweep_fast in a rocprofv3 trace are [offset 0] x reps, [offset 1] x reps, ...  Prints the HIP-event time of each.

    python tools/placement_counters.py --offsets-gib 32,96 --reps 3
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import _cabi as cabi  # noqa: E402
from datafusion_amd.kernels import HipKernels  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arena-gib", type=int, default=240)
    ap.add_argument("--shape", default="512,32,4,256,256")
    ap.add_argument("--offsets-gib", default="32,96")
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    shape = tuple(int(v) for v in a.shape.split(","))
    k = HipKernels(*shape, 4, "cuda:0")
    k.new_factored_residual()
    k.ra.zero_(), k.rb.zero_()
    arena = torch.zeros(a.arena_gib << 28, device="cuda:0")
    x = arena[:k.N].view(k.shape)
    for off in (float(v) for v in a.offsets_gib.split(",")):
        o = int(off * (1 << 28))
        p = arena[o:o + k.N].view(k.shape)
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            k.sweep(cabi.SWEEP_XP, x, None, p, factored=True)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            print(f"p {off:6.1f} GiB behind x: sweep {ms:.3f} ms = {16 * k.N / ms / 1e9:.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
