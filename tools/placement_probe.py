"""Does the duration of one sweep depend on WHERE its vectors were placed?  (DESIGN.md "open observation":
the nZ = 4 slabs of configs 3 / 4 run in a fast or a slow mode, fixed per process.)

Allocates the x, p pair several times inside ONE process - keeping or releasing the previous pairs so the driver hands
out different physical ranges - and times dfx_sweep_factored(XP) on each pair.

    python tools/placement_probe.py --shape 512,32,4,256,256 --rounds 8
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import _cabi as cabi  # noqa: E402
from datafusion_amd.kernels import HipKernels  # noqa: E402


def time_sweep(k, x, p, reps):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        k.sweep(cabi.SWEEP_XP, x, None, p, factored=True)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="512,32,4,256,256")
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--keep", type=int, default=3, help="keep at most this many earlier pairs allocated")
    ap.add_argument("--arena-gib", type=int, default=0,
                    help="instead: ONE allocation of this size; x and p are windows sliding through it (--stride-gib)")
    ap.add_argument("--stride-gib", type=float, default=4.0)
    ap.add_argument("--delta-scan", action="store_true", help="arena mode: keep x fixed, move p")
    ap.add_argument("--x-gib", type=float, default=0.0)
    a = ap.parse_args()
    shape = tuple(int(v) for v in a.shape.split(","))
    k = HipKernels(*shape, 4, "cuda:0")
    k.new_factored_residual()
    k.ra.zero_(), k.rb.zero_()
    nbytes = 4 * k.N
    if a.arena_gib:
        arena = torch.zeros(a.arena_gib << 28, device="cuda:0")          # floats
        stride = int(a.stride_gib * (1 << 28))
        if a.delta_scan:
            # x fixed at the start of the arena (or --x-gib), p at every stride further on: does the DISTANCE matter?
            x0 = int(a.x_gib * (1 << 28))
            x = arena[x0:x0 + k.N].view(k.shape)
            d = k.N
            while x0 + d + k.N <= arena.numel():
                p = arena[x0 + d:x0 + d + k.N].view(k.shape)
                ms = time_sweep(k, x, p, a.reps)
                print(f"x at {a.x_gib:5.1f} GiB, p {d / (1 << 28):6.2f} GiB behind: sweep {ms:.3f} ms = "
                      f"{4 * nbytes / ms / 1e6:.0f} GB/s", flush=True)
                d += stride
            return
        off = 0
        while off + 2 * k.N <= arena.numel():
            x = arena[off:off + k.N].view(k.shape)
            # one vector alone: read-only projection, and read + write in place (x += 0 * x)
            best_r = best_rw = 1e9
            for _ in range(a.reps):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record()
                k.sweep(cabi.SWEEP_PROJECT, x)
                e[1].record()
                k.axpy(x, x, 0.0, None)
                e[2].record()
                torch.cuda.synchronize()
                best_r, best_rw = min(best_r, e[0].elapsed_time(e[1])), min(best_rw, e[1].elapsed_time(e[2]))
            print(f"arena offset {off / (1 << 28):6.1f} GiB x alone: read {nbytes / best_r / 1e6:.0f} GB/s, "
                  f"read+write {2 * nbytes / best_rw / 1e6:.0f} GB/s", flush=True)
            for gap in (0, 1 << 20):                                      # p right behind x, and 4 MiB further
                if off + 2 * k.N + gap > arena.numel():
                    continue
                p = arena[off + k.N + gap:off + 2 * k.N + gap].view(k.shape)
                ms = time_sweep(k, x, p, a.reps)
                print(f"arena offset {off / (1 << 28):6.1f} GiB gap {gap * 4 >> 20} MiB: sweep {ms:.3f} ms = "
                      f"{4 * nbytes / ms / 1e6:.0f} GB/s", flush=True)
            off += stride
        return
    kept = []
    for i in range(a.rounds):
        x = torch.zeros(k.shape, device="cuda:0")
        p = torch.zeros(k.shape, device="cuda:0")
        ms = time_sweep(k, x, p, a.reps)
        print(f"round {i}: x @0x{x.data_ptr():x} p @0x{p.data_ptr():x}  sweep {ms:.3f} ms = "
              f"{4 * nbytes / ms / 1e6:.0f} GB/s", flush=True)
        kept.append((x, p))
        if len(kept) > a.keep:
            kept.pop(0)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
