"""Pair map of the placement effect (DESIGN.md "Placement of large vectors"): inside ONE arena cut into blocks of
one vector each, time the factored XP sweep for x in block i and p in block j, every i < j.  If "fast iff the two
vectors lie in different regions" holds, the matrix decomposes into classes of blocks.

    python tools/placement_map.py --arena-gib 240 --nt 256          # 8 GiB vectors (t256 x l32 x z4 x 256 x 256)
    python tools/placement_map.py --arena-gib 240 --nt 128 --max-blocks 24

Also prints, per block, a single vector streamed alone (read-only projection; read + write in place).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import _cabi as cabi  # noqa: E402
from datafusion_amd.kernels import HipKernels  # noqa: E402


def t_sweep(k, x, p, reps):
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
    ap.add_argument("--arena-gib", type=int, default=240)
    ap.add_argument("--nt", type=int, default=256, help="time bins: the vector is t<nt> x l32 x z4 x 256 x 256 (nt/32 GiB)")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--max-blocks", type=int, default=64)
    ap.add_argument("--alone", action="store_true", help="also time each block's vector alone")
    a = ap.parse_args()
    k = HipKernels(a.nt, 32, 4, 256, 256, 4, "cuda:0")
    k.new_factored_residual()
    k.ra.zero_(), k.rb.zero_()
    arena = torch.zeros(a.arena_gib << 28, device="cuda:0")
    nb = min(arena.numel() // k.N, a.max_blocks)
    nbytes = 4 * k.N
    print(f"# arena {a.arena_gib} GiB at 0x{arena.data_ptr():x}, vector {nbytes / 2**30:.1f} GiB, {nb} blocks; "
          f"entries: TB/s of the factored XP sweep (16 B per element) with x in block i (row), p in block j (column)",
          flush=True)
    blk = [arena[i * k.N:(i + 1) * k.N].view(k.shape) for i in range(nb)]
    if a.alone:
        for i in range(nb):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            k.sweep(cabi.SWEEP_PROJECT, blk[i])
            e[1].record()
            k.axpy(blk[i], blk[i], 0.0, None)
            e[2].record()
            torch.cuda.synchronize()
            print(f"alone {i:2d}: read {nbytes / e[0].elapsed_time(e[1]) / 1e9:.2f}  read+write "
                  f"{2 * nbytes / e[1].elapsed_time(e[2]) / 1e9:.2f} TB/s", flush=True)
    print("      " + " ".join(f"{j:4d}" for j in range(nb)))
    for i in range(nb):
        row = []
        for j in range(nb):
            if j == i:
                row.append("   .")
                continue
            if j < i:
                row.append("    ")
                continue
            ms = t_sweep(k, blk[i], blk[j], a.reps)
            row.append(f"{4 * nbytes / ms / 1e9:4.1f}")
        print(f"{i:4d}: " + " ".join(row), flush=True)


if __name__ == "__main__":
    main()
