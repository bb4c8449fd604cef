"""GPU experiment: does striping EACH vector over the three memory classes of an MI355X (datafusion_amd/placement.py)
at a granularity below the sweep's concurrent window beat "x whole in one class, p whole in another" (6.4 TB/s)?

The sweep streams a window of gz adjacent slabs of x and of p at any instant (C2: 8 slabs of 1 MiB).  With physical
chunks of 2 MiB mapped behind the vectors in a chosen order, that window can be made to span two or three classes at
once.  Fresh process: the driver hands physical memory out one class after the other (47 chunks of 2 GiB each,
profiles/r03_arena_probe_fresh.txt), so three groups of small chunks separated by fillers land in the three classes;
the groups' classes are then CHECKED with the pair rates (within a group 5.5, across groups 6.4) before any striped
layout is timed.

    python tools/stripe_probe.py [--shape 256,16,16,128] [--chunk-mib 2]
"""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import _cabi as cabi  # noqa: E402
from datafusion_amd.kernels import HipKernels  # noqa: E402
from datafusion_amd.placement import _Raw  # noqa: E402

GIB = 1 << 30


class Arena:
    def __init__(self, chunk_bytes, n):
        self.lib = cabi.lib()
        self.h = C.c_void_p()
        cabi.check(self.lib.dfx_arena_create(chunk_bytes, n, C.byref(self.h)), "dfx_arena_create")
        cb = C.c_int64()
        cabi.check(self.lib.dfx_arena_info(self.h, C.byref(cb), None, None), "dfx_arena_info")
        self.chunk_bytes, self.n = cb.value, n

    def grow(self, n):
        cabi.check(self.lib.dfx_arena_grow(self.h, n), "dfx_arena_grow")
        ids = list(range(self.n, self.n + n))
        self.n += n
        return ids

    def map(self, ids):
        arr = (C.c_int32 * len(ids))(*ids)
        ptr = C.c_void_p()
        cabi.check(self.lib.dfx_arena_map(self.h, arr, len(ids), C.byref(ptr)), "dfx_arena_map")
        return ptr.value

    def unmap(self, ptr):
        cabi.check(self.lib.dfx_arena_unmap(self.h, C.c_void_p(ptr)), "dfx_arena_unmap")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="256,16,16,128")
    ap.add_argument("--chunk-mib", type=int, default=2)
    ap.add_argument("--filler-gib", type=int, default=0, help="0: fill each class up to 94 GiB")
    a = ap.parse_args()
    n_t, n_l, n_z, xy = map(int, a.shape.split(","))
    dev = "cuda:0"
    torch.cuda.init()
    print("free GiB at start:", torch.cuda.mem_get_info()[0] / GIB, flush=True)
    k = HipKernels(n_t, n_l, n_z, xy, xy, 4, dev)
    k.new_factored_residual()
    k.ra.zero_(), k.rb.zero_()
    k.scal[cabi.S_ALPHA], k.scal[cabi.S_BETA] = 0.0, 0.0
    nbytes = 4 * k.N
    cb = a.chunk_mib << 20
    per = -(-nbytes // cb)                     # small chunks per vector
    group_bytes = 2 * per * cb
    filler = (a.filler_gib * GIB) if a.filler_gib else (94 * GIB - group_bytes)
    n_fill = max(0, filler // (2 * GIB))
    t0 = time.perf_counter()
    small = Arena(cb, 2 * per)
    print(f"chunk {small.chunk_bytes >> 20} MiB, {per} per vector, group {group_bytes / GIB:.1f} GiB, filler {n_fill} x 2 GiB",
          flush=True)
    groups = [list(range(0, 2 * per))]
    fillers = []
    for _ in range(2):
        if n_fill:
            fillers.append(Arena(2 * GIB, n_fill))
        groups.append(small.grow(2 * per))
    print(f"created {small.n} chunks in {time.perf_counter() - t0:.2f} s; free GiB now {torch.cuda.mem_get_info()[0] / GIB:.1f}",
          flush=True)

    def rate(x_ids, p_ids, reps=3, label=""):
        t0 = time.perf_counter()
        xp, pp = small.map(x_ids), small.map(p_ids)
        t_map = time.perf_counter() - t0
        x = torch.as_tensor(_Raw(xp, k.N), device=dev).view(k.shape).zero_()
        p = torch.as_tensor(_Raw(pp, k.N), device=dev).view(k.shape).zero_()
        best = 1e30
        for _ in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            k.sweep(cabi.SWEEP_XP, x, None, p, factored=True)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        del x, p
        small.unmap(xp), small.unmap(pp)
        r = 16.0 * k.N / best / 1e9
        print(f"  {label:58s} {best:8.3f} ms  {r:6.3f} TB/s  ({r / 8:.3f})   [map {t_map:.2f} s]", flush=True)
        return r

    g = groups
    print("## control: whole vectors inside / across groups")
    for i in range(3):
        rate(g[i][:per], g[i][per:], label=f"x, p both in group {i}")
    for i, j in ((0, 1), (0, 2), (1, 2)):
        rate(g[i][:per], g[j][:per], label=f"x in group {i}, p in group {j}")

    def striped(order, stripe, half):
        """A vector whose stripe-sized runs of chunks cycle through the groups in `order`; half 0 / 1 = which half of
        each group's chunks it draws from (x and p must not share chunks)."""
        cur = {gi: (half * per) for gi in set(order)}
        out, i = [], 0
        while len(out) < per:
            gi = order[i % len(order)]
            take = min(stripe, per - len(out))
            out += g[gi][cur[gi]:cur[gi] + take]
            cur[gi] += take
            i += 1
        return out

    print("## striped layouts")
    for stripe_mib in (a.chunk_mib, 4 * a.chunk_mib, 16 * a.chunk_mib, 64 * a.chunk_mib, 256 * a.chunk_mib):
        s = max(1, (stripe_mib << 20) // cb)
        if s * 3 > per:
            continue
        rate(striped((0, 1, 2), s, 0), striped((1, 2, 0), s, 1), label=f"stripe {stripe_mib} MiB: x ABC.., p BCA..")
        rate(striped((0, 1), s, 0), striped((1, 0), s, 1), label=f"stripe {stripe_mib} MiB: x AB.., p BA..")
        rate(striped((0, 1, 2), s, 0), striped((0, 1, 2), s, 1), label=f"stripe {stripe_mib} MiB: x ABC.., p ABC.. (same class)")
    print("## control again")
    rate(g[0][:per], g[1][:per], label="x in group 0, p in group 1")


if __name__ == "__main__":
    main()
