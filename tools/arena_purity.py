"""Are the physical chunks the driver hands out PURE (one memory class each)?  For every 2 GiB chunk of an arena over
(almost) all free memory: the probe on its two halves (a pure chunk: the slow, same-class rate) and on its quarters,
then its class.  Run fresh, and after the process has allocated, touched and freed the memory once (--dirty [--wait S]).

    python tools/arena_purity.py [--dirty] [--wait 0]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import _cabi as cabi, placement  # noqa: E402
from datafusion_amd.kernels import HipKernels  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dirty", action="store_true")
    ap.add_argument("--wait", type=float, default=0.0)
    ap.add_argument("--chunks", type=int, default=0)
    a = ap.parse_args()
    dev = "cuda:0"
    pool = placement.pool(dev)
    if a.dirty:
        big = torch.empty(int(torch.cuda.mem_get_info()[0] * 0.95) // 4, device=dev)
        big[::1 << 20] = 1.0
        torch.cuda.synchronize()
        del big
        torch.cuda.empty_cache()
        time.sleep(a.wait)
    t0 = time.perf_counter()
    n = a.chunks or pool._allocatable_chunks()
    ids = pool._create(n)
    t1 = time.perf_counter()
    base = pool._map(ids)
    CB = placement.CHUNK_BYTES
    for i in range(n):
        pool._zero(base + i * CB, CB // 4)
    torch.cuda.synchronize()
    print(f"{n} chunks created in {t1 - t0:.2f} s, mapped + zeroed in {time.perf_counter() - t1:.2f} s", flush=True)
    # quarter-chunk probe kernels (512 MiB per operand)
    kq = HipKernels(CB // 4 // (32 << 20), 32, 4, 256, 256, 4, dev)
    kq.new_factored_residual()
    kq.ra.zero_(), kq.rb.zero_()

    def rate_q(xp, pp):
        x = torch.as_tensor(placement._Raw(xp, kq.N), device=dev).view(kq.shape)
        p = torch.as_tensor(placement._Raw(pp, kq.N), device=dev).view(kq.shape)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            kq.sweep(cabi.SWEEP_XP, x, None, p, factored=True)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return 16.0 * kq.N / best / 1e6

    half, q = CB // 2, CB // 4
    ref = base
    print("chunk: halves  q0-q1 q2-q3 q0-q2 | vs chunk 0 (half probes: first half, second half)")
    for i in range(n):
        c = base + i * CB
        print(f"{i:4d}: {pool._rate(c, c + half):6.0f}  {rate_q(c, c + q):5.0f} {rate_q(c + 2 * q, c + 3 * q):5.0f} "
              f"{rate_q(c, c + 2 * q):5.0f} | {pool._rate(ref, c) if i else 0:6.0f} {pool._rate(ref, c + half) if i else 0:6.0f}", flush=True)
    pool._unmap(base)
    pool._destroy_arena()


if __name__ == "__main__":
    main()
