"""GPU experiment for datafusion_amd/placement.py: does the HIP virtual-memory API work here, how cleanly do the
chunk classes separate at different probe sizes, and what do vectors laid out by class gain?

    python tools/arena_probe.py [--whole]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import _cabi as cabi, placement  # noqa: E402
from datafusion_amd.kernels import HipKernels  # noqa: E402


def t_sweep(k, x, r, p, reps=4, factored=True):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        k.sweep(cabi.SWEEP_XP, x, r, p, factored=factored)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--whole", action="store_true", help="also the 128 GiB-per-vector volume")
    ap.add_argument("--dirty", action="store_true", help="first allocate, touch and free (almost) all memory")
    a = ap.parse_args()
    dev = "cuda:0"
    pool = placement.pool(dev)
    print("free GiB at start:", torch.cuda.mem_get_info()[0] / 2**30, flush=True)
    if a.dirty:
        # use the whole memory once, so that what follows is handed memory with a history (the driver clears it on
        # allocation, ~21 ms per GiB, and hands it out in a scattered order)
        t0 = time.perf_counter()
        big = torch.empty(int(torch.cuda.mem_get_info()[0] * 0.95) // 4, device=dev)
        big[::1 << 20] = 1.0
        torch.cuda.synchronize()
        del big
        torch.cuda.empty_cache()
        print(f"dirtied the device memory in {time.perf_counter() - t0:.2f} s", flush=True)

    def pair_case(shape, label, n_vec=2):
        k = HipKernels(*shape, 4, dev)
        k.new_factored_residual()
        k.ra.zero_(), k.rb.zero_()
        nb = 4 * k.N
        words = 4 if n_vec == 2 else 5
        # rank-aware
        vecs, info = pool.alloc_vectors(n_vec, k.N)
        vs = [v.view(k.shape).zero_() for v in vecs]
        ms = t_sweep(k, vs[0], None, vs[1]) if n_vec == 2 else t_sweep(k, vs[0], vs[2], vs[1], factored=False)
        print(f"{label}: rank-aware {n_vec} x {nb / 2**30:.0f} GiB: sweep {ms:.3f} ms = {words * nb / ms / 1e9:.3f} TB/s "
              f"({words * nb / ms / 8e9:.3f} of peak); set-up {info['seconds']:.2f} s, chunks created {info['chunks_created']}, "
              f"probes {info['probes']}, conflicts {info['positions_sharing_a_class']}, classes {info['class_counts']}", flush=True)
        for lg in pool.log:
            if "class_sequence" in lg:
                print(f"      batch of {lg['new_chunks']}: {lg['probes']} probes, same-class rate {lg['same_class_rate_GBps']:.0f}, "
                      f"probe rates {lg['probe_rates_GBps']}, {lg['class_sequence']}", flush=True)
        pool.log.clear()
        del vecs, vs
        # a second request of the same size reuses the classified chunks
        t0 = time.perf_counter()
        vecs, info = pool.alloc_vectors(n_vec, k.N)
        print(f"   second request: {time.perf_counter() - t0:.3f} s, probes {info['probes']}", flush=True)
        del vecs
        pool.trim()
        # plain torch allocations
        if n_vec * nb < 200 * 2**30:
            vs = [torch.zeros(k.shape, device=dev) for _ in range(n_vec)]
            ms = t_sweep(k, vs[0], None, vs[1]) if n_vec == 2 else t_sweep(k, vs[0], vs[2], vs[1], factored=False)
            print(f"{label}: torch.zeros {n_vec} x {nb / 2**30:.0f} GiB: sweep {ms:.3f} ms = {words * nb / ms / 1e9:.3f} TB/s",
                  flush=True)
            del vs
        torch.cuda.empty_cache()

    pair_case((512, 32, 4, 256, 256), "c3g8 slab")
    pair_case((512, 32, 4, 256, 256), "c3g8 slab, explicit residual", n_vec=3)
    pair_case((256, 16, 16, 128, 128), "C2")
    if a.whole:
        pair_case((512, 32, 32, 256, 256), "config 3 whole")


if __name__ == "__main__":
    main()
