"""The reference's own timing protocol (analysis/runtime.py:78-115) on this build: wall time of ONE call
`fusion(max_iterations=...)` as a user makes it - set-up excluded as there, the device-to-host copy of the
result (return_numpy=True, fusion.py:489-493) INCLUDED - for FusionCG (10 iterations) and FusionAdam
(100 iterations, lr 1e-8), tol = 3e-3, on a cells-like acquisition (the Zenodo file is not available offline).
Also prints the PCIe-inclusive rates DESIGN.md quotes next to bench.py's HBM-resident `value`.

    python tools/runtime_like_reference.py [--shape 64,16,20,128] [--runs 5] [--cpu]
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import FusionAdam, FusionCG  # noqa: E402
from oracle import fusion_oracle as O  # noqa: E402

W = {"spatial": 0.5, "spectro_temporal": 0.5}


def timed_call(make, call, runs):
    ts, iters, ctor = [], None, []
    for _ in range(runs):
        t0 = time.perf_counter()
        f = make()
        torch.cuda.synchronize()
        ctor.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        out = call(f)
        ts.append(time.perf_counter() - t0)
        iters = f.curr_iter
        del f, out
        torch.cuda.empty_cache()
        gc.collect()
    return {"mean_s": float(np.mean(ts)), "std_s": float(np.std(ts)), "min_s": float(np.min(ts)), "iterations": iters,
            "constructor_mean_s": float(np.mean(ctor[1:] or ctor))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64,16,20,128")
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--tol", default="3e-3", help="tolerance of the stopping rule, or 'none'")
    ap.add_argument("--cpu", action="store_true", help="also time the CPU oracle (torch-CPU port) on the same call")
    a = ap.parse_args()
    n_t, n_l, n_z, xy = map(int, a.shape.split(","))
    spc, cmos = O.cells_like_inputs(n_t, n_l, n_z, xy, 4, seed=0)
    tol = None if a.tol.lower() == "none" else float(a.tol)
    kw = dict(weights=W, init_type="baseline", mask_noise=False, tol=tol, total_energy=1, device="cuda", seed=42,
              verbose=False)
    res = {"shape": [n_t, n_l, n_z, xy, xy], "n_elements": n_t * n_l * n_z * xy * xy, "tol": tol}
    res["cg_numpy"] = timed_call(lambda: FusionCG(spc, cmos, **kw), lambda f: f(max_iterations=10), a.runs)
    res["cg_device"] = timed_call(lambda: FusionCG(spc, cmos, **kw),
                                  lambda f: (f(max_iterations=10, return_numpy=False), torch.cuda.synchronize()), a.runs)
    res["adam_numpy"] = timed_call(lambda: FusionAdam(spc, cmos, **kw), lambda f: f(max_iterations=100, lr=1e-8), a.runs)
    res["adam_device"] = timed_call(lambda: FusionAdam(spc, cmos, **kw),
                                    lambda f: (f(max_iterations=100, lr=1e-8, return_numpy=False),
                                               torch.cuda.synchronize()), a.runs)
    if a.cpu:
        t0 = time.perf_counter()
        pb = O.prepare(spc, cmos, weights=W, init_type="baseline", tol=tol)
        t1 = time.perf_counter()
        O.cg(pb, 10)
        t2 = time.perf_counter()
        res["cg_cpu_oracle"] = {"call_s": t2 - t1, "constructor_s": t1 - t0, "iterations": pb.curr_iter,
                                "threads": torch.get_num_threads()}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
