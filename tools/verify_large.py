"""Numerical check at a volume no CPU restatement fits: the fp32 HIP CG (5-D x and p) against the fp64 collapsed
(small-space) CG of oracle/fusion_oracle.py.  Default: t512 x l32 x z16 x 256 x 256
(2^34 elements, 64 GiB per vector) on one MI355X; with the factored residual BASELINE config 3 itself
(--shape 512,32,32,256: 2^35 elements, 128 GiB per vector, x + p = 256 GiB) fits one GPU.

    python tools/verify_large.py [--shape 512,32,16,256] [--iters 10] [--residual factored|explicit]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import FusionCG  # noqa: E402
from oracle import fusion_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="512,32,16,256")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--residual", default="factored", choices=["factored", "explicit"])
    ap.add_argument("--cells", action="store_true", help="cells-like inputs (blobs x decays x spectra) instead of uniform")
    ap.add_argument("--r", type=int, default=4, help="spatial_increase (other lane maps: 16 row-part tile, 3 flat map for any r ...)")
    a = ap.parse_args()
    n_t, n_l, n_z, xy = map(int, a.shape.split(","))
    r, dev = a.r, "cuda:0"
    w = {"spatial": 0.5, "spectro_temporal": 0.5}
    spc, cmos = (O.cells_like_inputs if a.cells else O.synthetic_inputs)(n_t, n_l, n_z, xy, r)
    t0 = time.time()
    f = FusionCG(spc, cmos, weights=w, init_type="baseline", tol=None, device=dev, residual=a.residual)
    f._project(f.x)
    u0, v0 = f.k.u.cpu().clone(), f.k.v.cpu().clone()
    sum_x0 = float(f.k.scal[15])
    f(a.iters, return_numpy=False)
    torch.cuda.synchronize()
    t_gpu = time.time() - t0
    rs_gpu = float(f.k.scal[1])
    print(f"GPU: {a.iters} iterations on N = {f.k.N} in {t_gpu:.1f}s (incl. set-up); rsnew = {rs_gpu:.6e}", flush=True)
    t0 = time.time()
    ca, cg, alphas, rs = O.collapsed_cg(f.spc.cpu(), f.cmos.cpu(), u0, v0, f.x_shape, r, 0.5, 0.5, a.iters)
    print(f"fp64 collapsed CG on the host: {time.time() - t0:.1f}s; rsnew = {rs[-1]:.6e}", flush=True)
    # regenerate x0 (deterministic) into fresh memory after dropping r and p
    x = f.x
    f._r = f._p = None
    torch.cuda.empty_cache()
    x0 = f._initialize()
    ca, cg = ca.to(dev), cg.to(dev)
    num = torch.zeros((), dtype=torch.float64, device=dev)
    den = torch.zeros((), dtype=torch.float64, device=dev)
    for t in range(n_t):
        up = cg[t].repeat_interleave(r, 1).repeat_interleave(r, 2)
        ref = x0[t].double() + ca[None] + up[:, None]
        num += ((x[t].double() - ref) ** 2).sum()
        den += (ref ** 2).sum()
    err = float(torch.sqrt(num / den))
    out = {"shape": [n_t, n_l, n_z, xy, xy], "r": r, "fast": int(f.k.plan.fast), "v_layers": int(f.k.v_layers), "n_elements": f.k.N, "iterations": a.iters, "residual": a.residual,
           "inputs": "cells-like" if a.cells else "uniform",
           "rel_l2_vs_fp64_collapsed": err, "rsnew_gpu": rs_gpu, "rsnew_fp64": rs[-1], "alpha_last_fp64": alphas[-1],
           "alpha_last_gpu": float(f.k.scal[24])}
    print(json.dumps(out), flush=True)
    assert err <= 1e-4


if __name__ == "__main__":
    main()
