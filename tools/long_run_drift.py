"""How do the fp32 solvers drift from an fp64 solve over many iterations?  Factored and explicit residual (HIP) and
the eager torch-CPU oracle (= the reference's arithmetic) against the fp64 collapsed CG, same inputs.

    python tools/long_run_drift.py [--shape 32,16,8,64] [--iters 50,150,300]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import FusionCG  # noqa: E402
from oracle import fusion_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="32,16,8,64")
    ap.add_argument("--iters", default="50,150,300")
    ap.add_argument("--cells", action="store_true")
    a = ap.parse_args()
    n_t, n_l, n_z, xy = map(int, a.shape.split(","))
    w = {"spatial": 0.5, "spectro_temporal": 0.5}
    gen = O.cells_like_inputs if a.cells else O.synthetic_inputs
    spc, cmos = gen(n_t, n_l, n_z, xy, 4, seed=3)
    out = {"shape": [n_t, n_l, n_z, xy, xy], "inputs": "cells-like" if a.cells else "uniform"}
    for k in (int(v) for v in a.iters.split(",")):
        f = FusionCG(spc, cmos, weights=w, init_type="baseline", tol=None, device="cuda:0")
        x0 = f.x.clone()
        f._project(x0)
        u0, v0 = f.k.u.cpu().clone(), f.k.v.cpu().clone()
        ca, cg, _, rs = O.collapsed_cg(f.spc.cpu(), f.cmos.cpu(), u0, v0, f.x_shape, 4, 0.5, 0.5, k)
        ref = O.expand_small(x0.cpu(), ca, cg, 4).double()
        row = {"rsnew_fp64": rs[-1]}
        for residual in ("factored", "explicit"):
            g = FusionCG(spc, cmos, weights=w, init_type="baseline", tol=None, device="cuda:0", residual=residual)
            x, _, _ = g(k, return_numpy=False)
            row[residual] = O.rel_l2(x.cpu(), ref)
            row[f"rsnew_{residual}"] = float(g.k.scal[1])
            if residual == "factored":   # (||bcast ra||^2 + ||bcast rb||^2) / ||r||^2: 1..2 when nothing cancels
                sc = g.k.scal.cpu()
                row["factor_norms_over_rr"] = float((g.k.M * sc[21] + n_z * 16 * sc[22]) / sc[11])
        pb = O.prepare(spc, cmos, weights=w, init_type="baseline", tol=None)
        row["oracle_fp32_cpu"] = O.rel_l2(O.cg(pb, k, with_loss=False), ref)
        # distance moved from x0, for scale
        row["moved"] = O.rel_l2(x0.cpu(), ref)
        out[str(k)] = row
        print(k, row, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
