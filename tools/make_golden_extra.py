"""Supplementary goldens (same rules as tools/make_golden.py: the REAL reference is imported read-only from
/root/reference in the build container, the oracle must reproduce it, only arrays are written):

  adam_tol_t6l4z3_32   FusionAdam leaving its loop through the tolerance test (fusion.py:348-358), non_neg=True

    MPLBACKEND=Agg PYTHONPATH=tools/oracle_shim:/root/reference:. python tools/make_golden_extra.py
"""
import os
import sys

import numpy as np
import torch

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(HERE, "oracle_shim"), "/root/reference", ROOT]

from datafusion.fusion import FusionAdam  # noqa: E402  (the reference)
from oracle import fusion_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
W = {"spatial": 0.5, "spectro_temporal": 0.5}


def main():
    torch.set_num_threads(4)
    spc, cmos = O.cells_like_inputs(6, 4, 3, 32, 4, seed=10)
    lr = 1e-7
    # pick a tolerance the run crosses between its 6th and 7th iteration
    pb = O.prepare(spc, cmos, weights=W, init_type="baseline", tol=1e-30)
    O.adam(pb, lr, 12, non_neg=True)
    tols = [t["tolerance"] for t in pb.trace]
    print("tolerances without a break:", tols)
    tol = float(np.sqrt(tols[5] * tols[6])) if tols[6] < tols[5] else float(0.5 * (tols[5] + tols[6]))
    k_stop = next(i + 1 for i, t in enumerate(tols) if t < tol)
    f = FusionAdam(spc, cmos, weights=W, init_type="baseline", tol=tol)
    x, _, _ = f(lr=lr, max_iterations=12, non_neg=True)
    pb = O.prepare(spc, cmos, weights=W, init_type="baseline", tol=tol)
    xo = O.adam(pb, lr, 12, non_neg=True)
    print(f"tol={tol:.6e}: reference stops at {f.curr_iter}, oracle at {pb.curr_iter} (expected {k_stop}); "
          f"rel_l2={O.rel_l2(xo, x):.3e} bitwise={np.array_equal(xo.numpy(), x)}")
    assert f.curr_iter == pb.curr_iter == k_stop and O.rel_l2(xo, x) <= 1e-6
    np.savez_compressed(os.path.join(OUT, "adam_tol_t6l4z3_32.npz"), spc=spc, cmos=cmos, x=x, lr=lr, tol=tol,
                        curr_iter=f.curr_iter, tolerances=np.array([t["tolerance"] for t in pb.trace]), w1=0.5, w2=0.5)
    print("wrote adam_tol_t6l4z3_32.npz", os.path.getsize(os.path.join(OUT, "adam_tol_t6l4z3_32.npz")))


if __name__ == "__main__":
    main()
