"""Extreme aspect ratios through the solvers against the CPU oracle: very many slabs of tiny planes, one huge plane, very
many z planes, one slab, planes narrower than a wave tile.  python tools/extreme_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import FusionAdam, FusionCG  # noqa: E402
from oracle import fusion_oracle as O  # noqa: E402

W = {"spatial": 0.4, "spectro_temporal": 0.6}
SHAPES = [(2048, 64, 1, 8, 4), (4096, 16, 2, 4, 4), (1, 1, 1, 1024, 4), (1, 1, 3, 2048, 8), (3, 1, 200, 16, 4),
          (2, 2, 129, 8, 2), (1, 7, 5, 512, 2), (5, 3, 2, 768, 3), (700, 3, 3, 24, 6), (2, 1, 40, 80, 5), (9, 9, 9, 36, 9)]
worst = 0.0
for n_t, n_l, n_z, xy, r in SHAPES:
    spc, cmos = O.synthetic_inputs(n_t, n_l, n_z, xy, r, seed=n_z)
    xo = O.cg(O.prepare(spc, cmos, weights=W, init_type="baseline", tol=None), 4, with_loss=False)
    errs = {}
    for res in ("factored", "explicit"):
        f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device="cuda:0", residual=res)
        x, _, _ = f(4)
        errs[res] = O.rel_l2(x, xo)
    lr = 1e-2 / (n_t * n_l * n_z * xy * xy)
    xa = O.adam(O.prepare(spc, cmos, weights=W, init_type="baseline", tol=None), lr, 2)
    for kw in (dict(), dict(fused=False)):
        g = FusionAdam(spc, cmos, weights=W, init_type="baseline", tol=None, device="cuda:0", **kw)
        x, _, _ = g(lr=lr, max_iterations=2)
        errs["adam" + ("" if not kw else "/autograd")] = O.rel_l2(x, xa)
    pl = f.k.plan
    print(f"t{n_t} l{n_l} z{n_z} {xy}x{xy} r{r}: fast={pl.fast} zc={pl.zc}/{pl.zs} n_zchunks={pl.n_zchunks} blocks={pl.n_blocks} "
          f"cpb={pl.cpb} " + " ".join(f"{k} {v:.2e}" for k, v in errs.items()), flush=True)
    worst = max(worst, max(errs.values()))
print(f"worst {worst:.3e}")
assert worst <= 2e-5
