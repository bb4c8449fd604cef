"""Where do the HIP path and the eager torch-CPU oracle part at BASELINE configs[1] size?  x0, alpha, rsnew and x after
each of the first iterations: oracle (fp32 eager, the reference's op sequence) vs HIP vs the float64 collapsed CG.

    python tools/diag_c2_oracle.py [--iters 2] [--shape 256,16,16,128,4]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import FusionCG, _cabi as cabi  # noqa: E402
from oracle import fusion_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--shape", default="256,16,16,128,4")
    a = ap.parse_args()
    shape = tuple(int(v) for v in a.shape.split(","))
    W = {"spatial": 0.5, "spectro_temporal": 0.5}
    spc, cmos = O.synthetic_inputs(*shape)
    pb = O.prepare(spc, cmos, weights=W, init_type="baseline", tol=None)
    x0_cpu = pb.x.clone()
    f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device="cuda:0", defer_x=False)
    x0 = f.x.clone()
    print(f"x0: rel-L2 HIP vs oracle {O.rel_l2(x0.cpu(), x0_cpu):.3e}; sum x0 oracle {float(x0_cpu.double().sum()):.9f} "
          f"HIP {float(x0.double().sum()):.9f}", flush=True)
    f._project(x0)
    u0, v0 = f.k.u.cpu().clone(), f.k.v.cpu().clone()
    aa, gg, alphas, rs = O.collapsed_cg(f.spc.cpu(), f.cmos.cpu(), u0, v0, f.x_shape, shape[4], 0.5, 0.5, a.iters)
    f.start()
    for it in range(a.iters):
        f.iterate(1)
        s = f.k.scal.cpu()
        O.cg(pb, 1, with_loss=False) if it == 0 else None
    # the oracle restarts r, p on every call: run it once for all iterations instead
    pb2 = O.prepare(spc, cmos, weights=W, init_type="baseline", tol=None)
    xo = O.cg(pb2, a.iters, with_loss=False)
    f2 = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device="cuda:0", defer_x=False)
    f2.start()
    for it in range(a.iters):
        f2.iterate(1)
        s = f2.k.scal.cpu()
        t = pb2.trace[it]
        print(f"iteration {it + 1}: alpha  oracle {t['alpha']:.9e}  HIP {float(s[cabi.S_ALPHA_LAST]):.9e}  fp64 {alphas[it]:.9e}\n"
              f"             rsnew  oracle {t['rsnew']:.9e}  HIP {float(s[cabi.S_RSNEW]):.9e}  fp64 {rs[it]:.9e}", flush=True)
    f2.finish()
    x = f2.x
    xc = O.expand_small(x0.cpu(), aa, gg, shape[4]) if x0.numel() <= (1 << 28) else None
    print(f"x after {a.iters}: rel-L2 HIP vs oracle {O.rel_l2(x.cpu(), xo):.3e}", flush=True)
    if xc is not None:
        print(f"   HIP vs fp64 collapsed {O.rel_l2(x.cpu(), xc):.3e}; oracle vs fp64 collapsed {O.rel_l2(xo, xc):.3e}")
    else:
        # one time bin at a time, in float64 on the host
        r = shape[4]
        num_h = den = num_o = 0.0
        xh = x.cpu()
        for t in range(xh.shape[0]):
            up = gg[t].repeat_interleave(r, 1).repeat_interleave(r, 2)
            ref = x0[t].cpu().double() + aa[None] + up[:, None]
            num_h += float(((xh[t].double() - ref) ** 2).sum())
            num_o += float(((xo[t].double() - ref) ** 2).sum())
            den += float((ref ** 2).sum())
        print(f"   HIP vs fp64 collapsed {(num_h / den) ** 0.5:.3e}; oracle vs fp64 collapsed {(num_o / den) ** 0.5:.3e}")
    # the step itself
    dx_o, dx_h = xo - x0_cpu, x.cpu() - x0.cpu()
    print(f"   ||x - x0|| / ||x0||: oracle {float(dx_o.double().norm() / x0_cpu.double().norm()):.4e} "
          f"HIP {float(dx_h.double().norm() / x0_cpu.double().norm()):.4e}; rel-L2 of the steps {O.rel_l2(dx_h, dx_o):.3e}")


if __name__ == "__main__":
    main()
