"""Kernel tuning harness (GPU box): times dfx_sweep(XP) and dfx_bcast on one geometry over
build variants (DFX_LIB=<variant .so>) and plan overrides.  Interleaved rounds in one process per
library; prints one line per configuration.

    DFX_LIB=build/libdfx_nt1_mw1.so python tools/tune_kernels.py --zc 8,4 --steps 32,64
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import _cabi as cabi  # noqa: E402
from datafusion_amd.kernels import HipKernels  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="256,16,16,128,128")
    ap.add_argument("--zc", default="0")
    ap.add_argument("--steps", default="0")
    ap.add_argument("--blocks", default="0", help="target block counts (0 = one per CU)")
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--pad", default="-1", help="bytes of skew between x, r, p carved from one buffer (-1: 3 allocations)")
    a = ap.parse_args()
    n_t, n_l, n_z, X, Y = map(int, a.shape.split(","))
    dev = "cuda:0"
    pads = [int(v) for v in a.pad.split(",")]
    arena = None
    if os.environ.get("TUNE_ARENA"):
        arena = torch.empty(3 * n_t * n_l * n_z * X * Y + 2 * (max(pads) // 4) + 1024, device=dev)
    for pad in pads:
        run(a, pad, n_t, n_l, n_z, X, Y, dev, arena)


def run(a, pad, n_t, n_l, n_z, X, Y, dev, arena=None):
    n = n_t * n_l * n_z * X * Y
    if arena is not None:
        vec = [arena[i * (n + pad // 4): i * (n + pad // 4) + n].view(n_t, n_l, n_z, X, Y) for i in range(3)]
        for v in vec:
            v.uniform_()
    elif pad < 0:
        vec = [torch.rand((n_t, n_l, n_z, X, Y), device=dev) for _ in range(3)]
    else:
        buf = torch.empty(3 * n + 2 * (pad // 4), device=dev)
        vec = [buf[i * (n + pad // 4): i * (n + pad // 4) + n].view(n_t, n_l, n_z, X, Y) for i in range(3)]
        for v in vec:
            v.uniform_()
    print("pad", pad, "base addresses", [hex(v.data_ptr()) for v in vec], flush=True)
    cfgs = [(int(z), int(s), int(b)) for z in a.zc.split(",") for s in a.steps.split(",") for b in a.blocks.split(",")]
    ks = [HipKernels(n_t, n_l, n_z, X, Y, 4, dev, b, z, s) for z, s, b in cfgs]
    for k in ks:
        k.scal[cabi.S_ALPHA] = 1e-3
        k.scal[cabi.S_BETA] = 0.5
    for k in ks:
        k.new_factored_residual()
        k.ra.zero_(), k.rb.zero_()
    m1a, m1b = torch.zeros_like(ks[0].u), torch.zeros_like(ks[0].v)
    res = {i: {"sweep": [], "bcast": [], "proj": [], "adam": [], "sweepf": [], "adamf": []} for i in range(len(ks))}
    for rnd in range(a.rounds + 1):
        for i, k in enumerate(ks):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
            e[0].record()
            k.sweep(cabi.SWEEP_XP, vec[0], vec[1], vec[2])
            e[1].record()
            k.bcast(vec[1], vec[1], k.u, k.v, None, 1e-3, 1e-3, -1.0, cabi.S_ALPHA)
            e[2].record()
            k.sweep(cabi.SWEEP_PROJECT, vec[0])
            e[3].record()
            k.adam_step(vec[0], vec[1], vec[2], k.u, k.v, None, 1e-8, 0.9, 0.999, 1e-8, 3, False)
            e[4].record()
            k.sweep(cabi.SWEEP_XP, vec[0], None, vec[2], factored=True)
            e[5].record()
            k.adam_step(vec[0], None, vec[2], k.u, k.v, None, 1e-8, 0.9, 0.999, 1e-8, 3, False, m1a=m1a, m1b=m1b)
            e[6].record()
            torch.cuda.synchronize()
            if rnd:
                res[i]["sweepf"].append(e[4].elapsed_time(e[5]))
                res[i]["adamf"].append(e[5].elapsed_time(e[6]))
                res[i]["proj"].append(e[2].elapsed_time(e[3]))
                res[i]["adam"].append(e[3].elapsed_time(e[4]))
                res[i]["sweep"].append(e[0].elapsed_time(e[1]))
                res[i]["bcast"].append(e[1].elapsed_time(e[2]))
    n = vec[0].numel()
    lib = os.path.basename(cabi.LIB_PATH)
    for i, k in enumerate(ks):
        sw, bc = sorted(res[i]["sweep"]), sorted(res[i]["bcast"])
        ms, mb = sw[len(sw) // 2], bc[len(bc) // 2]
        if os.environ.get("TUNE_VERBOSE"):
            print("   sweep rounds:", " ".join(f"{t:.2f}" for t in res[i]["sweep"]), flush=True)
        print(f"{lib} zc={k.plan.zc} steps={k.plan.steps} cpb={k.plan.cpb} blocks={k.plan.n_blocks}: sweep(+reduce_u) med {ms:.3f} ms "
              f"min {sw[0]:.3f} -> {20 * n / ms / 1e6:.0f} GB/s | bcast med {mb:.3f} min {bc[0]:.3f} -> "
              f"{8 * n / mb / 1e6:.0f} GB/s | iter {ms + mb:.3f} ms | project med {sorted(res[i]['proj'])[len(bc) // 2]:.3f} | adam med {sorted(res[i]['adam'])[len(bc) // 2]:.3f} | factored: sweep med "
              f"{sorted(res[i]['sweepf'])[len(bc) // 2]:.3f} adam med {sorted(res[i]['adamf'])[len(bc) // 2]:.3f}", flush=True)


if __name__ == "__main__":
    main()
