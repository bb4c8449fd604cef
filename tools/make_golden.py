"""Generate tests/golden/*.npz by running the REAL reference (imported read-only from
/root/reference with tools/oracle_shim standing in for the absent torchvision) and, in the same
run, assert that oracle/fusion_oracle.py reproduces it.  Runs in the build container only:

    MPLBACKEND=Agg PYTHONPATH=tools/oracle_shim:/root/reference:. python tools/make_golden.py

Only inputs and outputs (arrays) are written; nothing of the reference travels.
"""
import os
import sys

import numpy as np
import torch

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(HERE, "oracle_shim"), "/root/reference", ROOT]

from datafusion.fusion import FusionAdam, FusionCG  # noqa: E402  (the reference)
from datafusion.baseline import baseline as ref_baseline  # noqa: E402
from oracle import fusion_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
W = {"spatial": 0.5, "spectro_temporal": 0.5}


def uniform(n_t, n_l, n_z, xy, r, seed=0):
    spc, cmos = O.synthetic_inputs(n_t, n_l, n_z, xy, r, seed)
    return spc.numpy().astype(np.float64), cmos.numpy().astype(np.float64)


def run_cg(spc, cmos, iters, **kw):
    """Reference FusionCG with verbose history (spatial, st, total, rsnew per iteration)."""
    import matplotlib.pyplot as plt
    f = FusionCG(spc, cmos, verbose=True, **kw)
    x0 = f.x.clone().numpy()
    hist = {}
    # history is local to __call__; recover it from the scatter plots the reference draws
    x, spc_n, cmos_n = f(iters)
    fig = plt.gcf()
    cols = [ax.collections[0].get_offsets()[:, 1].data for ax in fig.axes]
    plt.close("all")
    hist = np.stack(cols, 1)  # (iters, 4): spatial, st, total, rsnew
    return f, x0, x, spc_n, cmos_n, hist


def case_cg(name, spc, cmos, iters_list, weights=W, check=1e-6, **kw):
    kw = dict(weights=weights, init_type="baseline", tol=None, **kw)
    out = {"spc": spc, "cmos": cmos, "w1": weights["spatial"], "w2": weights["spectro_temporal"]}
    for k in iters_list:
        f, x0, x, spc_n, cmos_n, hist = run_cg(spc, cmos, k, **kw)
        out[f"x_{k}"] = x
        out["x0"], out["spc_n"], out["cmos_n"] = x0, spc_n, cmos_n
        out["hist"] = hist
        out["spc_mask"] = f.spc_mask.numpy()
        out["cmos_mask"] = f.cmos_mask.numpy()
        # oracle must reproduce the reference
        okw = {a: b for a, b in kw.items() if a != "verbose"}
        pb = O.prepare(spc, cmos, **okw)
        assert O.rel_l2(pb.x, x0) <= 1e-7, (name, "x0", O.rel_l2(pb.x, x0))
        assert np.array_equal(pb.spc_mask.numpy(), out["spc_mask"])
        assert np.array_equal(pb.cmos_mask.numpy(), out["cmos_mask"])
        xo = O.cg(pb, k)
        e = O.rel_l2(xo, x)
        print(f"{name}: k={k} oracle-vs-reference rel_l2={e:.3e} bitwise={np.array_equal(xo.numpy(), x)}"
              f" curr_iter={f.curr_iter}")
        assert e <= check, (name, k, e)
        rs = np.array([t["rsnew"] for t in pb.trace])
        assert np.allclose(rs, hist[: len(rs), 3], rtol=1e-5), (rs, hist[:, 3])
        out[f"curr_iter_{k}"] = f.curr_iter
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def scrape_history():
    """The reference keeps `history` local to __call__; with verbose=True it draws one scatter plot per column."""
    import matplotlib.pyplot as plt
    fig = plt.gcf()
    cols = [ax.collections[0].get_offsets()[:, 1].data for ax in fig.axes]
    plt.close("all")
    return np.stack(cols, 1)


def adam_cases():
    """Adam variant: 5 steps, losses of every step and the first gradient - all taken from the REFERENCE objects:
    grad0 is the reference's own x.grad after a one-iteration call (fusion.py:338-341: loss.backward(), then
    _mask_gradients when mask_noise), the loss history is what its verbose run plots."""
    for name, mk, nn, lr in (("adam_t6l4z3_32", False, False, 1e-8), ("adam_t6l4z3_32_mask_nn", True, True, 1e-7)):
        spc, cmos = O.cells_like_inputs(6, 4, 3, 32, 4, seed=9)
        f = FusionAdam(spc, cmos, weights=W, init_type="baseline", tol=None, mask_noise=mk)
        x0 = f.x.clone().numpy()
        l1, l2 = f.loss()
        x, spc_n, cmos_n = f(lr=lr, max_iterations=5, non_neg=nn)
        f1 = FusionAdam(spc, cmos, weights=W, init_type="baseline", tol=None, mask_noise=mk)
        x_1, _, _ = f1(lr=lr, max_iterations=1, non_neg=nn)
        grad0 = f1.x.grad.detach().clone().numpy()                  # the reference's gradient (masked when mask_noise)
        fv = FusionAdam(spc, cmos, weights=W, init_type="baseline", tol=None, mask_noise=mk, verbose=True)
        xv, _, _ = fv(lr=lr, max_iterations=5, non_neg=nn)
        hist = scrape_history()                                      # (5, 3): spatial, spectro-temporal, total
        assert np.array_equal(xv, x) and hist.shape == (5, 3)
        pb = O.prepare(spc, cmos, weights=W, init_type="baseline", tol=None, mask_noise=mk)
        xo = O.adam(pb, lr, 5, non_neg=nn)
        e = O.rel_l2(xo, x)
        g_or = pb.trace[0]["grad"].numpy()
        if mk:
            g_or = g_or * pb.cmos_mask.numpy()[None, None]
        eg = O.rel_l2(g_or, grad0)
        print(f"{name}: oracle-vs-reference rel_l2 x={e:.3e} grad0={eg:.3e}")
        assert e <= 1e-6 and eg <= 1e-6
        assert abs(pb.trace[0]["spatial_loss"] - float(l1)) <= 1e-6 * abs(float(l1))
        lo = np.array([[t["spatial_loss"], t["spectro_temporal_loss"]] for t in pb.trace])
        assert np.allclose(lo, hist[:, :2], rtol=1e-5)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), spc=spc, cmos=cmos, x0=x0, x_1=x_1, x_5=x, lr=lr,
                            loss0=np.array([float(l1), float(l2)]), grad0=grad0, grad0_masked=np.array(mk),
                            losses=hist[:, :2], w1=0.5, w2=0.5)


def cg_verbose_tol_case():
    """verbose=True together with a tolerance: the history rows stop where the loop breaks (fusion.py:451-478: the
    break comes before the row of that iteration is written) and the rest stays zero."""
    spc, cmos = uniform(8, 4, 4, 32, 4, seed=6)
    f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=3e-3, verbose=True)
    x, _, _ = f(50)
    hist = scrape_history()
    pb = O.prepare(spc, cmos, weights=W, init_type="baseline", tol=3e-3)
    xo = O.cg(pb, 50)
    assert pb.curr_iter == f.curr_iter and O.rel_l2(xo, x) <= 1e-6
    assert hist.shape == (50, 4) and np.all(hist[f.curr_iter - 1:] == 0) and np.all(hist[:f.curr_iter - 1, 3] > 0)
    print("verbose + tol case: stop at", f.curr_iter, "rows written", f.curr_iter - 1)
    np.savez_compressed(os.path.join(OUT, "cg_verbose_tol_t8l4z4_32.npz"), spc=spc, cmos=cmos, x=x,
                        curr_iter=f.curr_iter, hist=hist, w1=0.5, w2=0.5)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    only = set(sys.argv[1:])
    if only:                       # regenerate some fixtures only: python tools/make_golden.py adam cg_verbose_tol
        if "adam" in only:
            adam_cases()
        if "cg_verbose_tol" in only:
            cg_verbose_tol_case()
        return

    # --- CG, uniform inputs, several shapes incl. non power of two and r in {2,4,8}
    case_cg("cg_t8l4z4_32_r4", *uniform(8, 4, 4, 32, 4), [1, 5, 20])
    case_cg("cg_t6l5z3_24_r4", *uniform(6, 5, 3, 24, 4, seed=1), [1, 20])
    case_cg("cg_t4l3z2_16_r2", *uniform(4, 3, 2, 16, 2, seed=2), [20])
    case_cg("cg_t4l2z3_32_r8", *uniform(4, 2, 3, 32, 8, seed=3), [20])
    # fast-path geometry (Y % 64 == 0, r = 4), unequal weights as in the notebooks (0.2 / 0.8)
    case_cg("cg_t3l2z5_64_r4_w28", *uniform(3, 2, 5, 64, 4, seed=4), [10],
            weights={"spatial": 0.2, "spectro_temporal": 0.8})
    # cells-like data, masked variant (beads notebook uses mask_noise=True), total_energy != 1
    spc, cmos = O.cells_like_inputs(8, 4, 5, 32, 4, seed=5)
    case_cg("cg_cells_t8l4z5_32_mask", spc, cmos, [10], mask_noise=True)
    case_cg("cg_cells_t8l4z5_32_e7", spc, cmos, [10], total_energy=7.0)

    # --- tolerance early break (README default tol=3e-3)
    spc, cmos = uniform(8, 4, 4, 32, 4, seed=6)
    f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=3e-3)
    x, _, _ = f(50)
    pb = O.prepare(spc, cmos, weights=W, init_type="baseline", tol=3e-3)
    xo = O.cg(pb, 50)
    assert pb.curr_iter == f.curr_iter and O.rel_l2(xo, x) <= 1e-6
    tols = np.array([t["tolerance"] for t in pb.trace])
    print("tol case: stop at", f.curr_iter, "tolerances", tols)
    np.savez_compressed(os.path.join(OUT, "cg_tol_t8l4z4_32.npz"), spc=spc, cmos=cmos, x=x,
                        curr_iter=f.curr_iter, tolerances=tols, w1=0.5, w2=0.5)

    # --- random init (CPU generator, seed 42) and zeros init (diverges by construction)
    spc, cmos = uniform(4, 3, 3, 16, 4, seed=7)
    f = FusionCG(spc, cmos, weights=W, init_type="random", tol=None, seed=42)
    x0 = f.x.clone().numpy()
    x, _, _ = f(5)
    pb = O.prepare(spc, cmos, weights=W, init_type="random", tol=None, seed=42)
    assert np.array_equal(pb.x.numpy(), x0)
    assert O.rel_l2(O.cg(pb, 5), x) <= 1e-6
    np.savez_compressed(os.path.join(OUT, "cg_random_t4l3z3_16.npz"), spc=spc, cmos=cmos, x0=x0, x_5=x,
                        w1=0.5, w2=0.5)
    f = FusionCG(spc, cmos, weights=W, init_type="zeros", tol=None)
    try:
        f(3)
        raise SystemExit("zeros init did not diverge?")
    except AssertionError as e:
        assert "diverged" in str(e)
        print("zeros init: AssertionError as expected:", e)

    # --- baseline() called directly, as the notebooks do
    spc, cmos = uniform(5, 3, 4, 24, 4, seed=8)
    xb = ref_baseline(cmos, spc, device="cpu", return_numpy=True)
    xo = O.baseline_init(torch.from_numpy(cmos.astype(np.float32)), torch.from_numpy(spc.astype(np.float32)))
    assert O.rel_l2(xo, xb) <= 1e-7
    np.savez_compressed(os.path.join(OUT, "baseline_t5l3z4_24.npz"), spc=spc, cmos=cmos, x=xb)

    adam_cases()
    cg_verbose_tol_case()
    sizes = {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))}
    print(sizes, sum(sizes.values()))


if __name__ == "__main__":
    main()
