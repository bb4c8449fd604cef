"""CPU suite (-m "not gpu"): oracle vs the reference-generated goldens, the collapsed fp64 oracle,
the C ABI surface, the launch planner and the host logic of FusionCG/FusionAdam driven through a
CPU stand-in backend (single process here; gloo world_size 2 in test_cpu_gloo.py)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import fusion_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def W(g):
    return {"spatial": float(g["w1"]), "spectro_temporal": float(g["w2"])}


CG_CASES = [
    ("cg_t8l4z4_32_r4", [1, 5, 20], {}),
    ("cg_t6l5z3_24_r4", [1, 20], {}),
    ("cg_t4l3z2_16_r2", [20], {}),
    ("cg_t4l2z3_32_r8", [20], {}),
    ("cg_t3l2z5_64_r4_w28", [10], {}),
    ("cg_cells_t8l4z5_32_mask", [10], {"mask_noise": True}),
    ("cg_cells_t8l4z5_32_e7", [10], {"total_energy": 7.0}),
]


# ---------------------------------------------------------------- oracle pinned by the goldens
@pytest.mark.parametrize("name,iters,kw", CG_CASES)
def test_oracle_reproduces_reference_cg(golden, name, iters, kw):
    g = golden(name)
    for k in iters:
        pb = O.prepare(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=None, **kw)
        assert O.rel_l2(pb.x, g["x0"]) <= 1e-7
        assert np.array_equal(pb.spc_mask.numpy(), g["spc_mask"])
        assert np.array_equal(pb.cmos_mask.numpy(), g["cmos_mask"])
        x = O.cg(pb, k)
        assert O.rel_l2(x, g[f"x_{k}"]) <= 1e-6
    rs = np.array([t["rsnew"] for t in pb.trace])
    assert np.allclose(rs, g["hist"][: len(rs), 3], rtol=1e-4)


def test_oracle_tolerance_random_zeros_baseline_adam(golden):
    g = golden("cg_tol_t8l4z4_32")
    pb = O.prepare(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=3e-3)
    x = O.cg(pb, 50)
    assert pb.curr_iter == int(g["curr_iter"]) and O.rel_l2(x, g["x"]) <= 1e-6
    g = golden("cg_random_t4l3z3_16")
    pb = O.prepare(g["spc"], g["cmos"], weights=W(g), init_type="random", tol=None, seed=42)
    assert O.rel_l2(pb.x, g["x0"]) <= 1e-7
    assert O.rel_l2(O.cg(pb, 5), g["x_5"]) <= 1e-6
    pb = O.prepare(g["spc"], g["cmos"], weights=W(g), init_type="zeros", tol=None)
    with pytest.raises(AssertionError, match="diverged"):
        O.cg(pb, 3)
    with pytest.raises(ValueError):
        O.prepare(g["spc"], g["cmos"], weights=W(g), init_type="nope")
    g = golden("baseline_t5l3z4_24")
    xb = O.baseline_init(torch.from_numpy(g["cmos"].astype(np.float32)), torch.from_numpy(g["spc"].astype(np.float32)))
    assert O.rel_l2(xb, g["x"]) <= 1e-7
    for name, mask, nn in (("adam_t6l4z3_32", False, False), ("adam_t6l4z3_32_mask_nn", True, True)):
        g = golden(name)
        pb = O.prepare(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=None, mask_noise=mask)
        x = O.adam(pb, float(g["lr"]), 5, non_neg=nn)
        assert O.rel_l2(x, g["x_5"]) <= 1e-6
        # grad0 is the REFERENCE's x.grad after its first backward() (masked when mask_noise, fusion.py:340-341)
        g0 = pb.trace[0]["grad"]
        if bool(g["grad0_masked"]):
            g0 = g0 * pb.cmos_mask[None, None]
        assert O.rel_l2(g0, g["grad0"]) <= 1e-6
        lo = np.array([[t["spatial_loss"], t["spectro_temporal_loss"]] for t in pb.trace])
        assert np.allclose(lo, g["losses"], rtol=1e-5)          # the reference's own per-iteration losses
        pb = O.prepare(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=None, mask_noise=mask)
        assert O.rel_l2(O.adam(pb, float(g["lr"]), 1, non_neg=nn), g["x_1"]) <= 1e-6
    g = golden("cg_verbose_tol_t8l4z4_32")
    pb = O.prepare(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=3e-3)
    x = O.cg(pb, 50)
    k = int(g["curr_iter"])
    assert pb.curr_iter == k and O.rel_l2(x, g["x"]) <= 1e-6
    rows = np.array([[t["spatial_loss"], t["spectro_temporal_loss"], t["spatial_loss"] + t["spectro_temporal_loss"],
                      t["rsnew"]] for t in pb.trace[:k - 1]])
    assert np.allclose(rows, g["hist"][:k - 1], rtol=1e-4) and np.all(g["hist"][k - 1:] == 0)


def test_operator_identities():
    """<Op x, y> = size <x, Op.T y> for the normalised transposes; A symmetric PSD."""
    torch.manual_seed(0)
    spc, cmos = O.synthetic_inputs(4, 3, 5, 16, 4)
    pb = O.prepare(spc, cmos, weights={"spatial": 0.3, "spectro_temporal": 0.7}, init_type="baseline", tol=None)
    p, q = torch.randn(pb.dims, dtype=torch.float64), torch.randn(pb.dims, dtype=torch.float64)
    pb.x = pb.x.double()
    ap, aq = O.normal_op(pb, p), O.normal_op(pb, q)
    assert abs(float((ap * q).sum() - (p * aq).sum())) <= 1e-10 * float(ap.norm() * q.norm())
    assert float((ap * p).sum()) >= 0
    y = torch.randn(4, 1, 5, 16, 16, dtype=torch.float64)
    x = torch.randn(4, 3, 5, 16, 16, dtype=torch.float64)
    lhs = (O.sum_keep(x, 1) * y).sum()
    rhs = 3 * (x * O.spread_mean(y, 3, 1)).sum()
    assert abs(float(lhs - rhs)) <= 1e-10 * abs(float(lhs))


def test_collapsed_oracle_matches_eager_oracle():
    """The fp64 small-space CG used to validate shapes no eager restatement fits."""
    w = {"spatial": 0.4, "spectro_temporal": 0.6}
    spc, cmos = O.synthetic_inputs(6, 4, 5, 32, 4, seed=2)
    pb = O.prepare(spc, cmos, weights=w, init_type="baseline", tol=None)
    x0 = pb.x.clone()
    u0, v0 = O.project(pb, x0)
    x = O.cg(pb, 8, with_loss=False)
    a, g, alphas, _ = O.collapsed_cg(pb.spc, pb.cmos, u0[0, 0], v0[:, :, 0], pb.dims, 4, 0.4, 0.6, 8)
    assert O.rel_l2(x, O.expand_small(x0, a, g, 4)) <= 2e-6
    assert np.allclose(alphas[:3], [t["alpha"] for t in pb.trace[:3]], rtol=1e-4)


# ---------------------------------------------------------------- C ABI surface
def _declared_functions():
    src = open(os.path.join(ROOT, "include", "dfx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dfx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from datafusion_amd import _cabi as cabi
    names = _declared_functions()
    assert len(names) >= 20
    assert sorted(cabi.SIGNATURES) == names, "ctypes table and include/dfx.h disagree"
    lib = C.CDLL(cabi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"libdfx.so does not export {n}"
    hdr = open(os.path.join(ROOT, "include", "dfx.h")).read()
    assert cabi.lib().dfx_version() == int(re.search(r"#define DFX_VERSION\s+(\d+)", hdr).group(1))


def test_header_constants_match_python_mirror():
    from datafusion_amd import _cabi as cabi
    src = open(os.path.join(ROOT, "include", "dfx.h")).read()
    for name in ("RSOLD", "RSNEW", "PAP", "ALPHA", "BETA", "TOL", "UU", "PP", "XX", "XP", "VV", "RR", "XXNEW",
                 "NONFINITE", "ITER", "SUM", "SCALE", "LOSS1", "LOSS2", "ADAM_DD", "ADAM_XX", "RA2", "RB2", "RAB"):
        m = re.search(rf"DFX_S_{name}\s*=\s*(\d+)", src)
        assert m and int(m.group(1)) == getattr(cabi, "S_" + name), name
    assert int(re.search(r"DFX_NSCAL\s*=\s*(\d+)", src).group(1)) == cabi.NSCAL
    assert int(re.search(r"#define DFX_NPART\s+(\d+)", src).group(1)) == cabi.NPART
    for i, name in enumerate(("PROJECT", "XP", "P", "PINIT")):
        assert int(re.search(rf"DFX_SWEEP_{name}\s*=\s*(\d+)", src).group(1)) == i


def _v_layers_per_zchunk(r, X, Y):
    """Layers of v_part per z-chunk (include/dfx.h, dfx_plan): r / 4 row parts for r = 16, 32, 64; r for a power-of-two r
    on planes whose rows are not 128-byte aligned (flat map); for the strip tile
    (any other r with lcm(r, 4) <= 64 and Y % 4 == 0 - and every r on planes whose rows are not 128-byte aligned, Y % 32 != 0)
    the most 4-row strips a block row meets; else 1."""
    import math
    if Y % 4 or (r in (2, 4, 8) and Y % 32 == 0):
        return 1
    if r in (16, 32, 64) and Y % 32 == 0:
        return r // 4
    if r in (2, 4, 8, 16, 32, 64):
        return r                                        # flat map: a layer per row of a block
    if 3 <= r <= 128:
        return 2 * r                                    # flat map for any r: a layer per row of a block, times two
    return 1


def _fast_expected(r, Y):
    import math
    return 1 if Y % 4 == 0 and 2 <= r <= 128 else 0


@pytest.mark.parametrize("dims,fast", [((256, 16, 16, 128, 128, 4), 1), ((512, 32, 4, 256, 256, 4), 1),
                                       ((64, 16, 20, 128, 128, 4), 1), ((8, 4, 4, 32, 32, 4), 1), ((6, 5, 3, 24, 24, 4), 1),
                                       ((4, 3, 2, 16, 16, 2), 1), ((3, 2, 11, 64, 128, 4), 1), ((4, 2, 3, 32, 32, 8), 1),
                                       ((2, 2, 2, 10, 6, 2), 0), ((2, 2, 2, 9, 15, 3), 0), ((7, 3, 5, 6, 132, 2), 1),
                                       ((2, 3, 9, 24, 40, 8), 1), ((3, 2, 9, 32, 48, 16), 1), ((2, 2, 3, 64, 96, 32), 1),
                                       ((2, 1, 2, 64, 128, 64), 1), ((3, 2, 5, 15, 120, 3), 1), ((2, 2, 3, 18, 36, 6), 1),
                                       ((2, 2, 3, 34, 68, 17), 1), ((2, 2, 2, 27, 36, 9), 1), ((2, 2, 2, 20, 50, 5), 0),
                                       ((256, 16, 16, 120, 120, 4), 1), ((3, 2, 5, 10, 36, 2), 1), ((2, 2, 9, 48, 80, 16), 1)])
def test_planner_covers_geometry(dims, fast):
    """dfx_plan_make is host-only code: runs without a GPU."""
    from datafusion_amd import _cabi as cabi
    g, p = cabi.Geom(*dims, 0), cabi.Plan()
    assert cabi.lib().dfx_plan_make(C.byref(g), 0, C.byref(p)) == 0
    n_t, n_l, n_z, X, Y, r = dims
    assert p.fast == fast
    if fast:
        assert 0 < p.zs <= p.zc and p.zs * p.n_zchunks >= n_z and p.steps * p.n_chunks >= n_t * n_l
        assert (p.n_zchunks - 1) * p.zs < n_z and p.n_zchunks * p.zs - n_z < p.n_zchunks      # balanced chunks
    else:
        assert p.n_zchunks == 1 and p.cpb == 1 and 1 <= p.n_chunks <= n_t * n_l and p.n_blocks % p.n_chunks == 0
    assert p.u_part_elems == (p.n_chunks // max(p.cpb, 1)) * n_z * X * Y
    assert p.fast == _fast_expected(r, Y)
    assert p.v_part_elems == p.n_zchunks * _v_layers_per_zchunk(r, X, Y) * n_t * n_l * (X // r) * (Y // r)
    assert p.part_elems >= max(p.n_blocks, cabi.EW_MAXBLOCKS) * cabi.NPART
    if fast:
        if r in (16, 32, 64) and Y % 32 == 0:
            wt = (X // r) * -(-Y // 64) * (r // 4)                        # bands x column groups of 64 x row parts
        elif r not in (2, 4, 8) or Y % 32:
            wt = -(-(X * Y // 4) // 64)                                   # flat maps: 64 float4 of the plane per wave
        else:
            wt = -(-(X // r) * (Y // r) // {2: 64, 4: 16, 8: 4}[r])      # wave tiles per plane: 64 / 16 / 4 blocks each
        assert p.cpb >= 1 and p.n_blocks == -(-wt // 4) * p.n_zchunks * -(-p.n_chunks // p.cpb)


def test_planner_rejects_bad_geometry():
    from datafusion_amd import _cabi as cabi
    p = cabi.Plan()
    for dims in ((0, 1, 1, 4, 4, 4), (2, 2, 2, 10, 8, 4), (2, 2, 2, 8, 8, 0)):
        assert cabi.lib().dfx_plan_make(C.byref(cabi.Geom(*dims, 0)), 0, C.byref(p)) == cabi_neg(), dims
    assert b"" != cabi.lib().dfx_last_error()


def cabi_neg():
    return -1


def test_factored_entry_points_validate_arguments_before_touching_the_gpu():
    from datafusion_amd import _cabi as cabi
    lib = cabi.lib()
    g, p = cabi.Geom(4, 3, 2, 16, 16, 4, 0), cabi.Plan()
    assert lib.dfx_plan_make(C.byref(g), 0, C.byref(p)) == 0
    one = C.c_void_p(4096)   # never dereferenced: every call below must fail in its argument checks
    assert lib.dfx_resid_update(C.byref(g), None, one, one, one, 1.0, 1.0, 1.0, None, 0, one, one, one, None) < 0
    assert b"NULL" in lib.dfx_last_error()
    assert lib.dfx_resid_update(C.byref(g), one, one, one, one, 1.0, 1.0, 1.0, None, 0, None, one, one, None) < 0
    args = (C.byref(g), C.byref(p), one, one, one, one, one, one, one, one, 0, p.cpb, 0, 1, None)
    assert lib.dfx_sweep_factored(cabi.SWEEP_PROJECT, *args) < 0          # nothing to factor in a projection
    bad = list(args)
    bad[3] = None                                                          # ra missing
    assert lib.dfx_sweep_factored(cabi.SWEEP_XP, *bad) < 0
    bad = list(args)
    bad[11] = p.cpb + 1                                                    # chunk range outside the plan
    assert lib.dfx_sweep_factored(cabi.SWEEP_XP, *bad) < 0
    with pytest.raises(ValueError, match="residual"):
        from datafusion_amd import FusionCG
        FusionCG(np.ones((2, 2, 2, 2), np.float32), np.ones((2, 8, 8), np.float32),
                 weights={"spatial": 0.5, "spectro_temporal": 0.5}, init_type="zeros", device="cpu", residual="nope")


def test_product_refuses_cpu_device():
    from datafusion_amd import FusionCG, _cabi as cabi
    spc, cmos = O.synthetic_inputs(2, 2, 2, 8, 4)
    with pytest.raises(cabi.DfxError, match="no CPU path"):
        FusionCG(spc, cmos, weights={"spatial": 0.5, "spectro_temporal": 0.5}, init_type="baseline", device="cpu")


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "datafusion_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            assert "oracle" not in re.sub(r'""".*?"""', "", open(os.path.join(pkg, fn)).read(), flags=re.S).replace(
                "# the oracle", ""), fn


# ---------------------------------------------------------------- host logic through the CPU stand-in
@pytest.fixture
def cpu_backend(monkeypatch):
    from datafusion_amd import fusion
    from cpu_backend import CpuStandIn
    monkeypatch.setattr(fusion.Fusion, "_backend_cls", CpuStandIn)


@pytest.mark.parametrize("residual", ["factored", "explicit"])
@pytest.mark.parametrize("defer", [True, False])
@pytest.mark.parametrize("name,iters,kw", CG_CASES[:2] + CG_CASES[5:])
def test_host_logic_cg_vs_goldens(golden, cpu_backend, name, iters, kw, defer, residual):
    from datafusion_amd import FusionCG
    g = golden(name)
    k = iters[-1]
    f = FusionCG(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=None, device="cpu", defer_x=defer,
                 residual=residual, **kw)
    assert O.rel_l2(f.x, g["x0"]) <= 1e-6
    x, _, _ = f(k)
    assert f.curr_iter == k and O.rel_l2(x, g[f"x_{k}"]) <= 2e-5


def test_host_logic_tolerance_break_and_reentry(golden, cpu_backend):
    from datafusion_amd import FusionCG
    g = golden("cg_tol_t8l4z4_32")
    for defer in (True, False):
        f = FusionCG(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=3e-3, device="cpu", defer_x=defer)
        x, _, _ = f(50)
        assert f.curr_iter == int(g["curr_iter"]) and O.rel_l2(x, g["x"]) <= 2e-5
    # a second call continues from the current x with fresh r, p (re-entrancy)
    f = FusionCG(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=None, device="cpu")
    f(2)
    f(3)
    pb = O.prepare(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=None)
    O.cg(pb, 2)
    O.cg(pb, 3)
    assert f.curr_iter == 5 and O.rel_l2(f.x, pb.x) <= 2e-5


def test_host_logic_zeros_diverges_and_verbose_history(golden, cpu_backend):
    from datafusion_amd import FusionCG
    g = golden("cg_t8l4z4_32_r4")
    f = FusionCG(g["spc"], g["cmos"], weights=W(g), init_type="zeros", tol=None, device="cpu")
    with pytest.raises(AssertionError, match="Conjugate gradient diverged"):
        f(3)
    f = FusionCG(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=None, device="cpu", verbose=True)
    f(5)
    assert np.allclose(f.history, g["hist"][:5], rtol=1e-3, atol=1e-13)        # every row, all four columns
    g = golden("cg_verbose_tol_t8l4z4_32")      # verbose + tolerance: rows stop where the loop breaks, the rest stays 0
    f = FusionCG(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=3e-3, device="cpu", verbose=True)
    f(50)
    assert f.curr_iter == int(g["curr_iter"]) and f.history.shape == g["hist"].shape
    assert np.allclose(f.history, g["hist"], rtol=1e-3, atol=1e-13)


@pytest.mark.parametrize("fused", [True, False])
def test_host_logic_adam(golden, cpu_backend, fused):
    from datafusion_amd import FusionAdam
    for name, mask, nn in (("adam_t6l4z3_32", False, False), ("adam_t6l4z3_32_mask_nn", True, True)):
        g = golden(name)
        if fused:
            fe = FusionAdam(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=None, device="cpu",
                            mask_noise=mask, moment1="explicit")
            xe, _, _ = fe(lr=float(g["lr"]), max_iterations=5, non_neg=nn)
            assert O.rel_l2(xe, g["x_5"]) <= 2e-5
        f = FusionAdam(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=None, device="cpu", mask_noise=mask,
                       fused=fused)
        l1, l2 = f.loss()
        assert np.allclose([float(l1), float(l2)], g["loss0"], rtol=1e-3)
        x, _, _ = f(lr=float(g["lr"]), max_iterations=5, non_neg=nn)
        assert O.rel_l2(x, g["x_5"]) <= 2e-5


def test_host_logic_adam_tolerance_break(golden, cpu_backend):
    from datafusion_amd import FusionAdam
    g = golden("adam_tol_t6l4z3_32")
    for kw in ({}, {"moment1": "explicit"}, {"fused": False}):
        f = FusionAdam(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=float(g["tol"]), device="cpu", **kw)
        x, _, _ = f(lr=float(g["lr"]), max_iterations=12, non_neg=True)
        assert f.curr_iter == int(g["curr_iter"]) and O.rel_l2(x, g["x"]) <= 2e-5


def test_oracle_reproduces_adam_tolerance_golden(golden):
    g = golden("adam_tol_t6l4z3_32")
    pb = O.prepare(g["spc"], g["cmos"], weights=W(g), init_type="baseline", tol=float(g["tol"]))
    x = O.adam(pb, float(g["lr"]), 12, non_neg=True)
    assert pb.curr_iter == int(g["curr_iter"]) and O.rel_l2(x, g["x"]) <= 1e-6


def test_operator_mirrors_refuse_cpu_tensors():
    from datafusion_amd import SumOperator, SumPoolOperator, _cabi as cabi
    with pytest.raises(cabi.DfxError, match="no CPU path"):
        SumOperator(size=3, integral_dim=0)(torch.randn(3, 4, 5, 8, 8))
    with pytest.raises(cabi.DfxError, match="no CPU path"):
        SumPoolOperator(size=4, channels=4).T(torch.randn(3, 4, 1, 2, 2))


def test_bench_byte_accounting():
    """32 N bytes per iteration = 5N (x,r,p read; x,r write) + 3N words; this build's
    kernels: sweep XP 5N words, bcast 2N words."""
    n = 256 * 16 * 16 * 128 * 128
    assert 32 * n == 34359738368
    assert (5 + 2) * 4 * n == 28 * n


def test_load_data_input_stage(tmp_path):
    """utils.py:411-427 restated without scikit-image (parity unpinned: see datafusion_amd/io.py)."""
    from datafusion_amd import load_data
    from datafusion_amd.io import resize_like_skimage
    rng = np.random.default_rng(0)
    cmos_file = rng.uniform(0.1, 1.0, (5, 64, 48))              # stored (z, a, b)
    spc = rng.uniform(0.1, 1.0, (6, 4, 8, 8))
    path = tmp_path / "acq.npz"
    np.savez(path, spc=spc, cmos=cmos_file, time_axis=np.arange(6.0), spectral_axis=np.arange(4.0))
    s, c, t, lam = load_data(path, max_xy_size=32)
    assert s.shape == spc.shape and np.array_equal(s, spc) and len(t) == 6 and len(lam) == 4
    assert c.shape == (5, 32, 32)
    # same-size request is the identity up to the transposes: cmos_out[z, j, i] = cmos_file[z, i, j]
    sq = rng.uniform(0.1, 1.0, (3, 16, 16))
    np.savez(path, spc=spc, cmos=sq, time_axis=t, spectral_axis=lam)
    _, c2, _, _ = load_data(path, max_xy_size=16)
    assert np.allclose(c2, np.transpose(sq, (0, 2, 1)), atol=1e-12)
    # shrinking by an integer factor keeps the mean level and smooth ramps (order-1 + anti-aliasing)
    ramp = np.tile(np.linspace(0, 1, 64)[:, None, None], (1, 64, 2))
    out = resize_like_skimage(ramp, (16, 16, 2))
    assert abs(out.mean() - ramp.mean()) < 5e-3 and np.all(np.diff(out[:, 0, 0]) > 0)
    # and feeds the solver's input contract: (t, l, x/r, y/r) and (z, x, y) with integer r
    assert c.shape[1] // s.shape[2] == 4 and c.shape[1] % s.shape[2] == 0
    # the boundary rule is scipy's "mirror" - what skimage translates its default mode "reflect" to (_to_ndimage_mode):
    # whole-sample symmetric, d c b | a b c d.  The first output of a 2x shrink by hand: sigma = 0.5, radius 2
    sig = rng.uniform(0.0, 1.0, 8)
    w = np.exp(-0.5 * np.arange(3) ** 2 / 0.25)
    w /= w[0] + 2 * w[1:].sum()
    f0 = w[0] * sig[0] + w[1] * (sig[1] + sig[1]) + w[2] * (sig[2] + sig[2])          # s[-1] = s[1], s[-2] = s[2]
    f1 = w[0] * sig[1] + w[1] * (sig[0] + sig[2]) + w[2] * (sig[1] + sig[3])          # s[-1] = s[1]
    got = resize_like_skimage(sig.reshape(8, 1, 1), (4, 1, 1))[:, 0, 0]
    assert abs(got[0] - 0.5 * (f0 + f1)) < 1e-12
    # integer stacks are rescaled like img_as_float (resize's preserve_range=False)
    u16 = (rng.uniform(0, 1, (6, 6, 1)) * 65535).astype(np.uint16)
    assert np.allclose(resize_like_skimage(u16, (6, 6, 1)), u16 / 65535.0, atol=1e-12)


def test_integration_doc_struct_layout_matches_header():
    """The ctypes stub printed in INTEGRATION.md must lay dfx_plan out like include/dfx.h."""
    from datafusion_amd import _cabi as cabi
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r'for n in \(("fast".*?)\)\] \+', doc, flags=re.S)
    names32 = re.findall(r'"(\w+)"', m.group(1))
    assert names32 == [f[0] for f in cabi.Plan._fields_ if f[1] is C.c_int32]
    hdr = open(os.path.join(ROOT, "include", "dfx.h")).read()
    body = re.search(r"typedef struct dfx_plan \{(.*?)\} dfx_plan;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    assert re.findall(r"int(?:32|64)_t\s+(\w+);", body) == [f[0] for f in cabi.Plan._fields_]


def test_sweep_slice_bounds_decrease_and_cover():
    """The bands a z-sharded sweep is cut into: non-empty, covering, and the last (whose all-reduce is exposed) the
    shortest."""
    from datafusion_amd.kernels import HipKernels
    for cpb in (1, 2, 3, 5, 8, 64, 256):
        for k in range(1, min(cpb, 16) + 1):
            b = HipKernels._slice_bounds(cpb, k)
            d = [b[i + 1] - b[i] for i in range(k)]
            assert b[0] == 0 and b[-1] == cpb and len(b) == k + 1 and min(d) >= 1
            assert d[-1] == min(d)


@pytest.mark.parametrize("n_z,expect", [(16, (8, 8, 2)), (20, (4, 4, 5)), (5, (8, 5, 1)), (10, (8, 5, 2)), (21, (8, 7, 3)),
                                        (9, (4, 3, 3)), (4, (4, 4, 1)), (1, (4, 1, 1)), (32, (8, 8, 4))])
def test_planner_balances_z_chunks(n_z, expect):
    """(zc, zs, n_zchunks): chunks of equal size zs = ceil(nZ / n) so that every workgroup streams the same bytes."""
    from datafusion_amd import _cabi as cabi
    g, p = cabi.Geom(8, 4, n_z, 32, 32, 4, 0), cabi.Plan()
    assert cabi.lib().dfx_plan_make(C.byref(g), 0, C.byref(p)) == 0
    assert (p.zc, p.zs, p.n_zchunks) == expect


@pytest.mark.parametrize("dims,expect", [((256, 16, 16, 128, 128, 4), (4, 2, 8, 256)), ((256, 16, 16, 128, 128, 16), (4, 2, 8, 256)),
                                         ((256, 16, 16, 128, 128, 2), (8, 8, 2, 256)), ((64, 16, 20, 128, 128, 4), (4, 4, 5, 240)),
                                         ((512, 32, 32, 256, 256, 4), (8, 8, 4, 256)), ((256, 16, 32, 512, 512, 4), (8, 8, 4, 1024)),
                                         ((512, 32, 4, 256, 256, 4), (4, 4, 1, 256)), ((256, 16, 16, 120, 120, 4), (8, 8, 2, 240))])
def test_planner_at_bench_sizes(dims, expect):
    """(zc, zs, n_zchunks, workgroups) on a 256-CU device for the shapes bench.py measures: z-chunks of two planes where
    they were measured to win (row maps r = 4, 8, 16 on planes of at most 16 tile groups, the chip still filled in one
    round: DESIGN.md section 4, profiles/r03_planes_per_zchunk.txt) and the padding-minimal chunking everywhere else."""
    from datafusion_amd import _cabi as cabi
    g, p = cabi.Geom(*dims, 0), cabi.Plan()
    assert cabi.lib().dfx_plan_make(C.byref(g), 256, C.byref(p)) == 0
    assert (p.zc, p.zs, p.n_zchunks, p.n_blocks) == expect


def test_header_is_plain_c_and_library_links_from_c(tmp_path):
    """include/dfx.h must be usable from C (the drop-in boundary is a C ABI): compile a C99 program against it with
    -pedantic, link libdfx.so, run the host-only planner."""
    import shutil
    import subprocess
    from datafusion_amd import _cabi as cabi
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "dfx.h"\n'
                   "int main(void) { dfx_geom g = {2, 3, 5, 8, 12, 4, 0}; dfx_plan p;\n"
                   "  if (dfx_version() != DFX_VERSION) return 2;\n"
                   "  if (dfx_plan_make(&g, 0, &p) != 0) return 3;\n"
                   "  return (p.fast == 1 && p.zs * p.n_zchunks >= 5 && dfx_last_error() != 0) ? 0 : 4; }\n")
    lib_dir = os.path.dirname(cabi.LIB_PATH)
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(exe), "-L", lib_dir, "-ldfx", f"-Wl,-rpath,{lib_dir}"], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def test_host_code_of_the_library_under_asan_and_ubsan(tmp_path):
    """libdfx's host side (planner, argument validation of every entry point) built with AddressSanitizer + UBSan
    (host code only: -Xarch_host) and driven over 20 000 random geometries by tests/native/host_sanitize.cpp.  Every
    call there returns before launching anything, so no GPU is needed."""
    import shutil
    import subprocess
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not (shutil.which("hipcc") and os.path.exists(clang)):
        pytest.skip("hipcc / clang++ not available")
    lib, exe = tmp_path / "libdfx_san.so", tmp_path / "host_sanitize"
    san = ["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined", "-Xarch_host",
           "-fno-sanitize-recover=undefined"]
    subprocess.check_call(["hipcc", "-O1", "-g", "--offload-arch=gfx950", "-shared", "-fPIC", *san, "-o", str(lib),
                           os.path.join(ROOT, "datafusion_amd", "csrc", "dfx_kernels.hip")], cwd=ROOT)
    subprocess.check_call([clang, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o",
                           str(exe), os.path.join(ROOT, "tests", "native", "host_sanitize.cpp"), str(lib),
                           f"-Wl,-rpath,{tmp_path}"], cwd=ROOT)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert p.returncode == 0 and "host_sanitize: OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


# ---------------------------------------------------------------- rank-aware placement (host logic, simulated memory)
class _SimPool:
    """DevicePool over a simulated GPU: chunk g lies in class phys[g]; the probe runs at 5500 GB/s when its two operands
    share a class and 6450 when they do not (with a little noise), as measured on MI355X."""

    def __new__(cls, phys, capacity, noise=0.0, seed=0):
        import random
        from datafusion_amd import placement

        class Sim(placement.DevicePool):
            def __init__(self):
                self.device, self.index = "sim", 0
                self.arena, self.n_chunks, self.cls, self.free = None, 0, {}, {}
                self.same_rate, self.live_views, self.log, self._probe_k = None, 0, [], None
                import threading
                self.lock = threading.RLock()
                self.rng = random.Random(seed)
                self.mapped, self.owned, self.views, self.n_probes = {}, set(), {}, 0
                self.slot = {}                                    # chunk id -> physical slot (the driver's choice)
                self.free_slots = list(range(capacity))

            def _create(self, n_more):
                assert n_more <= len(self.free_slots)
                ids = list(range(self.n_chunks, self.n_chunks + n_more))
                self.n_chunks += n_more
                self.owned |= set(ids)
                for g in ids:
                    self.slot[g] = self.free_slots.pop(0)         # lowest free physical slot first
                self.arena = True
                return ids

            def _destroy_arena(self):
                assert self.live_views == 0
                self.free_slots = sorted(self.free_slots + [self.slot[g] for g in self.owned])
                self.arena, self.n_chunks, self.cls, self.free, self.owned, self.slot = None, 0, {}, {}, set(), {}

            def _map(self, ids):
                assert all(g in self.owned and g not in self.mapped for g in ids)
                base = (len(self.views) + 1 + self.rng.randrange(1000)) << 48
                self.views[base] = list(ids)
                for i, g in enumerate(ids):
                    self.mapped[g] = base + i * placement.CHUNK_BYTES
                return base

            def _unmap(self, ptr):
                for g in self.views.pop(ptr):
                    del self.mapped[g]

            def _release(self, ids):
                for g in ids:
                    assert g not in self.mapped
                    self.owned.discard(g)
                    self.cls.pop(g, None)
                    self.free_slots.append(self.slot.pop(g))
                self.free_slots.sort()

            def _allocatable_chunks(self):
                return len(self.free_slots)

            def _chunk_at(self, ptr):
                for g, a in self.mapped.items():
                    if a <= ptr < a + placement.CHUNK_BYTES:
                        return g
                raise AssertionError("probe outside any mapped chunk")

            def _rate(self, xptr, pptr, reps=2):
                self.n_probes += 1
                a, b = phys[self.slot[self._chunk_at(xptr)]], phys[self.slot[self._chunk_at(pptr)]]
                return (5500.0 if a == b else 6450.0) * (1.0 + noise * (2 * self.rng.random() - 1))

            def _zero(self, ptr, n_float):
                self._chunk_at(ptr)

            def _wrap(self, view, n_float):
                return view          # the test keeps the view object itself alive

            def _device_ctx(self):
                import contextlib
                return contextlib.nullcontext()

            def trim_sim(self):
                self.trim()

        return Sim()


@pytest.mark.parametrize("order", ["fresh", "scattered"])
def test_rank_aware_pool_on_a_simulated_device(order):
    """datafusion_amd/placement.py without a GPU: the classification (self-calibrated threshold, references tried most
    recently matched first), the growth in batches, the layout, the reuse of classified chunks and the fallbacks."""
    import random
    from datafusion_amd import placement
    n = 141
    phys = [0] * 47 + [1] * 47 + [2] * 47            # what a fresh MI355X process is handed (r03_arena_probe.txt)
    if order == "scattered":
        random.Random(5).shuffle(phys)
    chunk_f = placement.CHUNK_BYTES // 4
    p = _SimPool(phys, n, noise=0.01)
    # two 16 GiB vectors: 8 chunks each
    vecs, info = p.alloc_vectors(2, 8 * chunk_f)
    assert info["positions_sharing_a_class"] == 0 and info["chunks_per_vector"] == 8 and len(vecs) == 2
    for a, b in zip(vecs[0].chunks, vecs[1].chunks):
        assert phys[p.slot[a]] != phys[p.slot[b]]
    assert len(p.owned) == 16 and p.live_views == 2 and not p.free          # spares went back to the driver
    if order == "fresh":
        assert info["chunks_created"] > 47                                  # had to allocate past the first class
    created_first = info["chunks_created"]
    # a request while those are alive cannot be served: the caller falls back to torch
    assert p.alloc_vectors(2, 8 * chunk_f) is None
    # freed vectors keep their classification: same request again, no probe
    del vecs
    probes = p.n_probes
    vecs, info = p.alloc_vectors(2, 8 * chunk_f)
    assert info["probes"] == 0 and p.n_probes == probes and info["positions_sharing_a_class"] == 0
    del vecs
    # three vectors (explicit residual): pairwise different at every position; the pool grows
    vecs, info = p.alloc_vectors(3, 8 * chunk_f)
    assert info["positions_sharing_a_class"] == 0
    for t in zip(*(v.chunks for v in vecs)):
        assert len({phys[p.slot[g]] for g in t}) == 3
    del vecs
    # the whole GPU: two vectors of 64 chunks each out of 141
    vecs, info = p.alloc_vectors(2, 64 * chunk_f)
    assert info["positions_sharing_a_class"] == 0
    for a, b in zip(vecs[0].chunks, vecs[1].chunks):
        assert phys[p.slot[a]] != phys[p.slot[b]]
    assert len(p.owned) == 128
    del vecs
    # more than the device holds
    assert p.alloc_vectors(2, 71 * chunk_f) is None
    p.trim()
    assert not p.owned and p.arena is None
    assert created_first <= 141


def test_rank_aware_layout_is_optimal_for_two_vectors_and_counts_conflicts():
    from datafusion_amd.placement import DevicePool as P
    assert P.conflicts(P.layout({0: 48, 1: 47, 2: 47}, 2, 64)) == 0
    assert P.conflicts(P.layout({0: 47, 1: 47, 2: 47}, 3, 47)) == 0
    assert P.conflicts(P.layout({0: 40, 1: 5}, 2, 20)) == 15             # only 5 positions can differ
    assert P.layout({0: 3}, 2, 2) is None                               # not enough chunks at all
    lay = P.layout({0: 10, 1: 10, 2: 10, 3: 1}, 2, 15)
    assert P.conflicts(lay) == 0 and len(lay) == 15
