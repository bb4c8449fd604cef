"""GPU tests added in round 3: the surface fixes the round-2 review asked for, the full-size meeting of the HIP path
with the eager oracle (BASELINE configs[0] and [1]), bench.py's self-launch, and the rank-aware placement."""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import fusion_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"
W = {"spatial": 0.5, "spectro_temporal": 0.5}
TIGHT = 2e-5
GIB = 1 << 30


def WG(g):
    return {"spatial": float(g["w1"]), "spectro_temporal": float(g["w2"])}


@pytest.mark.parametrize("residual,defer", [("factored", True), ("explicit", True), ("factored", False)])
def test_iterate_after_a_tolerance_stop_changes_nothing(residual, defer):
    """start(); iterate(50) leaves through the tolerance test; a further iterate(5) must not recompute alpha from the
    stale u(p), v(p) (x would get beta * alpha * p instead of the alpha * p it is owed): bit-equal to f(50)."""
    from datafusion_amd import FusionCG
    spc, cmos = O.synthetic_inputs(8, 4, 4, 32, 4, seed=6)
    kw = dict(weights=W, init_type="baseline", tol=3e-3, device=DEV, residual=residual, defer_x=defer)
    f = FusionCG(spc, cmos, **kw)
    f.start()
    f.iterate(50)
    assert f._stopped and f.curr_iter < 50
    it = f.curr_iter
    scal = f.k.scal.clone()
    f.iterate(5)
    f.iterate(0)
    assert f.curr_iter == it and torch.equal(scal, f.k.scal)
    f.finish()
    g = FusionCG(spc, cmos, **kw)
    xg, _, _ = g(50, return_numpy=False)
    assert g.curr_iter == it and torch.equal(f.x, xg)
    # a stopped solve stays stopped until start() is called again; then it continues from x like a second __call__
    f.start()
    f.iterate(3)
    f.finish()
    xg2, _, _ = g(3, return_numpy=False)
    assert torch.equal(f.x, xg2)


@pytest.mark.parametrize("kw", [{}, {"moment1": "explicit"}, {"fused": False}])
def test_adam_tolerance_method_matches_the_reference(golden, kw):
    """Fusion.tolerance() (fusion.py:244-252) after a FusionAdam call returns the last ||x - prev_x|| / ||x|| - the
    number the reference's loop compared with tol (its `tolerances` list in the golden), not a stale CG slot."""
    from datafusion_amd import FusionAdam
    g = golden("adam_tol_t6l4z3_32")
    f = FusionAdam(g["spc"], g["cmos"], weights=WG(g), init_type="baseline", tol=float(g["tol"]), device=DEV, **kw)
    assert float(f.tolerance()) == 0.0                              # before any step
    f(lr=float(g["lr"]), max_iterations=12, non_neg=True)
    assert f.curr_iter == int(g["curr_iter"])
    t = f.tolerance()
    assert isinstance(t, torch.Tensor) and t.ndim == 0
    ref = float(g["tolerances"][-1])
    assert abs(float(t) - ref) <= 1e-3 * ref and float(t) < float(g["tol"])
    if kw.get("fused", True):                                        # the device slot the C ABI documents
        from datafusion_amd import _cabi as cabi
        assert abs(float(f.k.scal[cabi.S_TOL]) - ref) <= 1e-3 * ref
    # a fixed number of steps without a tolerance: the method still reports the last step
    f2 = FusionAdam(g["spc"], g["cmos"], weights=WG(g), init_type="baseline", tol=None, device=DEV, **kw)
    f2(lr=float(g["lr"]), max_iterations=2, non_neg=True)
    assert abs(float(f2.tolerance()) - float(g["tolerances"][1])) <= 1e-3 * float(g["tolerances"][1])


def test_squared_l2_and_the_operator_attributes():
    """fusion.py:12-18 on a device tensor, and the attributes of the reference's operator objects (fusion.py:23-34,
    59-76: size, integral_dim, channels, kernel)."""
    from datafusion_amd import FusionCG, SumPoolOperator
    from datafusion_amd.fusion import squared_l2
    torch.manual_seed(0)
    x = torch.randn(3, 5, 7, device=DEV)
    s = squared_l2(x)
    assert isinstance(s, torch.Tensor) and s.device.type == "cuda" and s.ndim == 0
    assert abs(float(s) - float((x.double() ** 2).sum())) <= 1e-5 * float(s)
    assert float(squared_l2(torch.zeros(4, device=DEV))) == 0.0
    R = SumPoolOperator(size=4, channels=3, device=DEV)
    assert tuple(R.kernel.shape) == (3, 1, 4, 4) and bool((R.kernel == 1).all()) and R.kernel.device.type == "cuda"
    spc, cmos = O.synthetic_inputs(3, 5, 2, 16, 4)
    f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV)
    assert (f.T.size, f.T.integral_dim, f.S.size, f.S.integral_dim, f.D.size, f.D.integral_dim) == (3, 0, 5, 1, 2, 2)
    assert f.R.size == 4 and f.R.channels == 5 and tuple(f.R.kernel.shape) == (5, 1, 4, 4)
    # the reference's loss terms written with its own helper on the mirrors' outputs equal loss()
    l1, l2 = f.loss()
    m1 = 0.5 * squared_l2(f.cmos - f.T(f.S(f.x)).reshape(f.cmos.shape))
    m2 = 0.5 * squared_l2(f.spc - f.R(f.D(f.x)).reshape(f.spc.shape))
    assert np.allclose([float(l1), float(l2)], [float(m1), float(m2)], rtol=1e-4)


def _host_has(n_bytes):
    for ln in open("/proc/meminfo"):
        if ln.startswith("MemAvailable"):
            return int(ln.split()[1]) * 1024 >= n_bytes
    return False


def _rel_l2_vs_collapsed(x, x0, a, g, r):
    a, g = a.to(x.device), g.to(x.device)
    num = torch.zeros((), dtype=torch.float64, device=x.device)
    den = torch.zeros((), dtype=torch.float64, device=x.device)
    for t in range(x.shape[0]):
        ref = x0[t].double() + a[None] + g[t].repeat_interleave(r, 1).repeat_interleave(r, 2)[:, None]
        num += ((x[t].double() - ref) ** 2).sum()
        den += (ref ** 2).sum()
    return float(torch.sqrt(num / den))


def test_c2_full_size_two_iterations_vs_the_eager_oracle():
    """BASELINE.json configs[1] (t256 x l16 x z16 x 128 x 128) meets the reference's own op sequence
    (fusion.py:423-456 restated in oracle/fusion_oracle.py, eager fp32 torch on the host) directly, not through the
    collapsed algebra: two iterations, both residual representations.  At this size the eager fp32 dot products over
    2^30 elements are themselves 3e-3 off (alpha 0.14142 against 0.14189 in float64; the HIP path: 0.14189), so the
    distance HIP - oracle IS the oracle's distance from the float64 recurrence: asserted as such (triangle), with the
    HIP iterate within 1e-6 of float64, and against north_star's 1e-4 with what the reference's own rounding leaves."""
    from datafusion_amd import FusionCG
    if not _host_has(70 * GIB):
        pytest.skip("the eager oracle at C2 size holds ~52 GB of host temporaries")
    spc, cmos = O.synthetic_inputs(256, 16, 16, 128, 4)
    pb = O.prepare(spc, cmos, weights=W, init_type="baseline", tol=None)
    xo = O.cg(pb, 2, with_loss=False)
    for residual in ("factored", "explicit"):
        f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, residual=residual)
        x0 = f.x.clone()
        f._project(x0)
        u0, v0 = f.k.u.cpu().clone(), f.k.v.cpu().clone()
        x, _, _ = f(2, return_numpy=False)
        a, g, alphas, _ = O.collapsed_cg(f.spc.cpu(), f.cmos.cpu(), u0, v0, f.x_shape, 4, 0.5, 0.5, 2)
        e_h64 = _rel_l2_vs_collapsed(x, x0, a, g, 4)
        e_o64 = _rel_l2_vs_collapsed(xo.to(DEV), x0, a, g, 4)
        e_ho = float(O.rel_l2(x.cpu(), xo))
        print(f"C2 full size, 2 iterations, {residual} residual: rel-L2 HIP vs eager oracle {e_ho:.3e}; HIP vs float64 "
              f"{e_h64:.3e}; eager oracle vs float64 {e_o64:.3e}; alpha_1 oracle {pb.trace[0]['alpha']:.6e} float64 {alphas[0]:.6e}")
        assert e_h64 <= 1e-6, (residual, e_h64)
        assert abs(e_ho - e_o64) <= 2e-6, (residual, e_ho, e_o64)
        assert e_ho <= 1e-4 + 1e-6 or e_o64 > 1e-4, (residual, e_ho, e_o64)
        del f, x, x0
    del pb, xo


def test_readme_flow_at_its_stated_size(tmp_path):
    """BASELINE.json configs[0] at the size the README and analysis/runtime.py:73-76,100-115 use: load_data(npz,
    max_xy_size=128) on a cells-shaped acquisition (t64 x l16 x z20, CMOS stored at 256 x 256), FusionCG(weights 0.5 /
    0.5, baseline init, tol=3e-3)(20): stop index and iterate against the eager oracle.  (The Zenodo file itself is
    not available offline: cells-like synthetic data of that shape.)"""
    from datafusion_amd import FusionCG, load_data
    if not _host_has(30 * GIB):
        pytest.skip("the eager oracle at this size holds ~16 GB of host temporaries")
    spc, cmos = O.cells_like_inputs(64, 16, 20, 128, 4, seed=9)              # (64, 16, 32, 32), (20, 128, 128)
    spc, cmos = np.asarray(spc, dtype=np.float32), np.asarray(cmos, dtype=np.float32)
    big = torch.nn.functional.interpolate(torch.from_numpy(cmos)[None], scale_factor=2, mode="bilinear",
                                          align_corners=False)[0]
    path = tmp_path / "cells_synth_128.npz"
    np.savez(path, spc=spc, cmos=big.numpy(), time_axis=np.linspace(0, 12.5, 64), spectral_axis=np.linspace(500, 700, 16))
    spc_l, cmos_l, t_ax, l_ax = load_data(path, max_xy_size=128, device=DEV)
    assert spc_l.shape == (64, 16, 32, 32) and cmos_l.shape == (20, 128, 128)
    f = FusionCG(spc_l, cmos_l, weights=W, init_type="baseline", tol=3e-3, mask_noise=False, total_energy=1, device=DEV,
                 seed=42)
    x0 = f.x.clone()
    f._project(x0)
    u0, v0 = f.k.u.cpu().clone(), f.k.v.cpu().clone()
    x, spc_n, cmos_n = f(max_iterations=20)
    pb = O.prepare(torch.from_numpy(spc_l), torch.from_numpy(cmos_l), weights=W, init_type="baseline", tol=3e-3)
    xo = O.cg(pb, 20, with_loss=False)
    assert f.curr_iter == pb.curr_iter and x.shape == (64, 16, 20, 128, 128)
    assert O.rel_l2(spc_n, pb.spc) <= 1e-6 and O.rel_l2(cmos_n, pb.cmos) <= 1e-6
    # three-way, as at configs[1]'s size: at 3.4e8 elements the eager fp32 dot products of the reference's op sequence
    # are the larger error; the float64 recurrence (same number of iterations) tells the two apart
    a, g, _, _ = O.collapsed_cg(f.spc.cpu(), f.cmos.cpu(), u0, v0, f.x_shape, 4, 0.5, 0.5, f.curr_iter)
    e_h64 = _rel_l2_vs_collapsed(f.x, x0, a, g, 4)
    e_o64 = _rel_l2_vs_collapsed(xo.to(DEV), x0, a, g, 4)
    e_ho = float(O.rel_l2(x, xo))
    print(f"README flow at 128^2, t64 x l16 x z20: stopped after {f.curr_iter} (oracle {pb.curr_iter}); rel-L2 HIP vs eager "
          f"oracle {e_ho:.3e}; HIP vs float64 {e_h64:.3e}; eager oracle vs float64 {e_o64:.3e}")
    assert e_h64 <= 1e-6, e_h64
    assert abs(e_ho - e_o64) <= 2e-6, (e_ho, e_o64)
    assert e_ho <= 1e-4 + 1e-6 or e_o64 > 1e-4, (e_ho, e_o64)


@pytest.mark.parametrize("dims", [(5, 3, 6, 32, 64, 4), (4, 2, 9, 48, 16, 4), (3, 3, 5, 24, 40, 8), (6, 2, 4, 20, 36, 2),
                                  (3, 2, 3, 18, 30, 3), (4, 3, 5, 24, 36, 3), (3, 2, 4, 30, 60, 5), (2, 3, 9, 18, 96, 6),
                                  (2, 3, 3, 28, 56, 7), (3, 2, 3, 32, 48, 16), (2, 3, 10, 48, 80, 16), (3, 2, 5, 64, 96, 32),
                                  (2, 2, 3, 64, 128, 64)])
def test_solver_on_non_square_planes_vs_the_oracle(dims):
    """X != Y through the whole solver (the reference's baseline() assumes square planes, baseline.py:38, but its CG
    does not: init_type="random"), both residual representations, fast kernels (row maps r = 2, 4, 8 and row-part tile r = 16, 32, 64 when Y % 32 == 0, else the flat map; strip tile for other r)
    and generic path (r = 6, or Y % 4 != 0), unequal weights, against the eager oracle."""
    from datafusion_amd import FusionCG
    n_t, n_l, n_z, X, Y, r = dims
    g = torch.Generator().manual_seed(1)
    spc = 0.1 + 0.9 * torch.rand(n_t, n_l, X // r, Y // r, generator=g)
    cmos = 0.1 + 0.9 * torch.rand(n_z, X, Y, generator=g)
    w = {"spatial": 0.3, "spectro_temporal": 0.7}
    xo = O.cg(O.prepare(spc, cmos, weights=w, init_type="random", tol=None), 6, with_loss=False)
    for residual in ("factored", "explicit"):
        f = FusionCG(spc, cmos, weights=w, init_type="random", tol=None, device=DEV, residual=residual)
        assert f.k.plan.fast == (1 if Y % 4 == 0 and 2 <= r <= 128 else 0)
        x, _, _ = f(6, return_numpy=False)
        assert float(O.rel_l2(x.cpu(), xo)) <= TIGHT, (dims, residual)


def test_sharded_fuzz_three_processes_on_one_gpu():
    """tools/fuzz_sharded.py, 12 random cases: groups of 2 and 3 ranks (processes sharing cuda:0 over gloo), uneven plane
    counts, r in {2, 3, 4, 8}, masks, tolerance stops, banded sweeps - CG (factored / explicit, deferred / textbook) and
    Adam (fused factored / explicit first moment, autograd) with the REAL kernels, each gathered volume against the CPU
    oracle (100 cases: profiles/r03_fuzz_sharded.txt).  Seed 11 holds a case (t5 l4 z2 16x16 r8) where ONE element of
    10 240 has a gradient of the size of Adam's eps, so its first step differs by 0.26 lr between HIP and the oracle
    (2.2e-5 in rel-L2, sharded or not): the tool then checks sharded == unsharded (1e-7) and north_star's 1e-4."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_sharded.py"), "--cases", "12", "--seed", "11",
                        "--port", "29677"], capture_output=True, text=True, timeout=280, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    assert "12 cases, worst rel-L2" in p.stdout and "failed: []" in p.stdout, p.stdout[-2000:]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` as the driver runs it, without a launcher: the parent starts the ranks as child
    processes and passes rank 0's one line through.  One rank here (DFX_FORCE_COMM=1 takes the RCCL path with a
    one-rank group); asking for more GPUs than the box has fails at once, before anything touches a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "tiny", "--steps", "3",
                        "--warmup", "1", "--no-cpu", "--single"], capture_output=True, text=True, timeout=170,
                       env=dict(env, DFX_FORCE_COMM="1"))
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["comm"]["world_size"] == 1 and d["comm"]["backend"] == "nccl"
    assert d["timed_blocks"]["repeats"] == 3 and d["value"] > 0
    if torch.cuda.device_count() < 2:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny"],
                           capture_output=True, text=True, timeout=60, env=env)
        assert p.returncode != 0 and p.stdout.strip() == ""
        assert "needs 2 GPUs, found 1" in p.stderr


@pytest.mark.parametrize("shape,r", [("256,16,16,120", 4), ("256,16,16,120", 3), ("256,16,16,128", 16), ("256,16,16,100", 2)])
def test_lane_maps_at_bench_size_vs_fp64_collapsed_cg(shape, r):
    """Every lane map beside r = 4's row map at the size bench.py measures it (about 2^30 elements; offsets beyond 2^31
    bytes, hundreds of workgroups, thousands of slabs per workgroup): the flat map (r = 4 and r = 2 on planes whose rows are
    not 128-byte aligned), the flat map for any r (r = 3) and the row-part tile (r = 16); 10 CG iterations against the fp64
    collapsed CG, north_star's 1e-4 asserted by the tool, ~8e-8 measured (profiles/r03_verify_lane_maps.txt)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verify_large.py"), "--shape", shape, "--r", str(r)],
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["fast"] == 1 and out["rel_l2_vs_fp64_collapsed"] <= 1e-6, out


def test_bench_two_rank_control_flow_rehearsal():
    """The N > 1 control flow of bench.py with TWO real processes: self-launch, the strong-scaling leg (rank 0 runs the
    one-GPU anchor first while rank 1 waits at the barrier, then both run their halves), the weak-scaled headline, the
    max-over-ranks timing, one JSON line from rank 0, clean shutdown.  The box has one GPU, so the ranks share it and
    the group is gloo (staged through the host; RCCL refuses two ranks on one device): the numbers mean nothing, the
    line, its fields and the exit code do.  Two RCCL ranks on two GPUs remain unverified."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(DFX_BENCH_BACKEND="gloo", DFX_BENCH_SHARE_GPU="1", DFX_BENCH_STRONG="1", DFX_BENCH_STRONG_WORKLOAD="cells",
               DFX_BENCH_CONFIG4="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "3",
                        "--warmup", "1", "--no-cpu"], capture_output=True, text=True, timeout=280, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["comm"]["world_size"] == 2 and d["comm"]["backend"] == "gloo"
    assert d["value"] == pytest.approx(2 * d["iters_per_s"])
    assert "other_residual_mode" in d
    s = d["config3_strong_scaling"]
    assert s["n_gpus"] == 2 and s["scaling"] == "strong" and "10 z-planes per GPU" in s["workload"]
    assert s["comm"]["world_size"] == 2
    assert "iters_per_s" in s["one_gpu_anchor"], s["one_gpu_anchor"]
    assert s["strong_scaling_efficiency"] == pytest.approx(s["iters_per_s"] / (2 * s["one_gpu_anchor"]["iters_per_s"]))
    s4 = d["config4_strong_scaling"]          # the BASELINE configs[3] leg of an 8-GPU run (no anchor), same rehearsal volume
    assert s4["n_gpus"] == 2 and s4["one_gpu_anchor"] is None and s4["iters_per_s"] > 0


def test_rank_aware_placement_with_small_chunks_gives_the_same_bits(monkeypatch):
    """datafusion_amd/placement.py end to end on vectors of a few hundred MiB (128 MiB chunks, forced on): vectors mapped
    chunk by chunk through dfx_arena_* behave like torch allocations - same bits as placement="plain" - for two vectors
    (factored residual), three (explicit) and Adam; freed vectors are reused without new probes; trim() gives
    everything back."""
    from datafusion_amd import FusionAdam, FusionCG, placement
    monkeypatch.setattr(placement, "CHUNK_BYTES", 128 << 20)
    monkeypatch.setattr(placement, "MIN_VECTOR_BYTES", 0)
    monkeypatch.setattr(placement, "_pools", {})
    monkeypatch.setenv("DFX_ARENA_DEBUG", "1")
    # 64 MiB probes are too short to tell the memory classes apart reliably; this test is about the plumbing, so the
    # classes are made up (chunk id mod 3) while every probe still runs the real kernel on the real mapped chunks
    real = placement.DevicePool._pair_rate

    def made_up(self, a, b, at, off, reps=2):
        real(self, a, b, at, off, 1)
        return 5500.0 if a % 3 == b % 3 else 6450.0

    monkeypatch.setattr(placement.DevicePool, "_pair_rate", made_up)
    shape = (32, 16, 8, 128, 4)                                   # 2^26 elements = 256 MiB per vector: 2 chunks
    spc, cmos = O.cells_like_inputs(*shape, seed=31)
    free0 = torch.cuda.mem_get_info()[0]
    ref = {}
    for residual in ("factored", "explicit"):
        f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, residual=residual,
                     placement="plain")
        assert f.k.placement_info is None
        ref[residual], _, _ = f(6)
        del f
    for i, residual in enumerate(("factored", "explicit", "explicit")):
        f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, residual=residual)
        info = f.k.placement_info
        assert info is not None and info["vectors"] == (2 if residual == "factored" else 3)
        assert info["chunks_per_vector"] == 2 and "pair_sweep_GBps" in info
        if i == 2:
            assert info["probes"] == 0 and info["chunks_created"] == 0        # the freed vectors' chunks, still classified
        x, _, _ = f(6)
        assert np.array_equal(x, ref[residual])
        assert not f.k._reserved                                   # every reserved vector was handed out
        del f, x
    pool = placement.pool(DEV)
    assert pool.live_views == 0 and sum(len(v) for v in pool.free.values()) == 6
    fa = FusionAdam(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, placement="plain")
    xa, _, _ = fa(lr=1e-7, max_iterations=3)
    fb = FusionAdam(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV)
    assert fb.k.placement_info is not None and fb.k.placement_info["vectors"] == 2
    xb, _, _ = fb(lr=1e-7, max_iterations=3)
    assert np.array_equal(xa, xb)
    del fa, fb
    # random init goes through the reserved vectors too
    fr = FusionCG(spc, cmos, weights=W, init_type="random", tol=None, device=DEV)
    fp = FusionCG(spc, cmos, weights=W, init_type="random", tol=None, device=DEV, placement="plain")
    assert torch.equal(fr.x, fp.x)
    del fr, fp
    placement.trim()
    torch.cuda.empty_cache()
    assert pool.arena is None and torch.cuda.mem_get_info()[0] >= free0 - (1 << 30)
    # the layout is an optimisation: when the virtual-memory API fails, the solve runs on torch allocations
    from datafusion_amd import _cabi as cabi

    def broken(self, n_more):
        raise cabi.DfxError("dfx_arena_create failed (code 801): hipMemCreate: operation not supported")

    monkeypatch.setattr(placement.DevicePool, "_create", broken)
    fz = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV)
    assert "failed" in fz.k.placement_info and not fz.k._reserved
    xz, _, _ = fz(6)
    assert np.array_equal(xz, ref["factored"])
    del fz


def test_rank_aware_placement_on_a_16_gib_pair():
    """The per-GPU slab of BASELINE configs[2] at G = 8 (16 GiB per vector), default placement: a layout without two
    vectors in one memory class at any position, the pair streaming in the fast regime (0.79 of the 8 TB/s peak; the
    slow regime is 0.66 - 0.69), and the same bits as torch allocations."""
    from datafusion_amd import FusionCG, placement
    if placement.available_bytes(DEV) < 80 * GIB:
        pytest.skip("needs 80 GiB of free device memory")
    spc, cmos = O.synthetic_inputs(512, 32, 4, 256, 4)
    f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV)
    info = f.k.placement_info
    print("placement:", info)
    assert info is not None and info["positions_sharing_a_class"] == 0 and info["chunks_per_vector"] == 8
    # the fast regime, judged against THIS box's own slow regime (the probe's rate on two halves of one chunk, which
    # share a class by construction; MI355X: ~5.5 against ~6.4 TB/s) rather than against an absolute figure
    same = placement.pool(DEV).same_rate
    assert same and info["pair_sweep_GBps"] >= 1.06 * same, (info, same)
    x, _, _ = f(3, return_numpy=False)
    g = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, placement="plain")
    y, _, _ = g(3, return_numpy=False)
    assert all(torch.equal(x[t], y[t]) for t in range(0, x.shape[0], 37))
    assert int(sum(int(x[t].view(torch.int32).sum(dtype=torch.int64)) for t in range(x.shape[0]))) == \
        int(sum(int(y[t].view(torch.int32).sum(dtype=torch.int64)) for t in range(y.shape[0])))
    del f, g, x, y
    placement.trim()
    torch.cuda.empty_cache()


def test_dfx_adam_state_round_trips_with_torch_optim_adam():
    """DfxAdam's state is exactly torch.optim.Adam's (step, exp_avg, exp_avg_sq; the launch scratch is not state): a
    checkpoint of either optimizer continues in the other, eps = 0 included (IEEE sqrt / division in dfx_adam_apply),
    and last_sq_norms sums over all parameters of a step."""
    from datafusion_amd import DfxAdam
    torch.manual_seed(11)
    shapes = [(1000,), (257, 33)]
    x0 = [torch.randn(s, device=DEV) for s in shapes]
    grads = [[torch.randn(s, device=DEV) * (k + 1) for s in shapes] for k in range(4)]
    pa = [torch.nn.Parameter(t.clone()) for t in x0]
    pb = [torch.nn.Parameter(t.clone()) for t in x0]
    oa = torch.optim.Adam(pa, lr=1e-2, eps=0.0)
    ob = DfxAdam(pb, lr=1e-2, eps=0.0)

    def step(opt, ps, gs):
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        opt.step()

    for k in range(2):
        prev = [p.detach().clone() for p in pb]
        step(oa, pa, grads[k])
        step(ob, pb, grads[k])
        dd = sum(float(((p.detach() - q).double() ** 2).sum()) for p, q in zip(pb, prev))
        xx = sum(float((p.detach().double() ** 2).sum()) for p in pb)
        got = ob.last_sq_norms.cpu()
        assert abs(float(got[0]) - dd) <= 1e-6 * dd and abs(float(got[1]) - xx) <= 1e-6 * xx
    assert set(ob.state[pb[0]]) == {"step", "exp_avg", "exp_avg_sq"} == set(oa.state[pa[0]])
    # swap the checkpoints: torch's state into DfxAdam and DfxAdam's into torch
    sa, sb = oa.state_dict(), ob.state_dict()
    oa2 = torch.optim.Adam(pa, lr=1e-2, eps=0.0)
    ob2 = DfxAdam(pb, lr=1e-2, eps=0.0)
    oa2.load_state_dict(sb)
    ob2.load_state_dict(sa)
    for k in range(2, 4):
        step(oa2, pa, grads[k])
        step(ob2, pb, grads[k])
    for p, q in zip(pa, pb):
        assert O.rel_l2(q.detach().cpu(), p.detach().cpu()) <= 1e-6
        assert int(ob2.state[q]["step"]) == 4 == int(oa2.state[p]["step"])
        assert O.rel_l2(ob2.state[q]["exp_avg_sq"].cpu(), oa2.state[p]["exp_avg_sq"].cpu()) <= 1e-6


@pytest.mark.parametrize("dims", [(6, 4, 5, 32, 64, 4), (3, 2, 9, 24, 24, 2), (4, 3, 7, 16, 48, 8), (3, 2, 6, 24, 36, 3),
                                  (2, 3, 3, 14, 56, 7)])
def test_adam_apply_project_equals_apply_then_project(dims):
    """dfx_adam_apply_project (DfxAdam's step in the sweep's lane-mapped structure, fused with the projections of the
    stepped volume) against dfx_adam_apply followed by a projection sweep: x, exp_avg, exp_avg_sq bit for bit (the same
    IEEE arithmetic element by element), u, v and the two norms to rounding; clamp on and off, eps = 0 included."""
    from datafusion_amd import _cabi as cabi
    from datafusion_amd.kernels import HipKernels
    from datafusion_amd.optim import DfxAdam
    torch.manual_seed(5)
    n_t, n_l, n_z, X, Y, r = dims
    k = HipKernels(n_t, n_l, n_z, X, Y, r, DEV)
    assert k.plan.fast == 1
    for non_neg, eps in ((False, 1e-8), (True, 0.0)):
        x0 = torch.randn(k.shape, device=DEV)
        runs = []
        for proj in (None, "k"):
            x = torch.nn.Parameter(x0.clone())
            opt = DfxAdam([x], lr=1e-2, eps=eps, non_neg=non_neg, projectors={x: k} if proj else None)
            for step in range(3):
                x.grad = torch.sin(x0 * (step + 1.5))
                opt.step()
                assert (x in opt.projected) == bool(proj)
            st = opt.state[x]
            runs.append((x.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), opt.last_sq_norms.clone()))
        for a, b in zip(runs[0][:3], runs[1][:3]):
            assert torch.equal(a, b)
        assert torch.allclose(runs[0][3], runs[1][3], rtol=1e-6)
        xd = runs[1][0].double()
        u = xd.sum((0, 1))
        v = xd.sum(2).reshape(n_t, n_l, X // r, r, Y // r, r).sum((3, 5))
        assert O.rel_l2(k.u.cpu(), u.cpu()) <= 1e-6 and O.rel_l2(k.v.cpu(), v.cpu()) <= 1e-6


@pytest.mark.parametrize("mask", [False, True])
def test_autograd_adam_projects_inside_the_optimizer_step(mask):
    """FusionAdam(fused=False): with project_in_step (default) the optimizer step leaves u(x), v(x) of the stepped volume
    and the next forward skips its projection pass - the same iterate bit for bit as with the separate pass, both
    within the tight bar of the oracle; a write to x between two steps (torch's version counter) forces a fresh
    projection."""
    from datafusion_amd import FusionAdam
    spc, cmos = O.cells_like_inputs(6, 4, 5, 32, 4, seed=3)
    lr = 1e-2 / (6 * 4 * 5 * 32 * 32)
    xo = O.adam(O.prepare(spc, cmos, weights=W, init_type="baseline", tol=None, mask_noise=mask), lr, 4, non_neg=True)
    out = []
    for proj in (True, False):
        f = FusionAdam(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, fused=False, mask_noise=mask,
                       project_in_step=proj)
        x, _, _ = f(lr=lr, max_iterations=4, non_neg=True, return_numpy=False)
        assert (f._uv_of is not None) == proj
        assert float(O.rel_l2(x.detach().cpu(), xo)) <= TIGHT
        out.append(x.detach().clone())
    assert torch.equal(out[0], out[1])
    # x written behind the cache's back: the forward must project again
    f = FusionAdam(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, fused=False, mask_noise=mask)
    f(lr=lr, max_iterations=1, non_neg=True, return_numpy=False)
    with torch.no_grad():
        f.x.mul_(2.0)
    l1, l2 = f.loss_autograd()
    m1, m2 = f.loss()
    assert np.allclose([float(l1.detach()), float(l2.detach())], [float(m1), float(m2)], rtol=1e-6)


def test_fused_adam_rejects_eps_zero():
    """The fused 5-D kernels use the hardware sqrt / reciprocal, which need eps > 0 to keep the quotient finite where
    exp_avg_sq is denormal: refused with a clear error instead of returning inf."""
    from datafusion_amd import FusionAdam, _cabi as cabi
    spc, cmos = O.synthetic_inputs(3, 2, 2, 16, 4)
    f = FusionAdam(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, adam_eps=0.0)
    with pytest.raises(cabi.DfxError, match="eps > 0"):
        f(lr=1e-7, max_iterations=1)
