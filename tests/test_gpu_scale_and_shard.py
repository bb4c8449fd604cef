"""GPU tests at BASELINE.json's full single-GPU size (size-independent checks: fp64 collapsed CG,
energy conservation, schedule equivalence) and of the z-sharded path with the real HIP kernels
(two ranks sharing cuda:0, gloo group staged through the host)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import fusion_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"
W = {"spatial": 0.5, "spectro_temporal": 0.5}


def _gpu_rel_l2_vs_collapsed(x, x0, a, g, r):
    """|| x - (x0 + B1 a + B2 g) || / || . || in float64 on the GPU, one time bin at a time."""
    a = a.to(x.device)
    g = g.to(x.device)
    num = torch.zeros((), dtype=torch.float64, device=x.device)
    den = torch.zeros((), dtype=torch.float64, device=x.device)
    for t in range(x.shape[0]):
        up = g[t].repeat_interleave(r, 1).repeat_interleave(r, 2)          # (nL, X, Y)
        ref = x0[t].double() + a[None] + up[:, None]
        num += ((x[t].double() - ref) ** 2).sum()
        den += (ref ** 2).sum()
    return float(torch.sqrt(num / den))


def test_c2_full_size_50_iterations_vs_collapsed_fp64():
    """BASELINE.json configs[1]: t256 x l16 x z16 x 128 x 128, 50 iterations, tol=None."""
    from datafusion_amd import FusionCG
    spc, cmos = O.synthetic_inputs(256, 16, 16, 128, 4)
    f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV)
    x0 = f.x.clone()
    # energy conservation of the initialiser (cells.ipynb check): sum x = sum spc = sum cmos = 1
    assert abs(float(x0.double().sum()) - 1.0) <= 1e-5
    f._project(x0)
    u0, v0 = f.k.u.cpu().clone(), f.k.v.cpu().clone()
    x, _, _ = f(50, return_numpy=False)
    assert f.curr_iter == 50
    a, g, alphas, rs = O.collapsed_cg(f.spc.cpu(), f.cmos.cpu(), u0, v0, f.x_shape, 4, 0.5, 0.5, 50)
    err = _gpu_rel_l2_vs_collapsed(x, x0, a, g, 4)
    print(f"C2 50 iterations: rel-L2 vs fp64 collapsed CG = {err:.3e}; rsnew gpu {float(f.k.scal[1]):.4e} fp64 {rs[-1]:.4e}")
    assert err <= 1e-4
    # the residual the device tracked follows the fp64 one
    assert abs(float(f.k.scal[1]) - rs[-1]) <= 0.05 * rs[-1]
    # the textbook schedule with an explicit 5-D residual gives the same iterate
    del x0
    f2 = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, defer_x=False, residual="explicit")
    x2, _, _ = f2(50, return_numpy=False)
    d = float(torch.linalg.vector_norm((x2 - x).double().flatten()) / torch.linalg.vector_norm(x.double().flatten()))
    print(f"factored/deferred vs explicit/textbook: rel-L2 = {d:.3e}")
    assert d <= 5e-6, d


GIB = 1 << 30
# BASELINE.json configs[2] / configs[3] (t512 x l32 x z32 x 256^2 and t256 x l16 x z32 x 512^2, 2^35 elements):
# what ONE rank of the 8-GPU z-split holds (4 planes, 2^32 elements, 16 GiB per vector: element offsets pass 2^32) ...
SLABS_2P32 = {"c3g8": (512, 32, 4, 256, 4), "c4g8": (256, 16, 4, 512, 4)}
# ... and the whole volumes on one GPU (128 GiB per vector; x + p = 256 GiB: factored residual only)
WHOLE_2P35 = {"c3": (512, 32, 32, 256, 4), "c4": (256, 16, 32, 512, 4)}
_collapsed_cache = {}


def _need_free(n_bytes, what):
    from datafusion_amd import placement
    torch.cuda.empty_cache()
    free = placement.available_bytes(DEV)      # what the driver has + the classified chunks the process's pool keeps
    if free < n_bytes:
        msg = f"{what}: needs {n_bytes / GIB:.0f} GiB of free device memory, {free / GIB:.0f} GiB available"
        print("SKIP", msg)
        pytest.skip(msg)


@pytest.mark.parametrize("residual", ["factored", "explicit"])
@pytest.mark.parametrize("name", sorted(SLABS_2P32))
def test_config3_config4_per_gpu_slabs_2p32_elements_vs_collapsed_fp64(name, residual):
    """fusion.py:435-456 on the per-GPU slabs of configs 3 / 4 at G = 8: 10 iterations, both residual representations,
    against the float64 small-space CG (oracle: collapsed_cg), compared on the GPU in float64."""
    from datafusion_amd import FusionCG
    shape = SLABS_2P32[name]
    n_vec = (3 if residual == "explicit" else 2) + 1                      # + the copy of x0
    _need_free(n_vec * 16 * GIB + 4 * GIB, f"{name} slab, {residual} residual")
    spc, cmos = O.synthetic_inputs(*shape)
    f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, residual=residual)
    assert f.k.N == 1 << 32 and f.k.plan.fast == 1
    x0 = f.x.clone()
    f._project(x0)
    u0, v0 = f.k.u.cpu().clone(), f.k.v.cpu().clone()
    x, _, _ = f(10, return_numpy=False)
    if name not in _collapsed_cache:          # the same for both residual modes (u0, v0 are deterministic)
        _collapsed_cache[name] = O.collapsed_cg(f.spc.cpu(), f.cmos.cpu(), u0, v0, f.x_shape, 4, 0.5, 0.5, 10)
    a, g, alphas, rs = _collapsed_cache[name]
    err = _gpu_rel_l2_vs_collapsed(x, x0, a, g, 4)
    rs_gpu, al_gpu = float(f.k.scal[1]), float(f.k.scal[24])      # RSNEW, ALPHA_LAST
    print(f"{name} {residual}: N = 2^32, 10 iterations: rel-L2 vs fp64 collapsed CG = {err:.3e}; rsnew gpu {rs_gpu:.9e} "
          f"fp64 {rs[-1]:.9e}; alpha gpu {al_gpu:.7e} fp64 {alphas[-1]:.7e}")
    assert err <= 5e-6, err                     # bar 1e-4 (north_star); measured ~1e-7
    assert abs(rs_gpu - rs[-1]) <= 1e-5 * rs[-1] and abs(al_gpu - alphas[-1]) <= 1e-5 * abs(alphas[-1])
    del x, x0, f
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", sorted(WHOLE_2P35))
def test_config3_config4_whole_2p35_elements_on_one_gpu_vs_collapsed_fp64(name):
    """BASELINE.json configs[2] / configs[3] whole on ONE MI355X (2^35 elements; 64-bit slab offsets, 4 z-chunks, the
    running fp32 u flush over 256 chunks per workgroup): 10 iterations vs the float64 small-space CG."""
    from datafusion_amd import FusionCG
    shape = WHOLE_2P35[name]
    _need_free(270 * GIB, f"{name} whole (x + p = 256 GiB)")
    spc, cmos = O.synthetic_inputs(*shape)
    f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV)
    assert f.k.N == 1 << 35
    f._project(f.x)
    u0, v0 = f.k.u.cpu().clone(), f.k.v.cpu().clone()
    x, _, _ = f(10, return_numpy=False)
    rs_gpu = float(f.k.scal[1])
    a, g, alphas, rs = O.collapsed_cg(f.spc.cpu(), f.cmos.cpu(), u0, v0, f.x_shape, 4, 0.5, 0.5, 10)
    # x0 is deterministic: drop p and build it again next to x (there is no room for a copy during the solve)
    f._p = None
    f.k.ra = f.k.rb = f.k.zab = None
    torch.cuda.empty_cache()
    x0 = f._initialize()
    err = _gpu_rel_l2_vs_collapsed(x, x0, a, g, 4)
    print(f"{name} whole: N = 2^35, 10 iterations: rel-L2 vs fp64 collapsed CG = {err:.3e}; rsnew gpu {rs_gpu:.9e} "
          f"fp64 {rs[-1]:.9e}")
    assert err <= 5e-6, err
    assert abs(rs_gpu - rs[-1]) <= 1e-5 * rs[-1]
    del x, x0, f
    torch.cuda.empty_cache()


def _torch_adam_reference(x, spc_n, cmos_n, w1, w2, r, lr, steps, betas=(0.9, 0.999), eps=1e-8):
    """fusion.py:324-346 (loss -> autograd gradient -> torch.optim.Adam's update formulas) in plain fp32 torch on the
    GPU, one time bin at a time: the independent reference for volumes no eager run fits."""
    n_t, n_l, n_z, X, Y = x.shape
    m1, m2 = torch.zeros_like(x), torch.zeros_like(x)
    for step in range(1, steps + 1):
        u = torch.zeros((n_z, X, Y), dtype=torch.float64, device=x.device)
        v = torch.empty((n_t, n_l, X // r, Y // r), dtype=torch.float64, device=x.device)
        for t in range(n_t):
            xt = x[t].double()
            u += xt.sum(0)
            v[t] = xt.sum(1).reshape(n_l, X // r, r, Y // r, r).sum((2, 4))
        A = (-2.0 * w1) * (cmos_n - u.float())                                  # d loss / d x through T, S
        B = (-2.0 * w2) * (spc_n - v.float())                                   # ... through D, R
        bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
        for t in range(n_t):
            g = A[None] + B[t].repeat_interleave(r, 1).repeat_interleave(r, 2)[:, None]
            m1[t].lerp_(g, 1 - betas[0])
            m2[t].mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
            x[t].addcdiv_(m1[t], (m2[t].sqrt() / bc2 ** 0.5).add_(eps), value=-lr / bc1)
    return x


def test_fused_adam_on_a_2p32_element_slab_vs_plain_torch():
    """FusionAdam's fused step (fusion.py:324-346) on config 3's per-GPU slab, 3 steps: factored first moment vs the
    explicit one vs a plain-torch restatement run on the GPU."""
    from datafusion_amd import FusionAdam
    shape = SLABS_2P32["c3g8"]
    _need_free(7 * 16 * GIB, "Adam on the c3g8 slab")
    spc, cmos = O.synthetic_inputs(*shape)
    lr, steps = 2e-13, 3
    f = FusionAdam(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV)
    x0 = f.x.detach().clone()
    xf, spc_n, cmos_n = f(lr=lr, max_iterations=steps, return_numpy=False)
    xf = xf.detach()
    f._m2 = None
    fe = FusionAdam(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, moment1="explicit")
    xe, _, _ = fe(lr=lr, max_iterations=steps, return_numpy=False)
    xe = xe.detach()
    fe._m1 = fe._m2 = None
    torch.cuda.empty_cache()
    xr = _torch_adam_reference(x0.clone(), spc_n, cmos_n, 0.5, 0.5, 4, lr, steps)

    def rel(a, b):
        num = sum(float(((a[t].double() - b[t].double()) ** 2).sum()) for t in range(a.shape[0]))
        den = sum(float((b[t].double() ** 2).sum()) for t in range(a.shape[0]))
        return (num / den) ** 0.5

    e_x, e_fe = rel(xf, xr), rel(xf, xe)
    # the step itself (x - x0), so that the check cannot pass on x0 alone
    for t in range(x0.shape[0]):
        xf[t] -= x0[t]
        xr[t] -= x0[t]
    e_dx = rel(xf, xr)
    size = (sum(float((xr[t].double() ** 2).sum()) for t in range(x0.shape[0])) /
            sum(float((x0[t].double() ** 2).sum()) for t in range(x0.shape[0]))) ** 0.5
    print(f"Adam 2^32: rel-L2 x vs torch {e_x:.3e}, factored vs explicit m1 {e_fe:.3e}, step vs torch {e_dx:.3e} "
          f"(||step|| / ||x0|| = {size:.3e})")
    assert e_x <= 2e-6 and e_fe <= 2e-6
    assert 1e-4 <= size <= 0.5, size            # a visible step, not a no-op
    assert e_dx <= 2e-3, e_dx
    del xf, xe, xr, x0, f, fe
    torch.cuda.empty_cache()


def test_cells_like_mid_size_vs_oracle_with_tolerance():
    """Realistic distribution (second generator of SURVEY 8d), README-like call: tol=3e-3."""
    from datafusion_amd import FusionCG
    spc, cmos = O.cells_like_inputs(16, 8, 6, 64, 4, seed=4)
    w = {"spatial": 0.2, "spectro_temporal": 0.8}
    f = FusionCG(spc, cmos, weights=w, init_type="baseline", tol=3e-3, device=DEV)
    x, _, _ = f(40)
    pb = O.prepare(spc, cmos, weights=w, init_type="baseline", tol=3e-3)
    xo = O.cg(pb, 40, with_loss=False)
    assert f.curr_iter == pb.curr_iter
    assert O.rel_l2(x, xo) <= 2e-5


def _shard_worker(rank, world, port, q):
    sys.path[:0] = [ROOT]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from datafusion_amd import FusionCG
    from oracle import fusion_oracle as Or
    spc, cmos = Or.synthetic_inputs(6, 4, 12, 64, 4, seed=5)     # 12 planes -> 4 per rank (zc = 4)
    f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, group=dist.group.WORLD)
    x, _, _ = f(10, return_numpy=False)
    l1, l2 = f.loss()
    parts = [None] * world
    dist.all_gather_object(parts, (f.z_range, x.cpu().numpy()))
    if rank == 0:
        q.put((np.concatenate([p[1] for p in sorted(parts, key=lambda t: t[0])], axis=2), float(l1), float(l2)))
    dist.barrier()
    dist.destroy_process_group()


def test_z_sharded_hip_path_three_ranks_on_one_gpu():
    from datafusion_amd import FusionCG
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 3, 29631, q)) for r in range(3)]
    for p in procs:
        p.start()
    full, l1, l2 = q.get(timeout=110)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    spc, cmos = O.synthetic_inputs(6, 4, 12, 64, 4, seed=5)
    f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV)
    x, _, _ = f(10)
    assert O.rel_l2(full, x) <= 2e-6          # only the reduction order differs
    pb = O.prepare(spc, cmos, weights=W, init_type="baseline", tol=None)
    assert O.rel_l2(full, O.cg(pb, 10, with_loss=False)) <= 2e-5
    m1, m2 = f.loss()
    assert np.allclose([l1, l2], [float(m1), float(m2)], rtol=1e-3, atol=1e-16)


def _bit_checksum(x):
    """Exact checksum of a float32 tensor's bit patterns (int64 sum of the int32 views), one time bin at a time."""
    return int(sum(int(x[t].view(torch.int32).sum(dtype=torch.int64)) for t in range(x.shape[0])))


def _rccl_worker(port, q):
    sys.path[:0] = [ROOT]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DFX_FORCE_COMM="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    from datafusion_amd import FusionAdam, FusionCG
    from oracle import fusion_oracle as Or
    spc, cmos = Or.synthetic_inputs(6, 4, 8, 64, 4, seed=6)
    f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=3e-3, device=DEV, group=dist.group.WORLD)
    assert f.comm._always and f.comm.group is not None
    x, _, _ = f(10)
    it = f.curr_iter
    g = FusionAdam(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, group=dist.group.WORLD)
    xa, _, _ = g(lr=1e-7, max_iterations=3)
    probe = f.comm_probe                     # comm_slices="auto": one timed all-reduce of v decided the slicing
    del f, g
    torch.cuda.empty_cache()
    # sharded FusionAdam hands v over in bands (dfx_adam_step_chunks): a geometry with 4 chunks per workgroup, 3 bands
    # vs 1 band vs no group at all
    spc, cmos = Or.synthetic_inputs(32, 32, 4, 256, 4, seed=8)
    adam = []
    for kw in (dict(group=dist.group.WORLD, comm_slices=3), dict(group=dist.group.WORLD, comm_slices=1), dict()):
        for m1 in ("factored", "explicit"):
            a = FusionAdam(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, moment1=m1, **kw)
            xs, _, _ = a(lr=1e-9, max_iterations=3, return_numpy=False)
            adam.append((m1, kw.get("comm_slices", 0), a.k.plan.cpb, a.k._last_slices, _bit_checksum(xs.detach()),
                         float(a.tolerance())))
            del a, xs
    torch.cuda.empty_cache()
    # config 3's per-GPU slab at G = 8: its 256 MiB v all-reduced in 6 bands while the sweep goes on
    big = None
    if torch.cuda.mem_get_info()[0] > 40 * (1 << 30):
        spc, cmos = Or.synthetic_inputs(*SLABS_2P32["c3g8"])
        h = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV, group=dist.group.WORLD,
                     comm_slices=6)
        xb, _, _ = h(5, return_numpy=False)
        big = (_bit_checksum(xb), h.k.v.numel() * 4, h.comm_slices)
    q.put((x, it, xa, probe, big, adam))
    dist.destroy_process_group()


def test_collectives_through_real_rccl_one_rank():
    """Every all-reduce of the sharded path issued through backend nccl (= RCCL) on a one-rank group."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29641, q))
    p.start()
    x, it, xa, probe, big, adam = q.get(timeout=200)
    p.join(60)
    assert p.exitcode == 0
    for m1 in ("factored", "explicit"):
        rows = [r for r in adam if r[0] == m1]
        assert [r[1] for r in rows] == [3, 1, 0] and rows[0][2] >= 3 and [r[3] for r in rows] == [3, 1, 1], rows
        assert rows[0][4] == rows[1][4] == rows[2][4] and rows[0][5] == rows[1][5] == rows[2][5] > 0, rows
    assert probe is not None and 1 <= probe["slices"] <= 16 and probe["allreduce_v_s"] > 0, probe
    print("one-rank RCCL all-reduce probe:", probe)
    if big is not None:
        from datafusion_amd import FusionCG
        assert big[1] == 256 << 20 and big[2] == 6
        spc, cmos = O.synthetic_inputs(*SLABS_2P32["c3g8"])
        f = FusionCG(spc, cmos, weights=W, init_type="baseline", tol=None, device=DEV)
        xs, _, _ = f(5, return_numpy=False)
        assert _bit_checksum(xs) == big[0]        # slicing + a one-rank all-reduce change nothing, bit for bit
        del xs, f
        torch.cuda.empty_cache()
    spc, cmos = O.synthetic_inputs(6, 4, 8, 64, 4, seed=6)
    pb = O.prepare(spc, cmos, weights=W, init_type="baseline", tol=3e-3)
    xo = O.cg(pb, 10, with_loss=False)
    assert it == pb.curr_iter and O.rel_l2(x, xo) <= 2e-5
    pb = O.prepare(spc, cmos, weights=W, init_type="baseline", tol=None)
    assert O.rel_l2(xa, O.adam(pb, 1e-7, 3)) <= 2e-5


def test_readme_flow_on_a_synthetic_acquisition(tmp_path):
    """BASELINE.json configs[0] (the README example: load_data(npz, max_xy_size) -> FusionCG(tol=3e-3)(20)).
    The Zenodo file is not available offline: a cells-like acquisition of the same layout (utils.py:411-427:
    keys spc, cmos, time_axis, spectral_axis; cmos stored larger than max_xy_size) stands in for it, and the
    result is checked against the CPU oracle fed with the same loaded arrays."""
    from datafusion_amd import FusionCG, load_data
    spc, cmos = O.cells_like_inputs(12, 16, 6, 64, 4, seed=9)            # (t, l, 16, 16), (z, 64, 64)
    spc, cmos = np.asarray(spc, dtype=np.float32), np.asarray(cmos, dtype=np.float32)
    big = torch.nn.functional.interpolate(torch.from_numpy(cmos)[None], scale_factor=2, mode="bilinear",
                                          align_corners=False)[0]
    path = tmp_path / "cells_synth.npz"
    np.savez(path, spc=spc, cmos=big.numpy(), time_axis=np.linspace(0, 12.5, 12),
             spectral_axis=np.linspace(500, 700, 16))
    spc_l, cmos_l, t_ax, l_ax = load_data(path, max_xy_size=64, device=DEV)      # the resize runs on the GPU
    assert spc_l.shape == (12, 16, 16, 16) and cmos_l.shape == (6, 64, 64) and len(t_ax) == 12 and len(l_ax) == 16
    assert isinstance(cmos_l, np.ndarray) and cmos_l.dtype == np.float32        # a float32 stack stays float32
    _, cmos_h, _, _ = load_data(path, max_xy_size=64)                           # the same stage over scipy on the host
    assert O.rel_l2(cmos_l, cmos_h) <= 1e-6
    w = {"spatial": 0.5, "spectro_temporal": 0.5}
    f = FusionCG(spc_l, cmos_l, weights=w, init_type="baseline", tol=3e-3, mask_noise=False, total_energy=1,
                 device=DEV, seed=42)
    x, spc_n, cmos_n = f(max_iterations=20)
    pb = O.prepare(torch.from_numpy(spc_l.astype(np.float32)), torch.from_numpy(cmos_l.astype(np.float32)), weights=w,
                   init_type="baseline", tol=3e-3)
    xo = O.cg(pb, 20, with_loss=False)
    assert f.curr_iter == pb.curr_iter
    assert x.shape == (12, 16, 6, 64, 64) and O.rel_l2(x, xo) <= 2e-5
    assert O.rel_l2(spc_n, pb.spc) <= 1e-6 and O.rel_l2(cmos_n, pb.cmos) <= 1e-6
    # the slices the README plots afterwards (cells.ipynb): intensity image and global spectrum / decay
    assert O.rel_l2(f.marginals("tl").cpu(), torch.from_numpy(x).double().sum((0, 1))) <= 1e-6


@pytest.mark.parametrize("shape,size", [((5, 128, 128), 64), ((3, 96, 80), 32), ((2, 30, 30), 20), ((2, 16, 16), 32),
                                        ((2, 24, 24), 24), ((1, 100, 100), 33), ((4, 512, 512), 128), ((1, 264, 264), 8)])
def test_load_data_resize_on_device_vs_the_scipy_restatement(shape, size):
    """dfx_resize_stack (utils.py:420-422: Gaussian pre-filter + order-1 resize of the CMOS stack, with load_data's x/y
    swap) against the scipy.ndimage calls skimage.transform.resize is built from.  Parity vs scikit-image itself is
    unpinned (not installed); this pins the device kernels to the host restatement, in float64."""
    from datafusion_amd.io import resize_like_skimage, resize_stack_on_device
    rng = np.random.default_rng(3)
    stack = rng.random(shape)                                                   # (z, X, Y) float64, as stored
    ref = resize_like_skimage(np.transpose(stack, (1, 2, 0)), (size, size, shape[0]))
    ref = np.transpose(ref, (2, 1, 0))                                          # utils.py:422
    out = resize_stack_on_device(stack, size, DEV)
    assert out.dtype == torch.float64 and tuple(out.shape) == (shape[0], size, size)
    assert np.max(np.abs(out.cpu().numpy() - ref)) <= 1e-12 * np.max(np.abs(ref))
    plain = resize_stack_on_device(stack, size, DEV, transpose_out=False)       # without the swap
    assert torch.equal(plain.transpose(1, 2), out)
    o32 = resize_stack_on_device(stack.astype(np.float32), size, DEV)
    assert o32.dtype == torch.float32 and O.rel_l2(o32.cpu(), ref) <= 1e-6
    if shape[1] == size:                                                        # same size: the identity (up to the swap)
        assert np.array_equal(out.cpu().numpy(), np.transpose(stack, (0, 2, 1)))
    if shape[1] == 264:                     # shrinking 65-fold needs a Gaussian of radius 128: beyond the kernel's 64
        from datafusion_amd import _cabi as cabi
        with pytest.raises(cabi.DfxError, match="shrink factor too large"):
            resize_stack_on_device(rng.random((1, 260, 260)), 4, DEV)


def test_bench_prints_one_contract_line():
    """bench.py's stdout is exactly one JSON line with the fields the driver reads (tiny workload, child process)."""
    import json
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "3", "--warmup",
                        "1"], capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["unit"] == "it/s" and d["value"] > 0 and d["dtype"] == "f32" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    assert d["config"]["residual"] == "factored" and d["other_residual_mode"]["residual"] == "explicit"
    # the CPU leg's iterate is compared with the HIP path's on the same inputs; above 1e-4 bench.py exits non-zero
    assert d["parity"]["ok"] and d["parity"]["rel_l2_vs_cpu_oracle"] <= 1e-4 and d["parity"]["iterations"] == 2
    assert d["timed_blocks"]["repeats"] == 3 and len(d["timed_blocks"]["ms_per_step"]) == 3


def test_c_abi_without_python_or_torch():
    """tests/native/cabi_standalone.cpp (built by __graft_entry__.build()): the ABI driven from C++ with nothing but
    the HIP runtime - projection and a factored XP sweep checked against host loops."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "native", "cabi_standalone")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=100)
    assert p.returncode == 0 and "cabi_standalone: OK" in p.stdout, p.stdout + p.stderr
