"""z-sharded host logic under torch.distributed (gloo, world_size 2) on the CPU stand-in backend:
the sharded result must equal the single-process oracle on the same inputs."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, case, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from datafusion_amd import FusionAdam, FusionCG, fusion
    from cpu_backend import CpuStandIn
    from oracle import fusion_oracle as O
    fusion.Fusion._backend_cls = CpuStandIn
    w = {"spatial": 0.3, "spectro_temporal": 0.7}
    spc, cmos = O.cells_like_inputs(6, 3, 5, 16, 4, seed=1)   # 5 planes over 2 ranks: uneven 3 + 2
    kw = dict(weights=w, init_type=case["init"], tol=case["tol"], mask_noise=case["mask"], device="cpu",
              group=dist.group.WORLD, defer_x=case["defer"])
    if case["method"] == "cg":
        f = FusionCG(spc, cmos, residual=case.get("residual", "factored"), comm_slices=case.get("slices", "auto"), **kw)
        x, _, _ = f(case["iters"], return_numpy=False)
        if case.get("slices", "auto") == "auto":
            # the slicing of the sharded sweep comes from one all-reduce timed on this group, and every rank must
            # have drawn the same number from it (the bands are collectives: their sizes have to match)
            assert f.comm_probe is not None and f.comm_probe["v_bytes"] == 4 * f.k.v.numel()
            ks = [None] * world
            dist.all_gather_object(ks, f.comm_slices)
            assert len(set(ks)) == 1 and 1 <= ks[0] <= 16
        else:
            assert f.comm_probe is None and f.comm_slices == case["slices"]
    else:
        f = FusionAdam(spc, cmos, fused=case.get("fused", True), moment1=case.get("moment1", "factored"),
                       comm_slices=case.get("slices", "auto"), **kw)
        x, _, _ = f(lr=1e-7, max_iterations=case["iters"], return_numpy=False)
        if case.get("fused", True):
            assert f.comm_slices == (case["slices"] if "slices" in case else f.comm_probe["slices"])
        # the public tolerance() after an Adam call is ||x - prev_x|| / ||x|| of the last step over ALL shards (a
        # collective): every rank gets the same number
        tol = float(f.tolerance())
        tols = [None] * world
        dist.all_gather_object(tols, tol)
        assert tol > 0 and len(set(tols)) == 1
    l1, l2 = f.loss()
    parts = [None] * world
    dist.all_gather_object(parts, (f.z_range, x.detach().numpy()))
    if rank == 0:
        full = np.concatenate([p[1] for p in sorted(parts, key=lambda t: t[0])], axis=2)
        q.put((full, f.curr_iter, float(l1), float(l2)))
    dist.barrier()
    dist.destroy_process_group()


def _run(case, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = case.get("world", 2)
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return out


def _oracle(case):
    sys.path[:0] = [ROOT]
    from oracle import fusion_oracle as O
    w = {"spatial": 0.3, "spectro_temporal": 0.7}
    spc, cmos = O.cells_like_inputs(6, 3, 5, 16, 4, seed=1)
    pb = O.prepare(spc, cmos, weights=w, init_type=case["init"], tol=case["tol"], mask_noise=case["mask"])
    x = O.cg(pb, case["iters"]) if case["method"] == "cg" else O.adam(pb, 1e-7, case["iters"])
    l1, l2 = O.loss(pb)
    return O, x, pb.curr_iter, float(l1), float(l2)


def _check(case, port):
    full, it, l1, l2 = _run(case, port)
    O, x, it_o, l1o, l2o = _oracle(case)
    assert it == it_o
    assert O.rel_l2(full, x) <= 2e-5
    assert np.allclose([l1, l2], [l1o, l2o], rtol=2e-3, atol=1e-14)


def test_gloo_cg_deferred_masked():
    _check(dict(method="cg", init="baseline", tol=None, mask=True, defer=True, iters=8), 29611)


def test_gloo_cg_textbook_tolerance_break_random_init():
    _check(dict(method="cg", init="random", tol=3e-3, mask=False, defer=False, iters=30), 29612)


def test_gloo_cg_explicit_residual():
    _check(dict(method="cg", init="baseline", tol=None, mask=False, defer=True, iters=8, residual="explicit"), 29615)


def test_gloo_cg_four_ranks_uneven_planes():
    """5 planes over 4 ranks: 2 + 1 + 1 + 1."""
    _check(dict(method="cg", init="baseline", tol=3e-3, mask=False, defer=True, iters=12, world=4), 29617)


def test_gloo_adam_fused():
    _check(dict(method="adam", init="baseline", tol=None, mask=True, defer=True, iters=4), 29613)


def test_gloo_adam_fused_explicit_first_moment():
    _check(dict(method="adam", init="baseline", tol=None, mask=False, defer=True, iters=4, moment1="explicit"), 29616)


def test_gloo_adam_autograd():
    _check(dict(method="adam", init="baseline", tol=1e-12, mask=False, defer=True, iters=4, fused=False), 29614)


def test_gloo_adam_fixed_slicing_three_bands_and_one_band():
    """Sharded FusionAdam hands v over in bands like FusionCG (the all-reduce of each band overlaps the rest of the
    step): 3 bands and 1 band give the same iterate, bit for bit, and the oracle's."""
    a = _run(dict(method="adam", init="baseline", tol=None, mask=True, defer=True, iters=4, slices=3), 29620)
    b = _run(dict(method="adam", init="baseline", tol=None, mask=True, defer=True, iters=4, slices=1), 29621)
    assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
    _check(dict(method="adam", init="baseline", tol=None, mask=False, defer=True, iters=4, slices=3, moment1="explicit"), 29622)


def test_gloo_cg_fixed_slicing_three_bands_and_one_band():
    """comm_slices given as a number: no probe; 1 = a single all-reduce after the whole sweep (the setting that rules
    the overlapped path out), 3 = three bands."""
    _check(dict(method="cg", init="baseline", tol=None, mask=False, defer=True, iters=6, slices=3), 29618)
    _check(dict(method="cg", init="baseline", tol=None, mask=False, defer=True, iters=6, slices=1), 29619)
