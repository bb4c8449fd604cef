"""Random geometries through the Z-SHARDED path with the real kernels: three processes share cuda:0 (gloo group, staged
through the host - RCCL refuses several ranks on one device), groups of 2 and 3 ranks, uneven plane counts, every solver
variant (CG factored / explicit residual, deferred / textbook x update, banded sweeps; Adam fused with a factored /
explicit first moment, and the autograd formulation), masks, tolerance stops - each against the CPU oracle on the
gathered volume.  Not part of the test-suite (a minute); run on a GPU box after touching the sharded schedule, the band
bookkeeping or the planner.

    python tools/fuzz_sharded.py [--cases 24] [--seed 0]
"""
import argparse
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 3


def worker(rank, port, cases, seed, q, general_r=False):
    sys.path[:0] = [ROOT]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from datafusion_amd import FusionAdam, FusionCG
    from oracle import fusion_oracle as O
    groups = {2: dist.new_group([0, 1]), 3: dist.new_group([0, 1, 2])}
    rng = random.Random(seed)
    worst, failed = 0.0, []
    for case in range(cases):
        world = rng.choice([2, 3, 3])
        r = rng.choice([3, 5, 6, 7, 10, 12, 16, 16, 32] if general_r else [2, 4, 4, 4, 8, 3])
        xs = rng.randint(1, 48 // r)
        if general_r:
            xs = 4 * rng.randint(1, max(1, 24 // r))      # Y = xs r a multiple of 4: the strip tile (LaneMap<0>), the flat map for 16 / 32 (Y % 32 != 0 here)
        n_t, n_l = rng.randint(1, 10), rng.randint(1, 6)
        n_z = rng.randint(world, 17)
        xy = xs * r
        w = {"spatial": rng.choice([0.5, 0.2, 0.8]), "spectro_temporal": rng.choice([0.5, 0.8, 0.2])}
        mask = rng.random() < 0.3
        tol = rng.choice([None, None, 3e-3])
        iters = rng.randint(1, 7)
        slices = rng.choice([1, 2, 3, "auto"])
        gen = O.cells_like_inputs if rng.random() < 0.5 else O.synthetic_inputs
        spc, cmos = gen(n_t, n_l, n_z, xy, r, seed=1000 + case)
        lr = 1e-2 / (n_t * n_l * n_z * xy * xy)
        variants = [("cg", dict(residual=res, defer_x=d)) for res in ("factored", "explicit") for d in (True, False)]
        variants += [("adam", dict(moment1="factored")), ("adam", dict(moment1="explicit")), ("adam", dict(fused=False))]
        if rank >= world:
            continue                               # not a member of this case's group
        grp = groups[world]
        ref = {}
        diverged = [False]
        if rank == 0:
            try:
                pb = O.prepare(spc, cmos, weights=w, init_type="baseline", tol=tol, mask_noise=mask)
                ref["cg"] = (O.cg(pb, iters, with_loss=False).numpy(), pb.curr_iter)
                pb = O.prepare(spc, cmos, weights=w, init_type="baseline", tol=None, mask_noise=mask)
                ref["adam"] = (O.adam(pb, lr, 3, non_neg=mask).numpy(), 3)
            except AssertionError:
                diverged[0] = True          # fusion.py:448 on a degenerate input (e.g. a z-column masked to zero -> 0/0)
        dist.broadcast_object_list(diverged, src=0, group=grp)
        if diverged[0]:
            # the reference raises "Conjugate gradient diverged": so must every rank of the sharded product
            raised = 0
            for res in ("factored", "explicit"):
                try:
                    FusionCG(spc, cmos, weights=w, init_type="baseline", tol=tol, device="cuda:0", mask_noise=mask,
                             group=grp, comm_slices=slices, residual=res)(iters, return_numpy=False)
                except AssertionError:
                    raised += 1
            flags = [None] * world
            dist.all_gather_object(flags, raised, group=grp)
            if rank == 0:
                ok = all(v == 2 for v in flags)
                print(f"case {case}: {world} ranks, t{n_t} l{n_l} z{n_z} {xy}x{xy} r{r} mask={mask} tol={tol}: the oracle "
                      f"diverges (fusion.py:448); AssertionError on every rank, both residual modes: {ok}", flush=True)
                if not ok:
                    failed.append((case, "no AssertionError", flags))
            continue
        errs, fulls = {}, {}
        for kind, kw in variants:
            if kind == "cg":
                f = FusionCG(spc, cmos, weights=w, init_type="baseline", tol=tol, device="cuda:0", mask_noise=mask,
                             group=grp, comm_slices=slices, **kw)
                x, _, _ = f(iters, return_numpy=False)
            else:
                f = FusionAdam(spc, cmos, weights=w, init_type="baseline", tol=None, device="cuda:0", mask_noise=mask,
                               group=grp, comm_slices=slices, **kw)
                x, _, _ = f(lr=lr, max_iterations=3, non_neg=mask, return_numpy=False)
            parts = [None] * world
            dist.all_gather_object(parts, (f.z_range, x.detach().cpu().numpy(), f.curr_iter), group=grp)
            if rank == 0:
                full = np.concatenate([p[1] for p in sorted(parts, key=lambda t: t[0])], axis=2)
                name = kind + "/" + "/".join(f"{k}={v}" for k, v in kw.items())
                errs[name] = O.rel_l2(full, ref[kind][0])
                fulls[name] = full
                its = {p[2] for p in parts}
                if its != {ref[kind][1]}:
                    errs[name + " ITERATIONS"] = float("inf")
            del f, x
        if rank == 0:
            bad = {k: v for k, v in errs.items() if not v <= 2e-5}
            worst = max(worst, max(errs.values()))
            print(f"case {case}: {world} ranks, t{n_t} l{n_l} z{n_z} {xy}x{xy} r{r} mask={mask} tol={tol} iters={iters}"
                  f"->{ref['cg'][1]} slices={slices} max err {max(errs.values()):.2e}" + (f"  FAIL {bad}" if bad else ""),
                  flush=True)
            if bad:
                # above the tight bar: the same call unsharded (this rank alone) tells what the distance is.  Adam turns
                # a gradient that is zero up to rounding into a step of +-lr, so on a small volume HIP and the oracle can
                # differ by a few 1e-5 with or without shards; a SHARDING error shows as sharded != unsharded
                real = {}
                for kind, kw in variants:
                    name = kind + "/" + "/".join(f"{k}={v}" for k, v in kw.items())
                    if name not in bad:
                        continue
                    if kind == "cg":
                        x1, _, _ = FusionCG(spc, cmos, weights=w, init_type="baseline", tol=tol, device="cuda:0",
                                            mask_noise=mask, **kw)(iters)
                    else:
                        x1, _, _ = FusionAdam(spc, cmos, weights=w, init_type="baseline", tol=None, device="cuda:0",
                                              mask_noise=mask, **kw)(lr=lr, max_iterations=3, non_neg=mask)
                    e1, e2 = O.rel_l2(x1, ref[kind][0]), O.rel_l2(fulls[name], x1)
                    line = f"      {name}: unsharded vs oracle {e1:.3e}, sharded vs unsharded {e2:.3e}"
                    if kind == "adam":
                        # an element whose gradient is of the size of Adam's eps takes a step anywhere in +-lr depending on
                        # the rounding of v's partial sums, which differ between 1 and k shards (seed 22 --general-r, case
                        # 1: 255 of 69 120 elements, each off by <= 0.07 lr): such elements - at most 1 in 100, none
                        # further apart than the three steps can take it - are set aside, everything else must agree to 2e-6
                        a_, b_ = torch.as_tensor(fulls[name]).double().cpu(), torch.as_tensor(x1).double().cpu()
                        out = (a_ - b_).abs() > 1e-5 * b_.abs().max()
                        frac = float(out.double().mean())
                        if frac <= 1e-2 and float((a_ - b_).abs().max()) <= 3.003 * lr:
                            e2 = float(((a_ - b_)[~out].norm() / b_[~out].norm()))
                            line += f"; without the {int(out.sum())} eps-gradient elements {e2:.3e}"
                    print(line, flush=True)
                    # north_star's bar, and the shards add nothing.  A single spc pixel (xs = 1) is degenerate for
                    # Adam: the baseline start satisfies the spc term exactly, its gradient is pure rounding noise and
                    # Adam steps +-lr on noise - the reference's own iterate is noise there; only the shards are judged
                    if not ((bad[name] <= 1e-4 or (kind == "adam" and xs == 1)) and e2 <= 2e-6):
                        real[name] = (bad[name], e1, e2)
                if real:
                    failed.append((case, real))
    dist.barrier()
    if rank == 0:
        print(f"{cases} cases, worst rel-L2 vs the oracle {worst:.3e}, failed: {failed}", flush=True)
        q.put(failed)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=24)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--port", type=int, default=29671)
    ap.add_argument("--general-r", action="store_true", help="r in {3, 5, 6, 7, 10, 12, 16, 32} with Y % 4 == 0 only")
    a = ap.parse_args()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, a.port, a.cases, a.seed, q, a.general_r)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join()
    codes = [p.exitcode for p in procs]
    failed = q.get(timeout=5) if all(c == 0 for c in codes) else [("exit codes", codes)]
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
