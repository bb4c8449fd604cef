"""HipKernels: the (only) compute backend - one z-shard's buffers + launches into libdfx.so.

Every method enqueues kernels on torch's current HIP stream of `device` and returns without
synchronising.  The CG scalars live in `self.scal` (double[NSCAL]) on the device.
The orchestration in fusion.py is written against this small interface so that its sharding /
collective logic can be exercised on CPU (gloo) with a stand-in backend supplied by the tests.
"""
from __future__ import annotations

import ctypes as C
import functools
import os

import torch

from . import _cabi as cabi


def _on_device(fn):
    """Run a launching method with the shard's GPU current (a stream of cuda:1 cannot be launched on while
    cuda:0 is the current device); a no-op when it already is."""
    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        if torch.cuda.current_device() == self._dev_index:
            return fn(self, *args, **kwargs)
        with torch.cuda.device(self._dev_index):
            return fn(self, *args, **kwargs)
    return wrapped


class HipKernels:
    def __init__(self, n_t: int, n_l: int, n_z_local: int, X: int, Y: int, r: int, device, target_blocks: int = 0,
                 zc: int = 0, steps: int = 0):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise cabi.DfxError(f"datafusion_amd runs on AMD GPUs only (got device={device!r}); there is no CPU path")
        self.lib = cabi.lib()
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", self._dev_index)
        self.shape = (n_t, n_l, n_z_local, X, Y)
        self.r = r
        self.M, self.S = n_t * n_l, n_z_local * X * Y
        self.N = self.M * self.S
        self.Xs, self.Ys = X // r, Y // r
        self.geom = cabi.Geom(n_t, n_l, n_z_local, X, Y, r, 0)
        self.plan = cabi.Plan()
        with torch.cuda.device(self._dev_index):   # the planner sizes the grid to the current device's CU count
            cabi.check(self.lib.dfx_plan_make_ex(C.byref(self.geom), target_blocks, zc, steps, C.byref(self.plan)),
                       "dfx_plan_make_ex")
        dev = self.device
        self.scal = torch.zeros(cabi.NSCAL, dtype=torch.float64, device=dev)
        self.part = torch.empty(self.plan.part_elems, dtype=torch.float64, device=dev)
        self.u_part = torch.empty(self.plan.u_part_elems, dtype=torch.float32, device=dev)
        self.v_part = torch.empty(self.plan.v_part_elems, dtype=torch.float32, device=dev)
        self.u = torch.empty((n_z_local, X, Y), dtype=torch.float32, device=dev)
        # with one layer in v_part (one z-chunk, no row parts) the sweep writes v directly: no reduce pass, v aliases v_part
        self.v_layers = self.plan.v_part_elems // (self.M * self.Xs * self.Ys)
        self._v_own = (torch.empty((n_t, n_l, self.Xs, self.Ys), dtype=torch.float32, device=dev)
                       if self.v_layers > 1 else None)
        self.v = self._v_own if self._v_own is not None else self.v_part.view(n_t, n_l, self.Xs, self.Ys)
        self.ra = self.rb = self.zab = None   # factored residual (new_factored_residual)
        self._last_slices = 1                 # launches the last sweep was cut into (post_sweep folds that many row sets)
        self._reserved = []                   # 5-D vectors laid out together by reserve_vectors(), handed out in order
        self.placement_info = None
        self.timing = None   # bench.py sets a dict: name -> list of (start, end) torch.cuda.Event pairs
        self.timing_names = None   # ... and, to time only some entry points (an event pair costs ~2 us of stream time), their names

    # -- helpers --------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _g(self):
        return C.byref(self.geom), C.byref(self.plan)

    def _timed(self, name, fn, *args):
        """Launch through `fn`; when self.timing is set, bracket it with HIP events on the launch stream."""
        if self.timing is None or (self.timing_names is not None and name not in self.timing_names):
            return fn(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(self.device))
        rc = fn(*args)
        e1.record(torch.cuda.current_stream(self.device))
        self.timing.setdefault(name, []).append((e0, e1))
        return rc

    def slot_ptr(self, slot: int) -> int:
        return self.scal.data_ptr() + 8 * slot

    def reserve_vectors(self, n: int, mode: str = "rank", min_bytes: int | None = None) -> None:
        """Called once by a solver, before it takes its first 5-D vector, with the number of 5-D vectors it will sweep
        in lock-step (x, p [, r]; x, exp_avg_sq [, exp_avg]).  mode "rank" and vectors of at least
        placement.MIN_VECTOR_BYTES (8 GiB): all n are laid out together so that at equal offsets no two share a class
        of device memory (datafusion_amd/placement.py; +16 % on the sweep of 16 GiB vectors) and new_vector() /
        new_vector_paired() hand them out in order.  Otherwise, or when the pool cannot serve the request, nothing is
        reserved and vectors come from torch (new_vector_paired then falls back to its probe)."""
        from . import placement
        self._reserved, self.placement_info = [], None
        if mode != "rank" or n < 2 or 4 * self.N < (placement.MIN_VECTOR_BYTES if min_bytes is None else min_bytes):
            return
        try:
            got = placement.pool(self.device).alloc_vectors(n, self.N)
        except (cabi.DfxError, torch.cuda.OutOfMemoryError) as e:
            # the layout is an optimisation: if the virtual-memory API is not usable here (driver, granularity, address
            # space) or memory ran out while classifying, give the pool's chunks back and take torch allocations
            placement.trim(self.device)
            self.placement_info = {"failed": f"{type(e).__name__}: {e}"}
            return
        if got is None:
            return
        vecs, self.placement_info = got
        self._reserved = [v.view(self.shape) for v in vecs]
        if self.plan.fast:
            # check the layout with the real pass on the real vectors (a no-op sweep over zeroed operands, as in
            # new_vector_paired): what the solver will get, recorded next to the layout
            had = self.ra is not None
            self.new_factored_residual()
            saved = (self.ra.clone(), self.rb.clone(), self.scal.clone()) if had else (None, None, self.scal.clone())
            self.ra.zero_(), self.rb.zero_()
            self.scal[cabi.S_ALPHA], self.scal[cabi.S_BETA] = 0.0, 0.0
            a, b = self._reserved[0].zero_(), self._reserved[1].zero_()
            best = 1e30
            for _ in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream(self.device))
                self.sweep(cabi.SWEEP_XP, a, None, b, factored=True)
                e1.record(torch.cuda.current_stream(self.device))
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1))
            self.placement_info["pair_sweep_GBps"] = 16.0 * self.N / best / 1e6
            self.scal.copy_(saved[2])
            if had:
                self.ra.copy_(saved[0]), self.rb.copy_(saved[1])
            else:
                self.ra = self.rb = self.zab = None

    def new_vector(self) -> torch.Tensor:
        if self._reserved:
            return self._reserved.pop(0)
        try:
            return torch.empty(self.shape, dtype=torch.float32, device=self.device)
        except torch.cuda.OutOfMemoryError:
            from . import placement
            placement.trim(self.device)       # classified chunks kept for a later solve: the memory is needed now
            return torch.empty(self.shape, dtype=torch.float32, device=self.device)

    def new_factored_residual(self) -> None:
        """Allocate the two factors of the CG residual, r = ra[z,x,y] + rb[t,l,x/r,y/r] (include/dfx.h), and the
        Xs*Ys workspace of dfx_resid_update."""
        if self.ra is None:
            n_t, n_l, n_z, X, Y = self.shape
            self.ra = torch.empty((n_z, X, Y), dtype=torch.float32, device=self.device)
            self.rb = torch.empty((n_t, n_l, self.Xs, self.Ys), dtype=torch.float32, device=self.device)
            self.zab = torch.empty((self.Xs, self.Ys), dtype=torch.float64, device=self.device)

    @staticmethod
    def _slice_bounds(cpb: int, k: int):
        """Cut cpb chunks into k non-empty bands of decreasing length (weights k, k-1, ..., 1): every band's
        all-reduce but the last overlaps the rest of the sweep, so the last one - the exposed one - is the shortest."""
        total = k * (k + 1) // 2
        bounds, acc = [0], 0
        for j in range(k):
            acc += k - j
            lo = bounds[-1] + 1                      # at least one chunk per band
            hi = cpb - (k - 1 - j)                   # leave one for every band still to come
            bounds.append(min(max(round(cpb * acc / total), lo), hi))
        bounds[-1] = cpb
        return bounds

    # vectors at least this large are placed with a probe (see new_vector_paired)
    PAIR_PROBE_MIN_BYTES = int(os.environ.get("DFX_PAIR_PROBE_MIN_BYTES", 1 << 30))
    PAIR_PROBE_BIG_BYTES = 8 << 30
    HBM_PEAK_BYTES_PER_S = 8.0e12      # MI355X (gfx950), the only target of this build

    def new_vector_paired(self, ref: torch.Tensor, tries: int = 6) -> torch.Tensor:
        """A new 5-D vector that streams well TOGETHER with `ref`.

        On MI355X two large vectors swept in lock-step run at 6.4 TB/s when they lie in different regions of device
        memory and at 5.3 TB/s when they share one (tools/placement_probe.py, DESIGN.md "placement"); which region an
        allocation lands in is the driver's choice.  So candidates are allocated (the rejected ones are held until
        the choice is made, otherwise the driver hands the same memory out again), each is timed against `ref` with
        the real pass run as a no-op, and the fastest is kept.
          * vectors >= 8 GiB: up to `tries` candidates, one after the other; stops as soon as both regimes have been
            seen or a candidate is clearly in the good one;
          * vectors of 1 - 8 GiB (C2: 4 GiB) are far less sensitive (6.38 TB/s right behind `ref`, 6.47 TB/s some tens of
            GiB further on): three candidates, each pushed 32 GiB further by a spacer allocation."""
        nbytes = 4 * self.N
        if self._reserved or tries <= 1 or nbytes < self.PAIR_PROBE_MIN_BYTES:
            return self.new_vector()           # reserved: already laid out against `ref` by reserve_vectors()
        big = nbytes >= self.PAIR_PROBE_BIG_BYTES
        if not big:
            tries = min(tries, 3)
        # the probe is the real pass: the factored XP sweep with alpha = beta = 0 over zeroed residual factors and a
        # zeroed candidate (x += 0 * 0 leaves ref as it is; the candidate stays zero)
        had_factors = self.ra is not None
        self.new_factored_residual()
        saved = (self.ra.clone(), self.rb.clone()) if had_factors else None
        saved_scal = self.scal.clone()
        self.ra.zero_(), self.rb.zero_()
        self.scal[cabi.S_ALPHA], self.scal[cabi.S_BETA] = 0.0, 0.0
        cands, rates, spacers = [], [], []
        for i in range(tries):
            free = torch.cuda.mem_get_info(self.device)[0]
            if i and free < nbytes + (2 << 30):
                break
            if i and not big:
                if free < nbytes + (34 << 30):
                    break
                spacers.append(torch.empty(32 << 30, dtype=torch.uint8, device=self.device))
            v = self.new_vector().zero_()
            best = 1e30
            for _ in range(2 if big else 3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream(self.device))
                self.sweep(cabi.SWEEP_XP, ref, None, v, factored=True)
                e1.record(torch.cuda.current_stream(self.device))
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1))
            cands.append(v)
            rates.append(4.0 * nbytes / best)
            # big vectors: done once both regimes have been seen, or when this candidate is clearly in the good one
            # (MI355X: ~6.3 TB/s vs ~5.3 TB/s, i.e. 0.79 vs 0.66 of the 8 TB/s peak; mixed placements fall in between)
            if big and (max(rates) >= 1.08 * min(rates) or rates[-1] * 1e3 >= 0.76 * self.HBM_PEAK_BYTES_PER_S):
                break
        self.scal.copy_(saved_scal)
        if saved is not None:
            self.ra.copy_(saved[0]), self.rb.copy_(saved[1])
        else:
            self.ra = self.rb = self.zab = None
        keep = cands[max(range(len(cands)), key=rates.__getitem__)]
        self.pair_probe = [r / 1e6 for r in rates]            # GB/s of every candidate, for logs and tests
        del cands, v, spacers
        torch.cuda.empty_cache()
        return keep

    def _chk(self, *ts):
        for t in ts:
            if t is not None:
                assert t.device == self.scal.device and t.dtype == torch.float32 and t.is_contiguous()

    # -- 5-D passes -----------------------------------------------------------------------
    @_on_device
    def sweep(self, mode: int, x, r=None, p=None, slices: int = 1, after_slice=None, factored: bool = False,
              fold: bool = True) -> None:
        """u(.) -> self.u (+UU), local v partial -> self.v, PP/XX/XP -> scal.
        factored=True: r is not a 5-D array but self.ra + self.rb (dfx_sweep_factored); pass r=None.
        slices > 1 cuts the pass along the slab axis into that many launches (fast plans; at most cpb);
        after_slice(v_rows) is called after each one with the rows of self.v (viewed (nT*nL, Xs*Ys)) that
        are final - the hook the z-sharded solver uses to start their all-reduce while the sweep goes on.
        fold=False: only the 5-D launch(es) (and, with after_slice, the per-band v sums) are issued; u, v, UU and
        PP/XX/XP are left to post_sweep() (one pass + one fold launch instead of seven launches)."""
        self._chk(x, r, p)
        for t in (x, r, p):
            assert t is None or t.numel() == self.N
        g, pl = self._g()
        s = self._stream()
        P = self.plan
        k = max(1, min(int(slices), P.cpb, cabi.MAX_SLICES)) if P.fast else 1
        gz = P.n_chunks // P.cpb
        bounds = self._slice_bounds(P.cpb, k)
        vrows = self.v.view(self.M, self.Xs * self.Ys)
        if factored:
            assert r is None and self.ra is not None and mode != cabi.SWEEP_PROJECT
        self._last_slices = k
        final = cabi.LAST_FOLD if fold else cabi.LAST_NOFOLD
        for j in range(k):
            c_lo, c_hi = bounds[j], bounds[j + 1]
            last = final if j == k - 1 else cabi.LAST_NO
            if factored:
                cabi.check(self._timed(f"sweep{mode}f", self.lib.dfx_sweep_factored, mode, g, pl, cabi.ptr(x),
                                       cabi.ptr(self.ra), cabi.ptr(self.rb), cabi.ptr(p), cabi.ptr(self.scal),
                                       cabi.ptr(self.u_part), cabi.ptr(self.v_part), cabi.ptr(self.part), c_lo, c_hi,
                                       j, last, s), "dfx_sweep_factored")
            else:
                cabi.check(self._timed(f"sweep{mode}", self.lib.dfx_sweep_chunks, mode, g, pl, cabi.ptr(x),
                                       cabi.ptr(r), cabi.ptr(p), cabi.ptr(self.scal), cabi.ptr(self.u_part),
                                       cabi.ptr(self.v_part), cabi.ptr(self.part), c_lo, c_hi, j, last, s),
                           "dfx_sweep_chunks")
            m_lo, m_hi = min(c_lo * P.steps * gz, self.M), min(c_hi * P.steps * gz, self.M)
            if m_hi <= m_lo or (not fold and after_slice is None):
                continue                                  # post_sweep() sums the z-chunks of v in its own launch
            if self._v_own is not None:
                cabi.check(self.lib.dfx_reduce_v(g, pl, cabi.ptr(self.v_part), cabi.ptr(self.v), m_lo, m_hi, s),
                           "dfx_reduce_v")
            if after_slice is not None:
                after_slice(vrows[m_lo:m_hi])
        if fold:
            cabi.check(self.lib.dfx_reduce_u(g, pl, cabi.ptr(self.u_part), cabi.ptr(self.u), cabi.ptr(self.scal),
                                             cabi.ptr(self.part), s), "dfx_reduce_u")

    @_on_device
    def post_sweep(self, flags: int, c1: float = 0.0, c2: float = 0.0, eps: float = 0.0) -> None:
        """One pass + one fold launch for what follows a sweep(fold=False) or adam_step(fold=False): u and UU (POST_U), v and VV
        (POST_V; with POST_V_SUMMED self.v is already final, e.g. all-reduced, and only VV is taken), the scalar
        partials of the 5-D launch (POST_FOLD_SWEEP / POST_FOLD_ADAM), alpha of the next iteration (POST_ALPHA)."""
        g, pl = self._g()
        summed = bool(flags & cabi.POST_V_SUMMED)
        cabi.check(self._timed("post", self.lib.dfx_post_sweep, g, pl, cabi.ptr(self.u_part), cabi.ptr(self.u),
                               cabi.ptr(self.v if summed else self.v_part),
                               None if summed or self._v_own is None else cabi.ptr(self._v_own),
                               cabi.ptr(self.scal), cabi.ptr(self.part), self._last_slices, flags,
                               c1, c2, eps, self._stream()), "dfx_post_sweep")

    @_on_device
    def adam_step(self, x, m1, m2, A, B, mask, lr, beta1, beta2, eps, step: int, non_neg: bool, m1a=None,
                  m1b=None, fold: bool = True, slices: int = 1, after_slice=None) -> None:
        """One fused Adam iteration: x, m2 (and m1) in place; u(x_new) -> self.u, local v partial -> self.v;
        ADAM_DD = ||dx||^2, ADAM_XX = ||x_new||^2.  m1=None: the first moment is mask * (m1a + m1b), already advanced
        by lerp() for this step (dfx_adam_step_factored).
        slices > 1 / after_slice: as in sweep() - the pass is cut along the slab axis (dfx_adam_step_chunks) and
        after_slice(v_rows) is called with each finished band of self.v, so that a z-sharded solve can start the
        band's all-reduce while the rest of the volume is still being stepped."""
        self._chk(x, m1, m2, A, B, mask, m1a, m1b)
        assert x.numel() == m2.numel() == self.N and (m1 is None or m1.numel() == self.N)
        assert A.numel() == self.S and B.numel() == self.M * self.Xs * self.Ys and (mask is None or mask.numel() == self.S)
        g, pl = self._g()
        s = self._stream()
        P = self.plan
        flags = (cabi.ADAM_NONNEG if non_neg else 0) | (0 if fold else cabi.ADAM_NOFOLD)
        k = max(1, min(int(slices), P.cpb, cabi.MAX_SLICES)) if P.fast else 1
        self._last_slices = k
        tail = (cabi.ptr(m2), cabi.ptr(A), cabi.ptr(B), cabi.ptr(mask), lr, beta1, beta2, eps, step, flags,
                cabi.ptr(self.scal), cabi.ptr(self.u_part), cabi.ptr(self.v_part), cabi.ptr(self.part))
        name = "adam" if m1 is not None else "adamf"
        if m1 is None:
            assert m1a.numel() == self.S and m1b.numel() == self.M * self.Xs * self.Ys
        if k == 1 and after_slice is None:
            if m1 is not None:
                cabi.check(self._timed(name, self.lib.dfx_adam_step, g, pl, cabi.ptr(x), cabi.ptr(m1), *tail, s),
                           "dfx_adam_step")
            else:
                cabi.check(self._timed(name, self.lib.dfx_adam_step_factored, g, pl, cabi.ptr(x), cabi.ptr(m1a),
                                       cabi.ptr(m1b), *tail, s), "dfx_adam_step_factored")
        else:
            gz = P.n_chunks // P.cpb
            bounds = self._slice_bounds(P.cpb, k)
            vrows = self.v.view(self.M, self.Xs * self.Ys)
            for j in range(k):
                c_lo, c_hi = bounds[j], bounds[j + 1]
                cabi.check(self._timed(name, self.lib.dfx_adam_step_chunks, g, pl, cabi.ptr(x), cabi.ptr(m1),
                                       cabi.ptr(m1a) if m1 is None else None, cabi.ptr(m1b) if m1 is None else None,
                                       *tail, c_lo, c_hi, j, s), "dfx_adam_step_chunks")
                m_lo, m_hi = min(c_lo * P.steps * gz, self.M), min(c_hi * P.steps * gz, self.M)
                if m_hi <= m_lo or after_slice is None:
                    continue
                if self._v_own is not None:
                    cabi.check(self.lib.dfx_reduce_v(g, pl, cabi.ptr(self.v_part), cabi.ptr(self.v), m_lo, m_hi, s),
                               "dfx_reduce_v")
                after_slice(vrows[m_lo:m_hi])
        if not fold:
            return                                        # post_sweep(POST_U | POST_V | POST_FOLD_ADAM) follows
        cabi.check(self.lib.dfx_reduce_u(g, pl, cabi.ptr(self.u_part), cabi.ptr(self.u), cabi.ptr(self.scal),
                                         cabi.ptr(self.part), s), "dfx_reduce_u")
        if self._v_own is not None and after_slice is None:
            cabi.check(self.lib.dfx_reduce_v(g, pl, cabi.ptr(self.v_part), cabi.ptr(self.v), 0, self.M, s),
                       "dfx_reduce_v")

    @_on_device
    def adam_prep(self, cmos, cmask, spc, smask, A, B, m1a, m1b, w1: float, w2: float, lerp_weight: float) -> None:
        """The small-space half of a fused Adam iteration in one pass + its fold: A, B (d loss / d u, d loss / d v from
        self.u, self.v), the lerp of the first moment's factors (m1a, m1b; None: skip), LOSS1, LOSS2."""
        self._chk(cmos, cmask, spc, smask, A, B, m1a, m1b)
        assert cmos.numel() == A.numel() == self.S and spc.numel() == B.numel() == self.M * self.Xs * self.Ys
        assert cmask is None or cmask.numel() == self.S
        assert smask is None or spc.numel() % smask.numel() == 0
        cabi.check(self._timed("prep", self.lib.dfx_adam_prep, cabi.ptr(self.u), cabi.ptr(cmos), cabi.ptr(cmask),
                               cabi.ptr(A), cabi.ptr(m1a), self.S, cabi.ptr(self.v), cabi.ptr(spc), cabi.ptr(smask),
                               0 if smask is None else smask.numel(), cabi.ptr(B), cabi.ptr(m1b), B.numel(), w1, w2,
                               lerp_weight, cabi.ptr(self.scal), cabi.ptr(self.part), self._stream()), "dfx_adam_prep")

    @_on_device
    def adam_apply_project(self, x, grad, m1, m2, lr, beta1, beta2, eps, step: int, non_neg: bool) -> bool:
        """torch.optim.Adam's step on the explicit gradient (what DfxAdam does) fused with the projections of the
        stepped x: u(x) -> self.u, local v partial -> self.v, ADAM_DD / ADAM_XX -> scal (dfx_adam_apply_project +
        dfx_post_sweep).  Lane-mapped plans only: returns False (nothing done) on a generic plan."""
        if not self.plan.fast:
            return False
        self._chk(x, grad, m1, m2)
        assert x.numel() == grad.numel() == m1.numel() == m2.numel() == self.N
        g, pl = self._g()
        self._last_slices = 1
        cabi.check(self._timed("apply_project", self.lib.dfx_adam_apply_project, g, pl, cabi.ptr(x), cabi.ptr(grad),
                               cabi.ptr(m1), cabi.ptr(m2), lr, beta1, beta2, eps, step,
                               cabi.ADAM_NONNEG if non_neg else 0, cabi.ptr(self.scal), cabi.ptr(self.u_part),
                               cabi.ptr(self.v_part), cabi.ptr(self.part), self._stream()), "dfx_adam_apply_project")
        self.post_sweep(cabi.POST_U | cabi.POST_V | cabi.POST_FOLD_ADAM)
        return True

    def reserved_left(self) -> int:
        return len(self._reserved)

    @staticmethod
    def adam_optimizer(params, lr, betas, eps, non_neg, moment_buffers=None, projectors=None):
        """The optimizer of the autograd formulation (fusion.py:325): torch.optim.Adam's update as one launch;
        moment_buffers: {parameter: (exp_avg, exp_avg_sq)} placed by the caller (reserve_vectors) or None;
        projectors: {parameter: HipKernels} - the step of that parameter also leaves u, v of the stepped volume."""
        from .optim import DfxAdam
        return DfxAdam(params, lr=lr, betas=betas, eps=eps, non_neg=non_neg, moment_buffers=moment_buffers,
                       projectors=projectors)

    @_on_device
    def lerp(self, m, g, weight: float) -> None:
        """m += weight * (g - m) on small arrays (torch's exp_avg.lerp_)."""
        self._chk(m, g)
        assert m.numel() == g.numel()
        cabi.check(self.lib.dfx_lerp(cabi.ptr(m), cabi.ptr(g), m.numel(), weight, self._stream()), "dfx_lerp")

    @_on_device
    def sqnorm_v(self) -> None:
        """VV = ||v||^2 (call after the cross-shard sum of v)."""
        cabi.check(self.lib.dfx_sqnorm(cabi.ptr(self.v), self.v.numel(), cabi.ptr(self.scal), cabi.S_VV,
                                       cabi.ptr(self.part), self._stream()), "dfx_sqnorm")

    @_on_device
    def bcast(self, out, inp, A, B, mask, sa: float, sb: float, coef_host: float, coef_slot: int | None) -> None:
        """out = inp + coef*mask*(sa*A[z,x,y] + sb*B[t,l,x/r,y/r]); RR = sum(out^2)."""
        self._chk(out, inp, A, B, mask)
        assert out.numel() == self.N and A.numel() == self.S and B.numel() == self.M * self.Xs * self.Ys
        assert inp is None or inp.numel() == self.N
        assert mask is None or mask.numel() == self.S
        g, pl = self._g()
        cabi.check(self._timed("bcast", self.lib.dfx_bcast, g, pl, cabi.ptr(out), cabi.ptr(inp), cabi.ptr(A),
                               cabi.ptr(B), cabi.ptr(mask), sa, sb, coef_host,
                               None if coef_slot is None else self.slot_ptr(coef_slot),
                               cabi.ptr(self.scal), cabi.ptr(self.part), self._stream()), "dfx_bcast")

    @_on_device
    def resid_update(self, A, B, sa: float, sb: float, coef_host: float, coef_slot: int | None,
                     accumulate: bool) -> None:
        """ra (+)= coef*sa*A ; rb (+)= coef*sb*B ; RR = ||ra (+) rb||^2 over the 5-D volume (local z planes)."""
        self._chk(A, B)
        assert self.ra is not None and A.numel() == self.S and B.numel() == self.M * self.Xs * self.Ys
        g, _ = self._g()
        cabi.check(self._timed("resid", self.lib.dfx_resid_update, g, cabi.ptr(self.ra), cabi.ptr(self.rb), cabi.ptr(A),
                               cabi.ptr(B), sa, sb, coef_host, None if coef_slot is None else self.slot_ptr(coef_slot),
                               int(accumulate), cabi.ptr(self.zab), cabi.ptr(self.scal), cabi.ptr(self.part),
                               self._stream()), "dfx_resid_update")

    @_on_device
    def resid_cg_step(self, c1: float, c2: float, eps: float, deferred: bool) -> None:
        """The small-space half of a CG iteration on one GPU, three launches: ra -= ALPHA c1 u ; rb -= ALPHA c2 v ; then
        RR and dfx_cg_beta's bookkeeping (RSNEW, TOL, BETA, ...) in the one-workgroup fold launch."""
        assert self.ra is not None
        g, _ = self._g()
        cabi.check(self._timed("resid", self.lib.dfx_resid_cg_step, g, cabi.ptr(self.ra), cabi.ptr(self.rb),
                               cabi.ptr(self.u), cabi.ptr(self.v), c1, c2, eps, int(deferred), cabi.ptr(self.zab),
                               cabi.ptr(self.scal), cabi.ptr(self.part), self._stream()), "dfx_resid_cg_step")

    @_on_device
    def resid_update_noalloc(self, A, B, sa: float, sb: float, coef_host: float, coef_slot: int | None,
                             accumulate: bool) -> None:
        """resid_update through the entry point that takes no workspace (include/dfx.h: for callers that cannot
        allocate; the scratch is a static device array inside libdfx.so).  Same kernels, same results."""
        self._chk(A, B)
        assert self.ra is not None and A.numel() == self.S and B.numel() == self.M * self.Xs * self.Ys
        g, _ = self._g()
        cabi.check(self.lib.dfx_resid_update_noalloc(g, cabi.ptr(self.ra), cabi.ptr(self.rb), cabi.ptr(A), cabi.ptr(B),
                                                     sa, sb, coef_host,
                                                     None if coef_slot is None else self.slot_ptr(coef_slot),
                                                     int(accumulate), cabi.ptr(self.scal), self._stream()),
                   "dfx_resid_update_noalloc")

    @_on_device
    def axpy(self, y, x, coef_host: float, coef_slot: int | None) -> None:
        """y += coef * x ; XXNEW = ||y||^2."""
        self._chk(y, x)
        assert y.numel() == x.numel()
        cabi.check(self.lib.dfx_axpy(cabi.ptr(y), cabi.ptr(x), y.numel(), coef_host,
                                     None if coef_slot is None else self.slot_ptr(coef_slot), cabi.ptr(self.scal),
                                     cabi.ptr(self.part), self._stream()), "dfx_axpy")

    @_on_device
    def scale(self, y, mask, coef_host: float, coef_slot: int | None) -> None:
        self._chk(y, mask)
        cabi.check(self.lib.dfx_scale(cabi.ptr(y), y.numel(), cabi.ptr(mask), 0 if mask is None else mask.numel(),
                                      coef_host, None if coef_slot is None else self.slot_ptr(coef_slot),
                                      self._stream()), "dfx_scale")

    @_on_device
    def diff_scale(self, a, b, mask, coef: float):
        """coef * (a - b) * mask as a new small tensor shaped like a (mask may cover the trailing axes only)."""
        self._chk(a, b, mask)
        assert a.numel() == b.numel() and (mask is None or a.numel() % mask.numel() == 0)
        out = torch.empty_like(a)
        cabi.check(self.lib.dfx_diff_scale(cabi.ptr(out), cabi.ptr(a), cabi.ptr(b), cabi.ptr(mask),
                                           0 if mask is None else mask.numel(), a.numel(), coef, self._stream()),
                   "dfx_diff_scale")
        return out

    @_on_device
    def sum(self, y) -> None:
        """SUM = sum(y)."""
        self._chk(y)
        cabi.check(self.lib.dfx_sum(cabi.ptr(y), y.numel(), cabi.ptr(self.scal), cabi.ptr(self.part), self._stream()),
                   "dfx_sum")

    @_on_device
    def sum_axis(self, x, axis: int):
        """x.sum(axis) for axis 0 (time) or 1 (lambda) of the local (t, l, z, x, y) shard -> new tensor."""
        self._chk(x)
        n_t, n_l = self.shape[0], self.shape[1]
        if axis == 0:
            A, R, B, shape = 1, n_t, n_l * self.S, self.shape[1:]
        elif axis == 1:
            A, R, B, shape = n_t, n_l, self.S, (n_t,) + tuple(self.shape[2:])
        else:
            raise ValueError("sum_axis: axis must be 0 (time) or 1 (lambda); (0, 1) is project()")
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        cabi.check(self.lib.dfx_sum_axis(cabi.ptr(x), cabi.ptr(out), A, R, B, self._stream()), "dfx_sum_axis")
        return out

    @_on_device
    def recip(self, dst_slot: int, src_slot: int, num: float) -> None:
        cabi.check(self.lib.dfx_recip(self.slot_ptr(dst_slot), self.slot_ptr(src_slot), num, self._stream()),
                   "dfx_recip")

    @_on_device
    def sqdiff(self, a, b, mask, w: float, slot: int) -> None:
        """scal[slot] = w * sum mask*(a-b)^2 (mask periodic over the flat index)."""
        self._chk(a, b, mask)
        assert a.numel() == b.numel()
        cabi.check(self.lib.dfx_sqdiff(cabi.ptr(a), cabi.ptr(b), cabi.ptr(mask), 0 if mask is None else mask.numel(),
                                       a.numel(), w, cabi.ptr(self.scal), slot, cabi.ptr(self.part), self._stream()),
                   "dfx_sqdiff")

    @_on_device
    def baseline_init(self, spc, zw, x, coef_host: float, coef_slot: int | None) -> None:
        """x = coef * up(spc) * zw (x may be None: SUM only)."""
        self._chk(spc, zw, x)
        assert spc.numel() == self.M * self.Xs * self.Ys and zw.numel() == self.S
        g, _ = self._g()
        cabi.check(self.lib.dfx_baseline_init(g, cabi.ptr(spc), cabi.ptr(zw), cabi.ptr(x), coef_host,
                                              None if coef_slot is None else self.slot_ptr(coef_slot),
                                              cabi.ptr(self.scal), cabi.ptr(self.part), self._stream()),
                   "dfx_baseline_init")

    # -- scalar bookkeeping ---------------------------------------------------------------
    @_on_device
    def cg_start(self) -> None:
        cabi.check(self.lib.dfx_cg_start(cabi.ptr(self.scal), self._stream()), "dfx_cg_start")

    @_on_device
    def cg_alpha(self, c1: float, c2: float, eps: float) -> None:
        cabi.check(self.lib.dfx_cg_alpha(cabi.ptr(self.scal), c1, c2, eps, self._stream()), "dfx_cg_alpha")

    @_on_device
    def cg_beta(self, eps: float, deferred: bool) -> None:
        cabi.check(self.lib.dfx_cg_beta(cabi.ptr(self.scal), eps, int(deferred), self._stream()), "dfx_cg_beta")


# ---- stand-alone operator launches (SumOperator / SumPoolOperator mirrors) -------------------------------------
def _prep(t: torch.Tensor) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda":
        raise cabi.DfxError("datafusion_amd operators run on AMD GPU tensors only; there is no CPU path")
    return t.detach().float().contiguous()


def _prod(xs) -> int:
    n = 1
    for v in xs:
        n *= int(v)
    return n


def op_sum_axis(x: torch.Tensor, dim: int) -> torch.Tensor:
    """x.sum(dim, keepdim=True) through dfx_sum_axis."""
    x = _prep(x)
    shape = list(x.shape)
    A, R, B = _prod(shape[:dim]), shape[dim], _prod(shape[dim + 1:])
    out = torch.empty(shape[:dim] + [1] + shape[dim + 1:], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        cabi.check(cabi.lib().dfx_sum_axis(cabi.ptr(x), cabi.ptr(out), A, R, B,
                                           torch.cuda.current_stream(x.device).cuda_stream), "dfx_sum_axis")
    return out


def op_expand_axis(y: torch.Tensor, size: int, dim: int) -> torch.Tensor:
    """y.repeat_interleave(size, dim) / size through dfx_expand_axis."""
    y = _prep(y)
    shape = list(y.shape)
    A, R0, B = _prod(shape[:dim]), shape[dim], _prod(shape[dim + 1:])
    # each of the R0 entries along dim is repeated `size` times: (A*R0, size, B)
    out = torch.empty(shape[:dim] + [R0 * size] + shape[dim + 1:], dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        cabi.check(cabi.lib().dfx_expand_axis(cabi.ptr(y), cabi.ptr(out), A * R0, size, B, 1.0 / size,
                                              torch.cuda.current_stream(y.device).cuda_stream), "dfx_expand_axis")
    return out


def op_block_sum(x: torch.Tensor, r: int) -> torch.Tensor:
    x = _prep(x)
    X, Y = int(x.shape[-2]), int(x.shape[-1])
    out = torch.empty(list(x.shape[:-2]) + [X // r, Y // r], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        cabi.check(cabi.lib().dfx_block_sum(cabi.ptr(x), cabi.ptr(out), _prod(x.shape[:-2]), X, Y, r,
                                            torch.cuda.current_stream(x.device).cuda_stream), "dfx_block_sum")
    return out


def op_block_expand(y: torch.Tensor, r: int) -> torch.Tensor:
    y = _prep(y)
    X, Y = int(y.shape[-2]) * r, int(y.shape[-1]) * r
    out = torch.empty(list(y.shape[:-2]) + [X, Y], dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        cabi.check(cabi.lib().dfx_block_expand(cabi.ptr(y), cabi.ptr(out), _prod(y.shape[:-2]), X, Y, r, 1.0 / (r * r),
                                               torch.cuda.current_stream(y.device).cuda_stream), "dfx_block_expand")
    return out
