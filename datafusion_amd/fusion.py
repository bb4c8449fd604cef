"""Host side of the MI355X FusionCG / FusionAdam path: the reference's Python surface
(datafusion/fusion.py:101-496: same class names, keyword arguments, defaults, return values and
exceptions) over the HIP kernels in libdfx.so.  torch is used for device memory, streams,
torch.distributed (RCCL) and a few operations on the *small* cmos/spc-sized arrays; every pass
over a 5-D vector is a hand-written kernel (see kernels.py / csrc/dfx_kernels.hip).

Differences from the reference a caller can observe (documented, deliberate):
  * device defaults to "cuda" and must be an AMD GPU - there is no CPU path;
  * self.x is updated in place (the reference rebinds a fresh tensor each iteration,
    fusion.py:445) and self.prev_x stays None: ||x - prev_x|| is |alpha| ||p|| exactly, so the
    clone of fusion.py:441 and the subtraction of fusion.py:250 are never materialised;
  * with tol=None the finiteness assertion of fusion.py:448 is raised after the loop (the
    device keeps a sticky flag) instead of at the offending iteration: no host sync per iteration;
  * loss() is not evaluated every CG iteration unless verbose=True (fusion.py:458 computes it
    only to feed the progress bar; it does not influence x);
  * `group=` (a torch.distributed process group) z-shards the volume: every rank passes the full
    (small) spc and cmos, owns planes z_range of x, and gets its own shard back;
  * FusionCG keeps the residual r = b - A x as its two small-space factors by default
    (`residual="factored"`; `"explicit"` stores it as a third 5-D vector, the literal fusion.py:446): same
    recurrence, same numbers, one 5-D pass per iteration instead of two;
  * return_numpy=True copies x to the host through pinned bounce buffers (hostcopy.py): same array, ~8x sooner;
  * tensor inputs are converted to float32 (the reference keeps a torch.Tensor's dtype, fusion.py:138-139, and so
    runs in float64 when handed float64 tensors; ndarrays are cast to float32 by both).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import _cabi as cabi
from .dist import ShardComm, z_partition
from . import hostcopy, kernels
from .kernels import HipKernels


def squared_l2(x):
    """fusion.py:12-18."""
    return (x ** 2).sum()


class SumOperator:
    """API mirror of fusion.py:21-51: T, S, D as stand-alone callables on device tensors (dfx_sum_axis /
    dfx_expand_axis).  The solvers never call these: the sums are fused into dfx_sweep and the normalised
    transposes are broadcast-on-read inside dfx_bcast."""

    def __init__(self, size: int = 1, integral_dim: int = 0):
        self.size = size
        self.integral_dim = integral_dim

    def T(self, y: torch.Tensor) -> torch.Tensor:
        return kernels.op_expand_axis(y, int(self.size), self.integral_dim)      # fusion.py:42

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        self.size = x.shape[self.integral_dim]  # fusion.py:50
        return kernels.op_sum_axis(x, self.integral_dim)                         # fusion.py:51


class SumPoolOperator:
    """API mirror of fusion.py:54-98 (R and its normalised transpose) on (nT, nL, 1, X, Y) device tensors
    (dfx_block_sum / dfx_block_expand)."""

    def __init__(self, size: int = 4, channels: int = 16, device: str = "cuda"):
        self.size = size
        self.channels = channels
        self._device = device
        self._kernel = None

    @property
    def kernel(self) -> torch.Tensor:
        """fusion.py:75: the depthwise ones kernel of the reference's conv2d / conv_transpose2d, for callers that read
        it (built on first use; dfx_block_sum / dfx_block_expand need no kernel tensor)."""
        if self._kernel is None:
            self._kernel = torch.ones((self.channels, 1, self.size, self.size), dtype=torch.float32, device=self._device)
        return self._kernel

    def T(self, y: torch.Tensor) -> torch.Tensor:
        return kernels.op_block_expand(y, int(self.size))                        # fusion.py:84-87

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return kernels.op_block_sum(x, int(self.size))                           # fusion.py:95-97


class Fusion:
    """Base class (fusion.py:101-292): input conversion, masks, energy normalisation, init."""

    _backend_cls = HipKernels  # tests swap in a CPU stand-in to exercise the sharding logic under gloo

    def __init__(self, spc, cmos, *, weights: dict, init_type: str, tol: float | None = 3e-3,
                 mask_noise: bool = False, total_energy: float = 1.0, device: str = "cuda", seed: int = 42,
                 verbose: bool = False, group=None, defer_x: bool = True, target_blocks: int = 0,
                 placement_tries: int = 6, comm_slices: int | str = "auto", placement: str = "rank"):
        dev = torch.device(device)
        self.spc = self._to_device(spc, "spc", dev)      # fusion.py:135-148
        self.cmos = self._to_device(cmos, "cmos", dev)
        self.verbose = verbose
        self.weights = weights
        self.init_type = init_type
        self.curr_iter = 0
        self.tol = tol
        self.seed = seed
        self.mask_noise = mask_noise
        self.device = device
        self.total_energy = float(total_energy)
        self.n_times, self.n_lambdas = int(self.spc.shape[0]), int(self.spc.shape[1])
        self.z_dim, self.xy_dim = int(self.cmos.shape[0]), int(self.cmos.shape[1])
        self.x_shape = (self.n_times, self.n_lambdas, self.z_dim, int(self.cmos.shape[1]), int(self.cmos.shape[2]))
        self.spatial_increase = int(self.cmos.shape[-1]) // int(self.spc.shape[-1])   # fusion.py:169
        self.prev_x = None
        r = self.spatial_increase
        if r < 1 or self.cmos.shape[1] != r * self.spc.shape[2] or self.cmos.shape[2] != r * self.spc.shape[3]:
            raise ValueError("cmos (z, x, y) must be an integer multiple r of spc (t, l, x/r, y/r) in x and y")
        self.defer_x = bool(defer_x) and not verbose
        self.placement_tries = int(placement_tries)   # HipKernels.new_vector_paired: large vectors only
        # launches a z-sharded 5-D pass is cut into so that the all-reduce of v overlaps it.  "auto": decided at the
        # start of a solve from one timed all-reduce of v; a number: that many launches (1 = one all-reduce after the
        # whole pass, no overlap - the setting to rule the asynchronous path out with)
        self._comm_slices_arg = comm_slices if comm_slices == "auto" else int(comm_slices)
        self.comm_slices = 1 if comm_slices == "auto" else int(comm_slices)
        self.comm_probe = None
        self.comm_timing = None   # bench.py: dict name -> list of (start, end) events around the waits for collectives

        self.T = SumOperator(size=self.n_times, integral_dim=0)        # fusion.py:172-179
        self.S = SumOperator(size=self.n_lambdas, integral_dim=1)
        self.D = SumOperator(size=self.z_dim, integral_dim=2)
        self.R = SumPoolOperator(size=r, channels=self.n_lambdas, device=device)

        # z-sharding (new): this rank owns planes [z0, z1)
        self.comm = ShardComm(group)
        self.z_range = z_partition(self.z_dim, self.comm.size, self.comm.rank)
        z0, z1 = self.z_range
        self.local_shape = (self.n_times, self.n_lambdas, z1 - z0, self.x_shape[3], self.x_shape[4])

        self.spc_mask, self.cmos_mask = self._get_masks()               # fusion.py:182
        if self.mask_noise:                                              # fusion.py:185-187
            self.spc = self.spc * self.spc_mask.float()
            self.cmos = self.cmos * self.cmos_mask.float()
        self.spc_mask = self.spc_mask.squeeze(0)                         # fusion.py:189
        self.spc = self.normalize_energy(self.spc, total_energy).contiguous()     # fusion.py:192-193
        self.cmos = self.normalize_energy(self.cmos, total_energy).contiguous()

        self.k = self._backend_cls(self.n_times, self.n_lambdas, z1 - z0, self.x_shape[3], self.x_shape[4], r, dev,
                                   target_blocks)
        # placement of the 5-D vectors in device memory: "rank" (default) lays out all vectors of a solve together by
        # measured memory class when they are >= 8 GiB each (placement.py); "probe": the second vector is chosen among
        # up to placement_tries torch allocations by timing (round 1 - 2 behaviour); "plain": torch allocations as they come
        if placement not in ("rank", "probe", "plain"):
            raise ValueError("placement must be 'rank', 'probe' or 'plain'")
        if placement == "plain":
            self.placement_tries = 1
        self.k.reserve_vectors(self._n_volumes(), placement, self._placement_min_bytes())
        self._cmos_loc = self.cmos[z0:z1].contiguous()
        self._cmask_loc = self.cmos_mask[z0:z1].float().contiguous()
        self._smask_f = self.spc_mask.float().contiguous()
        self.x = self._initialize()                                     # fusion.py:196-199

    # ---- helpers ------------------------------------------------------------------------
    @staticmethod
    def _to_device(a, name, dev):
        if isinstance(a, np.ndarray):
            return torch.from_numpy(a.astype(np.float32)).to(dev)
        if isinstance(a, torch.Tensor):
            return a.to(dev).float()
        raise TypeError(f"{name} must be a numpy.ndarray or torch.Tensor")

    @staticmethod
    def normalize_energy(tensor: torch.Tensor, total_energy: int | float = 1) -> torch.Tensor:
        """fusion.py:204-209 - for the small (spc / cmos sized) tensors; the 5-D x is normalised by
        dfx_sum + dfx_scale (or inside dfx_baseline_init)."""
        return total_energy * tensor / tensor.sum()

    def _get_masks(self):
        """fusion.py:280-292 on the small arrays."""
        cmos_mask = self.cmos > (0.05 * self.cmos.max())
        any_z = cmos_mask.any(dim=0).unsqueeze(0).float()
        size = [int(self.spc.shape[-2]), int(self.spc.shape[-1])]
        if list(any_z.shape[-2:]) != size:
            any_z = F.interpolate(any_z.unsqueeze(0), size=size, mode="bilinear", align_corners=False,
                                  antialias=False).squeeze(0)
        return any_z.bool(), cmos_mask

    def _n_volumes(self) -> int:
        """5-D vectors this solver sweeps in lock-step (x included); 0: not known here."""
        return 0

    def _placement_min_bytes(self):
        """Vector size from which the vectors are laid out by memory class; None: placement.MIN_VECTOR_BYTES."""
        return None

    def _initialize(self) -> torch.Tensor:
        """fusion.py:254-272 + the masking / normalisation of fusion.py:197-199, on the device."""
        k, z0, z1 = self.k, *self.z_range
        if self.init_type == "baseline":
            # baseline.py:46 cmos / sum_z cmos needs every plane: computed on the full (small) cmos
            zw = self.cmos / self.cmos.sum(dim=0, keepdim=True)
            if self.mask_noise:
                zw = torch.where(self.cmos_mask, zw, torch.zeros_like(zw))   # fusion.py:198
            zw = zw[z0:z1].contiguous()
            x = k.new_vector()
            k.baseline_init(self.spc, zw, None, 1.0, None)       # pass 1: SUM only, nothing written
            self.comm.sum_(k.scal[cabi.S_SUM:cabi.S_SUM + 1])
            k.recip(cabi.S_SCALE, cabi.S_SUM, self.total_energy)
            k.baseline_init(self.spc, zw, x, 1.0, cabi.S_SCALE)  # pass 2: x = E/sum * up(spc) * zw
            return x
        if self.init_type == "random":
            # the reference draws on `device`'s generator (fusion.py:262-264); parity is defined against
            # its torch-CPU run, so draw on the CPU generator and upload this rank's planes
            torch.manual_seed(self.seed)
            lo, hi = self.cmos.min().cpu(), self.cmos.max().cpu()
            full = (lo - hi) * torch.rand(self.x_shape) + hi
            x = k.new_vector()
            x.copy_(full[:, :, z0:z1])
        elif self.init_type == "zeros":
            torch.manual_seed(self.seed)
            x = k.new_vector().zero_()
        else:
            raise ValueError("Invalid initialization type.")                 # fusion.py:271
        if self.mask_noise:
            k.scale(x, self._cmask_loc, 1.0, None)                           # fusion.py:198
        k.sum(x)
        self.comm.sum_(k.scal[cabi.S_SUM:cabi.S_SUM + 1])
        k.recip(cabi.S_SCALE, cabi.S_SUM, self.total_energy)
        k.scale(x, None, 1.0, cabi.S_SCALE)                                  # fusion.py:199
        return x

    def _project(self, x) -> None:
        """u(x) -> k.u (local planes), v(x) -> k.v (summed over shards)."""
        self._uv_of = None
        self.k.sweep(cabi.SWEEP_PROJECT, x)
        self.comm.sum_(self.k.v)

    def _losses_from_uv(self):
        k = self.k
        w1, w2 = float(self.weights["spatial"]), float(self.weights["spectro_temporal"])
        k.sqdiff(self._cmos_loc, k.u, self._cmask_loc if self.mask_noise else None, w1, cabi.S_LOSS1)
        k.sqdiff(self.spc, k.v, self._smask_f if self.mask_noise else None, w2, cabi.S_LOSS2)
        self.comm.sum_(k.scal[cabi.S_LOSS1:cabi.S_LOSS1 + 1])
        return k.scal[cabi.S_LOSS1].float(), k.scal[cabi.S_LOSS2].float()

    SLICE_LAUNCH_COST_S = 50e-6   # measured, one-rank RCCL group at C2: 2.81 / 2.86 / 2.96 / 3.15 ms per iteration
                                  # for 1 / 2 / 4 / 8 launches (kernel tail + relaunch, the band's dfx_reduce_v, hand-over)

    def _auto_slices(self) -> int:
        """How many launches to cut a sharded sweep into.  Every band's all-reduce but the last hides behind the
        bands still being swept; bands shrink linearly (HipKernels._slice_bounds), so with k launches the exposed
        one carries 2/(k(k+1)) of v.  Each extra launch costs ~50 us on an MI355X.  The all-reduce is priced from a
        MEASUREMENT on this group, taken here: one all-reduce of the v buffer and one of 8 floats (latency), max over
        ranks -> T(bytes) = lat + bytes / bw.  Every rank computes the same k from the same two numbers."""
        if self.comm.size <= 1 and not self.comm._always:
            return 1
        if self._comm_slices_arg != "auto":
            return int(self._comm_slices_arg)
        if self.comm_probe is not None:
            return self.comm_probe["slices"]
        t_full, t_lat = self.comm.measure_allreduce(self.k.v)
        nbytes = 4.0 * self.k.v.numel()
        bw = nbytes / max(t_full - t_lat, 1e-7)
        cost = {k: t_lat + nbytes * 2.0 / (k * (k + 1)) / bw + self.SLICE_LAUNCH_COST_S * (k - 1)
                for k in range(1, cabi.MAX_SLICES + 1)}
        best = min(cost, key=cost.get)
        self.comm_probe = {"v_bytes": int(nbytes), "allreduce_v_s": t_full, "allreduce_8_floats_s": t_lat,
                           "busbw_GBps": nbytes / max(t_full - t_lat, 1e-7) / 1e9, "slices": best,
                           "modelled_exposed_s": cost[best]}
        return best

    def _comm_timed(self, name, fn):
        """Run `fn` (which makes the compute stream wait for collectives); when comm_timing is set, bracket it with
        events on the compute stream: their distance is the time the stream stood still, i.e. exposed communication."""
        if self.comm_timing is None:
            return fn()
        st = torch.cuda.current_stream(self.k.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        out = fn()
        e1.record(st)
        self.comm_timing.setdefault(name, []).append((e0, e1))
        return out

    def _save_solver_state(self):
        """Hook for solvers whose loop keeps live state in k.u / k.v / scal (FusionCG): called before a projection of
        x made from OUTSIDE the loop (loss(), marginals()); returns what _restore_solver_state needs, or None."""
        return None

    def _restore_solver_state(self, saved) -> None:
        pass

    def loss(self):
        """fusion.py:211-242 -> (spatial_loss, spectro_temporal_loss) as 0-dim device tensors.  Safe to call between
        two FusionCG.iterate() calls: the solver's projections and scalars are saved and put back."""
        saved = self._save_solver_state()
        x = self.x.detach() if isinstance(self.x, torch.Tensor) else self.x
        self._project(x)
        out = self._losses_from_uv()
        out = (out[0].clone(), out[1].clone())
        self._restore_solver_state(saved)
        return out

    def marginals(self, which: str = "tl") -> torch.Tensor:
        """Device-side reductions of the fused volume (what the notebooks compute with numpy after a
        4-128 GiB device-to-host copy, cells.ipynb cell 27bf9830e81df430), on this rank's z planes:
          "tl" -> x.sum(axis=(0, 1))  (z, x, y)   [= u(x)]
          "t"  -> x.sum(axis=0)       (l, z, x, y)
          "l"  -> x.sum(axis=1)       (t, z, x, y)
          "zxy"-> R(D(x))             (t, l, x/r, y/r), summed over shards  [= v(x)]"""
        if which in ("tl", "zxy"):
            saved = self._save_solver_state()
            self._project(self.x.detach())
            out = (self.k.u if which == "tl" else self.k.v).clone()
            self._restore_solver_state(saved)
            return out
        x = self.x.detach()
        if which in ("t", "l"):
            return self.k.sum_axis(x, 0 if which == "t" else 1)
        raise ValueError("which must be one of 'tl', 't', 'l', 'zxy'")

    def tolerance(self) -> torch.Tensor:
        """fusion.py:244-252: ||x - prev_x|| / ||x|| of the last completed iteration (device scalar)."""
        return self.k.scal[cabi.S_TOL].float()

    def _result(self, return_numpy: bool):
        x = self.x.detach() if isinstance(self.x, torch.Tensor) else self.x
        if return_numpy:                                                     # fusion.py:489-496
            return hostcopy.to_numpy(x), self.spc.cpu().numpy(), self.cmos.cpu().numpy()
        return self.x, self.spc, self.cmos

    def __call__(self, *args, **kwargs):
        raise NotImplementedError("This method should be implemented by subclasses.")


class FusionCG(Fusion):
    """Conjugate gradient on the normal equations (fusion.py:389-496)."""

    def __init__(self, *args, residual: str = "factored", **kwargs):
        """residual="factored" (default): r = b - A x is kept as the two small arrays it is made of,
        r = ra[z,x,y] + rb[t,l,x/r,y/r] (include/dfx.h, dfx_resid_update) - one 4N-word pass per iteration and
        two resident 5-D vectors (x, p).  residual="explicit": r is a third 5-D vector updated by dfx_bcast
        (2N words) and read by the sweep (5N words), the literal transcription of fusion.py:446."""
        if residual not in ("factored", "explicit"):
            raise ValueError("residual must be 'factored' or 'explicit'")
        self.residual = residual
        super().__init__(*args, **kwargs)
        self.w1 = self.weights["spatial"]
        self.w2 = self.weights["spectro_temporal"]
        self._r = self._p = None
        self._pending_x = False
        self._stopped = False

    def _save_solver_state(self):
        """loss() / marginals() project x into k.u, k.v and overwrite UU, PP, XX, XP - which, between two iterate()
        calls, hold u(p), v(p) and the norms the next iteration's alpha, residual update and tolerance are computed
        from.  Apply an x update still owed (loop left by tolerance), then save them."""
        if self._p is None:
            return None
        if self._pending_x:
            self.k.axpy(self.x, self._p, 1.0, cabi.S_ALPHA)
            self._pending_x = False
        k = self.k
        return k.u.clone(), k.v.clone(), k.scal[cabi.S_UU:cabi.S_XP + 1].clone()

    def _restore_solver_state(self, saved) -> None:
        if saved is not None:
            k = self.k
            k.u.copy_(saved[0]), k.v.copy_(saved[1]), k.scal[cabi.S_UU:cabi.S_XP + 1].copy_(saved[2])

    def _n_volumes(self) -> int:
        return 2 if self.residual == "factored" else 3      # x, p (, r)

    def _sweep(self, mode, eps: float | None = None) -> None:
        """One pass over x, (r,) p and everything that follows it: u(p), v(p), UU, VV, PP, XX, XP - and, when `eps` is
        given, alpha of the next iteration (fusion.py:444).
        One GPU: the 5-D launch + one dfx_post_sweep call (a reduction pass and a one-workgroup fold).
        z-sharded: the pass is cut into `comm_slices` launches along the slab axis and the all-reduce of each
        finished band of v (the RD partials, the one real exchange) is started on RCCL's stream while the
        next band is still being swept; u, UU and the scalar partials take one dfx_post_sweep launch, their all-reduce
        (4 doubles) follows the last band's, and ||v||^2 (+ alpha) a second one once v is complete."""
        k, c = self.k, self.comm
        fact = self.residual == "factored"
        c1, c2 = self._coefs()
        al = cabi.POST_ALPHA if eps is not None else 0
        e = 0.0 if eps is None else float(eps)
        if c.size > 1 or c._always:
            pending = []
            k.sweep(mode, self.x, self._r, self._p, slices=self.comm_slices,
                    after_slice=lambda rows: pending.append(c.sum_async(rows)), factored=fact, fold=False)
            k.post_sweep(cabi.POST_U | cabi.POST_FOLD_SWEEP)
            self._comm_timed("wait_v", lambda: c.wait(pending))
            self._comm_timed("scalars4", lambda: c.sum_(k.scal[cabi.S_UU:cabi.S_XP + 1]))   # UU, PP, XX, XP partials
            k.post_sweep(cabi.POST_V | cabi.POST_V_SUMMED | al, c1, c2, e)    # VV identical on every rank
        else:
            k.sweep(mode, self.x, self._r, self._p, factored=fact, fold=False)
            k.post_sweep(cabi.POST_U | cabi.POST_V | cabi.POST_FOLD_SWEEP | al, c1, c2, e)

    def _balance_gauge(self, du, dv, c1, c2):
        """The split r = ra + rb is not unique: a (x/r, y/r) array g can be moved between the two (ra + g, rb - g).
        Every update adds to the z/block-mean of ra and to the (t,l)-mean of rb in the fixed ratio w1 : w2, so a
        start whose means are not in that ratio leaves a constant of opposite sign in ra and rb that r = ra + rb
        has to cancel in fp32 for the rest of the solve.  Start with the means in the ratio w1 : w2: then both
        factors shrink with r.  (Small arrays, once per solve; summed over shards.)"""
        if not (c1 > 0 and c2 > 0):
            return du, dv
        r, n_z = self.spatial_increase, self.z_dim
        xs, ys = int(self.spc.shape[-2]), int(self.spc.shape[-1])
        a_sum = du.sum(0).reshape(xs, r, ys, r).sum((1, 3)).contiguous()      # this rank's planes
        self.comm.sum_(a_sum)
        a_g = a_sum * (c1 / (n_z * r * r))                                    # z/block-mean of ra
        b_g = dv.mean((0, 1)) * c2                                            # (t,l)-mean of rb
        w1, w2 = float(self.w1), float(self.w2)
        shift = (a_g + b_g) * (w1 / (w1 + w2)) - a_g                          # moved from rb into ra
        du = du + (shift / c1).repeat_interleave(r, 0).repeat_interleave(r, 1).unsqueeze(0)
        dv = dv - shift / c2
        return du.contiguous(), dv.contiguous()

    def _coefs(self):
        # A p = c1 * bcast_{t,l} u(p) + c2 * bcast_{z,block} v(p)   (fusion.py:423-426 with the 1/size
        # factors of the normalised transposes folded in)
        n_t, n_l, n_z, r = self.n_times, self.n_lambdas, self.z_dim, self.spatial_increase
        return float(self.w1) / (n_t * n_l), float(self.w2) / (n_z * r * r)

    def _placement_tries(self, planned_iterations) -> int:
        """Placing a large vector by measurement (HipKernels.new_vector_paired) costs 0.1 - 0.3 s of allocations and
        probe sweeps: worth it only when enough iterations follow.  Vectors >= 8 GiB gain ~19 % per iteration (probe
        from 30 planned iterations on, or when the count is unknown); 1 - 8 GiB vectors gain 1.5 - 3 % (from 300)."""
        big = 4 * self.k.N >= (8 << 30)
        if planned_iterations is None:
            return self.placement_tries if big else 1
        return self.placement_tries if planned_iterations >= (30 if big else 300) else 1

    @torch.no_grad()
    def start(self, planned_iterations: int | None = None) -> None:
        """r0 = b - A(x0), p0 = r0, rsold (fusion.py:428-437), from the current self.x."""
        k, c = self.k, self.comm
        c1, c2 = self._coefs()
        fact = self.residual == "factored"
        self.comm_slices = self._auto_slices()              # k.v is scratch here: it is overwritten by the projection below
        if self._p is None:
            self._p = k.new_vector_paired(self.x, self._placement_tries(planned_iterations))
            if fact:
                k.new_factored_residual()
            else:
                self._r = k.new_vector()
        x, rv, pv = self.x, self._r, self._p
        # r0 = c1 bcast(cmos_n - u(x0)) + c2 bcast(spc_n - v(x0))
        self._project(x)
        du = k.diff_scale(self._cmos_loc, k.u, None, 1.0)
        dv = k.diff_scale(self.spc, k.v, None, 1.0)
        if fact:
            du, dv = self._balance_gauge(du, dv, c1, c2)
            k.resid_update(du, dv, c1, c2, 1.0, None, False)
        else:
            k.bcast(rv, None, du, dv, None, c1, c2, 1.0, None)
        c.sum_(k.scal[cabi.S_RR:cabi.S_RR + 1])
        k.cg_start()                                                            # rsold (fusion.py:437)
        self._sweep(cabi.SWEEP_PINIT)                                           # p = r (fusion.py:436)
        self._pending_x = False
        self._stopped = False

    @torch.no_grad()
    def iterate(self, n_iterations: int, eps: float = 1e-10, history=None) -> None:
        """`n_iterations` passes of the loop body fusion.py:439-456 (stops early on tolerance).
        Enqueues kernels only; with tol=None and verbose=False it never synchronises."""
        k, c = self.k, self.comm
        c1, c2 = self._coefs()
        x, rv, pv = self.x, self._r, self._p
        deferred = self.defer_x
        one_gpu = not (c.size > 1 or c._always)
        if self._stopped or int(n_iterations) <= 0:
            # the loop was left through the tolerance test (fusion.py:451-453): RSOLD is already the new residual but
            # UU, VV still belong to the old p (the sweep was skipped), so ALPHA - which finish() may still owe to x -
            # must not be recomputed.  A stopped solve stays stopped until start() is called again.
            return
        # alpha of the first iteration of this call; every later one is computed by the launch that follows the sweep
        k.cg_alpha(c1, c2, eps)                                                 # fusion.py:443-444
        if history is not None and deferred:
            # a per-iteration loss needs the current x, and the two schedules cannot alternate (SWEEP_P leaves no
            # ||x||^2, <x,p> for a deferred tolerance): construct with defer_x=False (verbose=True does)
            raise ValueError("iterate(history=...) needs the textbook schedule: construct with defer_x=False")
        for i in range(int(n_iterations)):
            if self._stopped:
                break
            self.curr_iter += 1
            if not deferred:
                k.axpy(x, pv, 1.0, cabi.S_ALPHA)                                # fusion.py:445
                c.sum_(k.scal[cabi.S_XXNEW:cabi.S_XXNEW + 1])
            else:
                self._pending_x = True
            if self.residual == "factored" and one_gpu:                         # fusion.py:446-448, 455-456
                k.resid_cg_step(c1, c2, eps, deferred)                          # r update, rsnew, tol, beta: 3 launches
            else:
                if self.residual == "factored":                                 # fusion.py:446-447
                    k.resid_update(k.u, k.v, c1, c2, -1.0, cabi.S_ALPHA, True)
                else:
                    k.bcast(rv, rv, k.u, k.v, None, c1, c2, -1.0, cabi.S_ALPHA)
                self._comm_timed("scalar_rr", lambda: c.sum_(k.scal[cabi.S_RR:cabi.S_RR + 1]))
                k.cg_beta(eps, deferred)                                        # rsnew, tol, beta
            if self.tol is not None or self.verbose or history is not None:
                s = k.scal.cpu()                                                # one sync, as fusion.py:448
                assert bool(s[cabi.S_NONFINITE] == 0), "Conjugate gradient diverged"
                if self.tol is not None and float(s[cabi.S_TOL]) < self.tol:    # fusion.py:451-453
                    self._stopped = True
                    break
            if history is not None:                                             # fusion.py:458-478
                l1, l2 = self.loss()                                            # x is current: textbook schedule
                history[i] = [float(l1), float(l2), float(l1) + float(l2), float(s[cabi.S_RSNEW])]
            self._sweep(cabi.SWEEP_XP if deferred else cabi.SWEEP_P, eps)       # fusion.py:445,455 (+ next alpha)
            self._pending_x = False

    @torch.no_grad()
    def finish(self) -> None:
        """Apply an x update still owed (loop left by tolerance) and raise the assertion of
        fusion.py:448 if any residual norm was not finite."""
        k = self.k
        if self._pending_x:
            k.axpy(self.x, self._p, 1.0, cabi.S_ALPHA)
            self._pending_x = False
        s = k.scal.cpu()
        assert bool(s[cabi.S_NONFINITE] == 0), "Conjugate gradient diverged"   # fusion.py:448

    @torch.no_grad()
    def __call__(self, max_iterations: int, eps: float = 1e-10, return_numpy: bool = True):
        history = np.zeros((int(max_iterations), len(self.weights) + 2)) if self.verbose else None
        self.start(planned_iterations=int(max_iterations))
        self.iterate(max_iterations, eps, history)
        self.finish()
        self.history = history
        return self._result(return_numpy)


class _FusionLossFn(torch.autograd.Function):
    """loss() of fusion.py:211-242 as one autograd node over the HIP kernels: forward = dfx_sweep
    (PROJECT) + small-space residuals; backward = dfx_bcast writing
    g1 * (-2 w1) bcast_{t,l}(m1 (cmos - u)) + g2 * (-2 w2) bcast_{z,block}(m2 (spc - v))."""

    @staticmethod
    def forward(ctx, x, fusion, mask_grad):
        # k.u, k.v may already hold the projections of exactly this x: the optimizer step that produced it left them
        # (DfxAdam with a projector, dfx_adam_apply_project) and nothing has written x since (torch's version counter)
        if getattr(fusion, "_uv_of", None) != (x.data_ptr(), x._version):
            fusion._project(x.detach())
        k = fusion.k
        du = k.diff_scale(fusion._cmos_loc, k.u, fusion._cmask_loc if fusion.mask_noise else None, 1.0)
        dv = k.diff_scale(fusion.spc, k.v, fusion._smask_f if fusion.mask_noise else None, 1.0)
        ctx.fusion, ctx.mask_grad = fusion, mask_grad
        ctx.save_for_backward(du, dv)
        l1, l2 = fusion._losses_from_uv()
        return l1.clone(), l2.clone()

    @staticmethod
    def backward(ctx, g1, g2):
        f = ctx.fusion
        du, dv = ctx.saved_tensors
        w1, w2 = float(f.weights["spatial"]), float(f.weights["spectro_temporal"])
        A = ((-2.0 * w1) * g1.float()) * du
        B = ((-2.0 * w2) * g2.float()) * dv
        grad = f.k.new_vector()
        f.k.bcast(grad, None, A.contiguous(), B.contiguous(), f._cmask_loc if ctx.mask_grad else None,
                  1.0, 1.0, 1.0, None)
        return grad, None, None


class FusionAdam(Fusion):
    """Adam + backprop variant (fusion.py:295-386).

    fused=True (default): one dfx_adam_step pass per iteration does the gradient (broadcast on read,
    never stored), torch.optim.Adam's update, the optional clamp, ||x - prev_x||, ||x|| and the next
    iteration's projections: 6N words per iteration.  fused=False: the reference's structure -
    loss() as a custom torch.autograd.Function over the same HIP kernels, loss.backward(),
    torch.optim.Adam.step()."""

    def __init__(self, *args, fused: bool = True, betas=(0.9, 0.999), adam_eps: float = 1e-8,
                 moment1: str = "factored", project_in_step: bool = True, **kwargs):
        """moment1="factored" (default, fused=True only): exp_avg is linear in the gradients, which are all of the
        form mask * (A[z,x,y] + B[t,l,x/r,y/r]), so it is kept as two small arrays advanced by dfx_lerp and never
        stored as a 5-D vector: x and exp_avg_sq are the only volumes read and written (4N words per iteration).
        moment1="explicit": exp_avg is a 5-D vector as in torch.optim.Adam (6N words)."""
        if moment1 not in ("factored", "explicit"):
            raise ValueError("moment1 must be 'factored' or 'explicit'")
        self.fused = fused
        self.moment1 = moment1
        self.project_in_step = bool(project_in_step)     # fused=False only: see _run_autograd
        self._uv_of = None
        super().__init__(*args, **kwargs)
        self.betas, self.adam_eps = betas, adam_eps
        self._m1 = self._m2 = self._m1a = self._m1b = self._gA = self._gB = None
        self._adam_step = 0
        self._last_norms = None
        self._auto_bufs = None

    def tolerance(self) -> torch.Tensor:
        """fusion.py:244-252 as FusionAdam uses it (fusion.py:348): ||x - prev_x|| / ||x|| of the last step; prev_x is
        never materialised, the two squared norms come out of the step's own pass.  Summed over shards: with a
        process group every rank must call it (it is a collective).  Before the first step: 0."""
        if self._last_norms is None:
            return super().tolerance()
        d = self._last_norms.clone()
        self.comm.sum_(d)
        return torch.sqrt(d[0] / d[1]).float()

    def _n_volumes(self) -> int:
        # fused: x, exp_avg_sq (, exp_avg); the autograd formulation: x, exp_avg, exp_avg_sq (its gradient is a fresh
        # allocation of every backward(), wherever torch puts it)
        return (2 if self.moment1 == "factored" else 3) if self.fused else 3

    def _placement_min_bytes(self):
        # the FLAT optimizer step of the autograd formulation (dfx_adam_apply: seven streams over four arrays) gains from
        # spreading its arrays over the memory classes already at C2's 4 GiB (0.72 -> 0.78 of peak); the lane-mapped
        # projecting step that replaces it on fast plans (dfx_adam_apply_project) does not (C2: 4.87 ms placed against
        # 4.78 / 5.00 ms from torch, two runs each) and the 0.25 s of probes would cost more than 40 iterations; the
        # two-vector sweeps gain only from 8 GiB on (measured: tools/exp_c2_min_vector.sh)
        if self.fused or (self.project_in_step and getattr(getattr(self.k, "plan", None), "fast", 0)):
            return None
        return 1 << 30

    def loss_autograd(self, mask_grad: bool = False):
        return _FusionLossFn.apply(self.x, self, mask_grad)

    def __call__(self, lr: float, max_iterations: int, non_neg: bool = False, return_numpy: bool = True):
        history = np.zeros((max_iterations, len(self.weights) + 1)) if self.verbose else None
        self.x = torch.nn.Parameter(self.x, requires_grad=True)                 # fusion.py:324-325
        if self.fused:
            self._run_fused(lr, max_iterations, non_neg, history)
        else:
            self._run_autograd(lr, max_iterations, non_neg, history)
        self.history = history
        return self._result(return_numpy)

    @torch.no_grad()
    def _run_fused(self, lr, max_iterations, non_neg, history):
        k, c = self.k, self.comm
        w1, w2 = float(self.weights["spatial"]), float(self.weights["spectro_temporal"])
        x = self.x.data
        # a fresh optimizer per call, like fusion.py:325
        fact = self.moment1 == "factored"
        if self._m2 is None:                       # buffers are kept for later calls (state is reset below)
            big = 4 * k.N >= (8 << 30)      # the sweep-based probe did not help Adam below that (C2: 331 vs 331 it/s)
            self._m2 = k.new_vector_paired(x, self.placement_tries if big and max_iterations >= 30 else 1)
            self._m1 = None if fact else k.new_vector()
            self._m1a = torch.empty_like(k.u) if fact else None
            self._m1b = torch.empty_like(k.v) if fact else None
        for t in (self._m1, self._m1a, self._m1b, self._m2):
            if t is not None:
                t.zero_()
        self._adam_step = 0
        self.comm_slices = self._auto_slices()             # k.v is scratch here: the projection below overwrites it
        self._project(x)                                   # u(x), v(x) for the first loss / gradient
        cmask = self._cmask_loc if self.mask_noise else None
        smask = self._smask_f if self.mask_noise else None
        if self._gA is None:
            self._gA, self._gB = torch.empty_like(k.u), torch.empty_like(k.v)
        A, B = self._gA, self._gB
        sharded = c.size > 1 or c._always
        for i in range(max_iterations):
            self.curr_iter += 1
            self._adam_step += 1
            # one pass + fold: the losses (fusion.py:332), d loss / d u and d loss / d v (autograd of fusion.py:216-232) and
            # exp_avg.lerp_(grad, 1 - beta1) on the first moment's two factors - all on the small arrays
            k.adam_prep(self._cmos_loc, cmask, self.spc, smask, A, B, self._m1a, self._m1b, w1, w2,
                        1.0 - self.betas[0])
            if history is not None:
                c.sum_(k.scal[cabi.S_LOSS1:cabi.S_LOSS1 + 1])                   # the cmos term is a sum over shards
                l1, l2 = k.scal[cabi.S_LOSS1].float(), k.scal[cabi.S_LOSS2].float()
            # one 5-D pass (mask: fusion.py:274-278), then dfx_post_sweep for u(x), v(x), ||dx||^2, ||x||^2
            if sharded:
                # as FusionCG._sweep: the pass is cut into comm_slices launches and the all-reduce of each finished
                # band of v(x) (the RD partials of the next iteration's loss) runs on RCCL's stream under the bands
                # still being stepped; only the last, shortest band's is exposed
                pending = []
                k.adam_step(x, self._m1, self._m2, A, B, cmask, lr, self.betas[0], self.betas[1], self.adam_eps,
                            self._adam_step, non_neg, m1a=self._m1a, m1b=self._m1b, fold=False,
                            slices=self.comm_slices, after_slice=lambda rows: pending.append(c.sum_async(rows)))
                k.post_sweep(cabi.POST_U | cabi.POST_FOLD_ADAM)
                c.wait(pending)
            else:
                k.adam_step(x, self._m1, self._m2, A, B, cmask, lr, self.betas[0], self.betas[1], self.adam_eps,
                            self._adam_step, non_neg, m1a=self._m1a, m1b=self._m1b, fold=False)
                k.post_sweep(cabi.POST_U | cabi.POST_V | cabi.POST_FOLD_ADAM)
            self._last_norms = k.scal[cabi.S_ADAM_DD:cabi.S_ADAM_XX + 1]        # this shard's ||dx||^2, ||x||^2
            tolerance = None
            if self.tol is not None:                                            # fusion.py:348
                tolerance = float(self.tolerance())
            if tolerance is not None and tolerance < self.tol:                  # fusion.py:357-358
                break
            if history is not None:
                history[i] = [float(l1), float(l2), float(l1) + float(l2)]

    def _run_autograd(self, lr, max_iterations, non_neg, history):
        """fusion.py:324-358 as written there - loss() through autograd, an optimizer step on the explicit gradient -
        with the loss as one autograd node over the HIP kernels and torch.optim.Adam's update as one launch (DfxAdam:
        the clamp of fusion.py:344-346 and the norms of the tolerance are fused into it)."""
        if self._auto_bufs is None and self.k.reserved_left() >= 2:
            self._auto_bufs = (self.k.new_vector(), self.k.new_vector())       # laid out next to x by memory class
        bufs = {self.x: self._auto_bufs} if self._auto_bufs is not None else None
        # project_in_step: the optimizer step also leaves u(x), v(x) of the stepped volume (dfx_adam_apply_project), so
        # the next iteration's loss needs no projection pass of its own: 8N instead of 9N words per iteration
        proj = {self.x: self.k} if self.project_in_step else None
        optimizer = self.k.adam_optimizer([self.x], lr, self.betas, self.adam_eps, non_neg, bufs, proj)     # DfxAdam
        self._uv_of = None
        for i in range(max_iterations):
            self.curr_iter += 1
            optimizer.zero_grad()
            # the gradient mask of fusion.py:274-278 is applied inside the backward kernel
            spatial_loss, spectro_temporal_loss = self.loss_autograd(mask_grad=self.mask_noise)
            loss = spatial_loss + spectro_temporal_loss
            loss.backward()
            optimizer.step()                                                    # + clamp (fusion.py:344-346)
            if self.x in getattr(optimizer, "projected", ()):
                self.comm.sum_(self.k.v)                                        # the RD partials of the stepped volume
                self._uv_of = (self.x.data_ptr(), self.x._version)
            self._last_norms = optimizer.last_sq_norms                          # ||x - prev_x||^2, ||x||^2 (this shard)
            tolerance = None
            if self.tol is not None:                                            # fusion.py:348
                tolerance = float(self.tolerance())
            if tolerance is not None and tolerance < self.tol:
                break
            if history is not None:
                history[i] = [spatial_loss.item(), spectro_temporal_loss.item(), loss.item()]
