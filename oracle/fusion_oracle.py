"""CPU oracle for the FusionCG / FusionAdam / baseline() hot path of FluoLab/datafusion.

TEST INFRASTRUCTURE ONLY.  Nothing under ``datafusion_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and
there only as the checker / the reported CPU baseline, never as the thing shipped.

It is an independent torch-CPU restatement of the reference algorithm (layout (t, lambda, z, x, y),
fp32) that performs the *same ATen op sequence* as the reference, so that on the same torch build
it reproduces the reference bit for bit (checked by ``tools/make_golden.py`` in the build
container, where /root/reference is importable, and pinned by ``tests/golden/*.npz``).

Parity status: the reference ships no tests, fixtures or known-answer vectors for this path
(SURVEY.md section 4), and torchvision is absent from the image, so the goldens are
"reference code + torch 2.10 F.interpolate standing in for torchvision.transforms.Resize".
Against the reference's *own* test-suite the parity is therefore unpinned; against the reference
code run in the build container it is pinned by the committed vectors.

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Linear operators (datafusion/fusion.py:21-98)
# --------------------------------------------------------------------------------------


def sum_keep(x: torch.Tensor, dim: int) -> torch.Tensor:
    """T, S, D forward: sum over one axis, keepdim (fusion.py:44-51)."""
    return x.sum(dim=dim, keepdim=True)


def spread_mean(y: torch.Tensor, size: int, dim: int) -> torch.Tensor:
    """T.T, S.T, D.T: replicate `size` times along `dim`, divide by `size` (fusion.py:36-42).
    NB: the adjoint divided by `size` - part of the contract."""
    return y.repeat_interleave(size, dim=dim) / size


def block_sum(x: torch.Tensor, r: int) -> torch.Tensor:
    """R forward (fusion.py:89-98): depthwise ones r x r conv, stride r, on (nT, nL, 1, X, Y)."""
    n_l = x.shape[1]
    k = torch.ones(n_l, 1, r, r, dtype=x.dtype)
    return F.conv2d(x.squeeze(2), k, stride=r, groups=n_l, bias=None).unsqueeze(2)


def block_spread(y: torch.Tensor, r: int) -> torch.Tensor:
    """R.T (fusion.py:78-87): depthwise transposed conv (block replicate) divided by r^2."""
    n_l = y.shape[1]
    k = torch.ones(n_l, 1, r, r, dtype=y.dtype)
    return F.conv_transpose2d(y.squeeze(2), k, stride=r, groups=n_l, bias=None).unsqueeze(2) / r**2


def normalize_energy(t: torch.Tensor, total_energy: float = 1.0) -> torch.Tensor:
    """fusion.py:204-209."""
    return total_energy * t / t.sum()


# --------------------------------------------------------------------------------------
# baseline() initialiser (datafusion/baseline.py:8-52)
# --------------------------------------------------------------------------------------


def upsample_bilinear(img: torch.Tensor, size: tuple[int, int], antialias: bool) -> torch.Tensor:
    """What torchvision's tensor `resize` does (baseline.py:40-44, fusion.py:286-291):
    F.interpolate(bilinear, align_corners=False, antialias=...), with the same-size early-out."""
    if tuple(img.shape[-2:]) == tuple(size):
        return img
    squeeze = img.ndim < 4
    if squeeze:
        img = img.unsqueeze(0)
    out = F.interpolate(img, size=list(size), mode="bilinear", align_corners=False, antialias=antialias)
    return out.squeeze(0) if squeeze else out


def baseline_init(cmos: torch.Tensor, spc: torch.Tensor) -> torch.Tensor:
    """x[t, :, z] = upsample(spc[t]) * cmos[z] / sum_z cmos   (baseline.py:38-50).
    The reference allocates (nT, nL, nZ, X, X): a square field of view is assumed (baseline.py:38)."""
    n_t, n_l = spc.shape[0], spc.shape[1]
    n_z, xy = cmos.shape[0], cmos.shape[1]
    x = torch.zeros((n_t, n_l, n_z, xy, xy))
    zfrac = cmos / cmos.sum(dim=0, keepdim=True)  # baseline.py:46
    for t in range(n_t):  # baseline.py:48-50
        up = upsample_bilinear(spc[t], (cmos.shape[-2], cmos.shape[-1]), antialias=True)
        for z in range(n_z):
            x[t, :, z, :, :] = up * zfrac[z]
    return x


# --------------------------------------------------------------------------------------
# Problem set-up (Fusion.__init__, fusion.py:108-199)
# --------------------------------------------------------------------------------------


@dataclass
class Problem:
    spc: torch.Tensor          # normalised, (nT, nL, Xs, Ys)
    cmos: torch.Tensor         # normalised, (nZ, X, Y)
    x: torch.Tensor            # current iterate, (nT, nL, nZ, X, Y)
    spc_mask: torch.Tensor     # bool (Xs, Ys)          (fusion.py:189)
    cmos_mask: torch.Tensor    # bool (nZ, X, Y)
    w1: float
    w2: float
    r: int
    mask_noise: bool
    tol: float | None
    curr_iter: int = 0
    trace: list = field(default_factory=list)

    @property
    def dims(self):
        return tuple(self.x.shape)


def get_masks(spc: torch.Tensor, cmos: torch.Tensor):
    """fusion.py:280-292: cmos_mask = cmos > 5% of max; spc_mask = bilinear (no antialias)
    downsample of any_z(cmos_mask) to the spc grid, cast to bool (non-zero -> True)."""
    cmos_mask = cmos > (0.05 * cmos.max())
    spc_mask = upsample_bilinear(
        cmos_mask.any(dim=0).unsqueeze(0).float(), (spc.shape[-2], spc.shape[-1]), antialias=False
    ).bool()
    return spc_mask, cmos_mask


def prepare(spc, cmos, *, weights, init_type, tol=3e-3, mask_noise=False, total_energy=1.0, seed=42) -> Problem:
    """Fusion.__init__ (fusion.py:135-199) on the CPU."""
    spc = _as_f32(spc, "spc")
    cmos = _as_f32(cmos, "cmos")
    r = cmos.shape[-1] // spc.shape[-1]                       # fusion.py:169
    spc_mask, cmos_mask = get_masks(spc, cmos)                # fusion.py:182
    if mask_noise:                                            # fusion.py:185-187
        spc = spc * spc_mask.float()
        cmos = cmos * cmos_mask.float()
    spc_mask = spc_mask.squeeze(0)                            # fusion.py:189
    spc = normalize_energy(spc, total_energy)                 # fusion.py:192-193
    cmos = normalize_energy(cmos, total_energy)
    shape = (spc.shape[0], spc.shape[1], cmos.shape[0], cmos.shape[1], cmos.shape[2])
    torch.manual_seed(seed)                                   # fusion.py:259
    if init_type == "random":                                 # fusion.py:262-264
        x = (cmos.min() - cmos.max()) * torch.rand(shape) + cmos.max()
    elif init_type == "zeros":                                # fusion.py:266
        x = torch.zeros(shape)
    elif init_type == "baseline":                             # fusion.py:268
        x = baseline_init(cmos, spc)
    else:
        raise ValueError("Invalid initialization type.")      # fusion.py:271
    if mask_noise:                                            # fusion.py:197-198
        x[:, :, ~cmos_mask] = 0.0
    x = normalize_energy(x, total_energy)                     # fusion.py:199
    return Problem(spc, cmos, x, spc_mask, cmos_mask, float(weights["spatial"]),
                   float(weights["spectro_temporal"]), r, mask_noise, tol)


def _as_f32(a, name):
    if isinstance(a, np.ndarray):                             # fusion.py:135-148
        return torch.from_numpy(a.astype(np.float32))
    if isinstance(a, torch.Tensor):
        return a.to("cpu")
    raise TypeError(f"{name} must be a numpy.ndarray or torch.Tensor")


# --------------------------------------------------------------------------------------
# loss / tolerance (fusion.py:211-252)
# --------------------------------------------------------------------------------------


def project(pb: Problem, x: torch.Tensor):
    """u = T(S(x)) -> (1,1,nZ,X,Y);  v = R(D(x)) -> (nT,nL,1,Xs,Ys)."""
    u = sum_keep(sum_keep(x, 1), 0)                            # fusion.py:227 (S then T)
    v = block_sum(sum_keep(x, 2), pb.r)                        # fusion.py:231
    return u, v


def loss(pb: Problem, x: torch.Tensor | None = None):
    """fusion.py:211-242 -> (spatial_loss, spectro_temporal_loss), no 1/2 factor."""
    x = pb.x if x is None else x
    u, v = project(pb, x)
    if pb.mask_noise:                                          # fusion.py:216-223
        d1 = pb.cmos[pb.cmos_mask] - u[0, 0, pb.cmos_mask]
        d2 = pb.spc[:, :, pb.spc_mask] - v.squeeze(2)[:, :, pb.spc_mask]
    else:                                                      # fusion.py:225-232
        d1 = pb.cmos.flatten() - u.flatten()
        d2 = pb.spc.flatten() - v.squeeze(2).flatten()
    return pb.w1 * (d1**2).sum(), pb.w2 * (d2**2).sum()


def tolerance(x: torch.Tensor, prev_x: torch.Tensor) -> torch.Tensor:
    """fusion.py:244-252."""
    return torch.linalg.vector_norm(x.flatten() - prev_x.flatten()) / torch.linalg.vector_norm(x.flatten())


# --------------------------------------------------------------------------------------
# FusionCG.__call__ (fusion.py:399-496)
# --------------------------------------------------------------------------------------


def normal_op(pb: Problem, x: torch.Tensor) -> torch.Tensor:
    """A(x) = w1 T.T S.T S T x + w2 D.T R.T R D x   (fusion.py:423-426)."""
    n_t, n_l, n_z = x.shape[0], x.shape[1], x.shape[2]
    a1 = spread_mean(spread_mean(sum_keep(sum_keep(x, 0), 1), n_l, 1), n_t, 0)
    a2 = spread_mean(block_spread(block_sum(sum_keep(x, 2), pb.r), pb.r), n_z, 2)
    return pb.w1 * a1 + pb.w2 * a2


def rhs(pb: Problem) -> torch.Tensor:
    """b (fusion.py:428-430)."""
    n_t, n_l, n_z = pb.dims[0], pb.dims[1], pb.dims[2]
    b1 = spread_mean(spread_mean(pb.cmos.unsqueeze(0).unsqueeze(0), n_l, 1), n_t, 0)
    b2 = spread_mean(block_spread(pb.spc.unsqueeze(2), pb.r), n_z, 2)
    return pb.w1 * b1 + pb.w2 * b2


@torch.no_grad()
def cg(pb: Problem, max_iterations: int, eps: float = 1e-10, with_loss: bool = True,
       timings: dict | None = None) -> torch.Tensor:
    """The loop of fusion.py:435-456, same op order, same eps placement.  Appends one dict per
    iteration to pb.trace.  `with_loss=False` skips the per-iteration loss() (fusion.py:458),
    which does not influence x.  `timings` (bench.py's cpu_baseline leg): filled with "setup_s" (b, r0, rsold:
    fusion.py:428-437, paid once per call) and "iter_s" (wall time of each loop pass)."""
    import time
    t_start = time.perf_counter()
    b = rhs(pb)
    r = b - normal_op(pb, pb.x)                                # fusion.py:435
    p = r
    rsold = torch.dot(r.flatten(), r.flatten())
    if timings is not None:
        timings["setup_s"] = time.perf_counter() - t_start
        timings["iter_s"] = []
    t_iter = time.perf_counter()
    for _ in range(int(max_iterations)):
        if timings is not None and pb.trace and len(timings["iter_s"]) < len(pb.trace):
            now = time.perf_counter()
            timings["iter_s"].append(now - t_iter)
            t_iter = now
        pb.curr_iter += 1
        prev_x = pb.x.clone() if pb.tol is not None else None  # fusion.py:441
        ap = normal_op(pb, p)
        alpha = rsold / (torch.dot(p.flatten(), ap.flatten()) + eps)   # fusion.py:444
        pb.x = pb.x + p * alpha
        r = r - ap * alpha
        rsnew = torch.dot(r.flatten(), r.flatten())
        assert rsnew.isfinite(), "Conjugate gradient diverged"  # fusion.py:448
        tol_v = tolerance(pb.x, prev_x) if pb.tol is not None else None
        rec = {"alpha": float(alpha), "rsnew": float(rsnew),
               "tolerance": None if tol_v is None else float(tol_v)}
        if pb.tol is not None and tol_v < pb.tol:               # fusion.py:452-453
            pb.trace.append(rec)
            break
        p = r + p * (rsnew / (rsold + eps))                     # fusion.py:455
        rsold = rsnew
        if with_loss:                                           # fusion.py:458
            l1, l2 = loss(pb)
            rec["spatial_loss"], rec["spectro_temporal_loss"] = float(l1), float(l2)
        pb.trace.append(rec)
    if timings is not None and len(timings["iter_s"]) < len(pb.trace):
        timings["iter_s"].append(time.perf_counter() - t_iter)
    return pb.x


# --------------------------------------------------------------------------------------
# FusionAdam.__call__ (fusion.py:303-386)
# --------------------------------------------------------------------------------------


def adam(pb: Problem, lr: float, max_iterations: int, non_neg: bool = False) -> torch.Tensor:
    """fusion.py:324-358: autograd through loss(), torch.optim.Adam(amsgrad=False)."""
    x = torch.nn.Parameter(pb.x.clone(), requires_grad=True)
    opt = torch.optim.Adam([x], lr=lr, amsgrad=False)
    for _ in range(max_iterations):
        pb.curr_iter += 1
        prev_x = x.detach().clone() if pb.tol is not None else None
        opt.zero_grad()
        l1, l2 = loss(pb, x)
        (l1 + l2).backward()
        grad0 = x.grad.detach().clone() if not pb.trace else None
        if pb.mask_noise:                                       # fusion.py:274-278
            x.grad[:, :, ~pb.cmos_mask] = 0.0
        opt.step()
        if non_neg:                                             # fusion.py:344-346
            with torch.no_grad():
                x.copy_(x.data.clamp(min=0))
        tol_v = float(tolerance(x.detach(), prev_x)) if pb.tol is not None else None
        rec = {"spatial_loss": float(l1.detach()), "spectro_temporal_loss": float(l2.detach()), "tolerance": tol_v}
        if grad0 is not None:
            rec["grad"] = grad0
        pb.trace.append(rec)
        if tol_v is not None and tol_v < pb.tol:
            break
    pb.x = x.detach()
    return pb.x


# --------------------------------------------------------------------------------------
# Collapsed float64 oracle - for shapes no eager restatement fits.
# x_k = x_0 + bcast_{t,l}(a_k) + bcast_{z,block}(g_k); the whole CG runs on (a, g).
# --------------------------------------------------------------------------------------


class _Small:
    """Element of span{B1 a, B2 g}: B1 a broadcasts a (nZ,X,Y) over (t,l); B2 g broadcasts
    g (nT,nL,Xs,Ys) over z and each r x r block."""

    __slots__ = ("a", "g")

    def __init__(self, a, g):
        self.a, self.g = a, g

    def axpy(self, s, o):
        return _Small(self.a + s * o.a, self.g + s * o.g)


def collapsed_cg(spc_n, cmos_n, u0, v0, dims, r, w1, w2, max_iterations, eps=1e-10):
    """Float64 CG in the small space.  Inputs: normalised spc (nT,nL,Xs,Ys), cmos (nZ,X,Y),
    u0 = u(x0) (nZ,X,Y), v0 = v(x0) (nT,nL,Xs,Ys), all converted to float64.
    Returns (a, g, alphas, rsnews) with x_k = x0 + B1 a + B2 g."""
    n_t, n_l, n_z, X, Y = dims
    f8 = lambda t: torch.as_tensor(t).double()
    spc_n, cmos_n, u0, v0 = f8(spc_n), f8(cmos_n), f8(u0), f8(v0)
    c1, c2 = w1 / (n_t * n_l), w2 / (n_z * r * r)
    m1, m2 = n_t * n_l, n_z * r * r

    def blk(a):  # sum_z sum_block a -> (Xs, Ys)
        return a.sum(0).reshape(X // r, r, Y // r, r).sum((1, 3))

    def up(g2):  # (Xs,Ys) -> (X,Y) replicate
        return g2.repeat_interleave(r, 0).repeat_interleave(r, 1)

    def dot(p, q):
        return m1 * (p.a * q.a).sum() + m2 * (p.g * q.g).sum() + (blk(p.a) * q.g.sum((0, 1))).sum() \
            + (blk(q.a) * p.g.sum((0, 1))).sum()

    def apply_a(p):
        # u(B1 a + B2 g) = m1 a + up(sum_{t,l} g) ; v(...) = blk(a) + m2 g
        u = m1 * p.a + up(p.g.sum((0, 1)))[None]
        v = blk(p.a)[None, None] + m2 * p.g
        return _Small(c1 * u, c2 * v)

    rr = _Small(c1 * (cmos_n - u0), c2 * (spc_n - v0))
    pp = rr
    xs = _Small(torch.zeros_like(cmos_n), torch.zeros_like(spc_n))
    rsold = dot(rr, rr)
    alphas, rsnews = [], []
    for _ in range(max_iterations):
        ap = apply_a(pp)
        alpha = rsold / (dot(pp, ap) + eps)
        xs = xs.axpy(alpha, pp)
        rr = rr.axpy(-alpha, ap)
        rsnew = dot(rr, rr)
        pp = rr.axpy(rsnew / (rsold + eps), pp)
        rsold = rsnew
        alphas.append(float(alpha))
        rsnews.append(float(rsnew))
    return xs.a, xs.g, alphas, rsnews


def expand_small(x0: torch.Tensor, a: torch.Tensor, g: torch.Tensor, r: int) -> torch.Tensor:
    """x0 + B1 a + B2 g as a float64 5-D tensor (small shapes only)."""
    up = g.repeat_interleave(r, 2).repeat_interleave(r, 3)
    return x0.double() + a[None, None] + up[:, :, None]


def rel_l2(a, b) -> float:
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return float(torch.linalg.vector_norm(a - b) / torch.linalg.vector_norm(b))


def synthetic_inputs(n_t, n_l, n_z, xy, r=4, seed=0):
    """Bench/test inputs of SURVEY.md section 8(d): spc then cmos drawn from one CPU generator,
    0.1 + 0.9 U[0,1), strictly positive."""
    g = torch.Generator().manual_seed(seed)
    spc = 0.1 + 0.9 * torch.rand(n_t, n_l, xy // r, xy // r, generator=g)
    cmos = 0.1 + 0.9 * torch.rand(n_z, xy, xy, generator=g)
    return spc, cmos


def cells_like_inputs(n_t, n_l, n_z, xy, r=4, seed=0, noise=0.02):
    """Synthetic 'cells-like' acquisition (the Zenodo data are not available offline):
    a few Gaussian blobs per z-plane on a dark background, mono-exponential decays in t,
    Gaussian emission spectra in lambda, plus a small positive noise floor.  Returned as float64
    ndarrays like utils.py:411-427 load_data() returns: spc (t, l, xs, ys), cmos (z, x, y)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(xy), np.arange(xy))
    vol = np.zeros((n_z, xy, xy))
    n_blobs = 5
    cx, cy = rng.uniform(0.2, 0.8, (2, n_blobs)) * xy
    cz = rng.uniform(0, n_z, n_blobs)
    sig = rng.uniform(0.04, 0.09, n_blobs) * xy
    tau = rng.uniform(0.1, 0.4, n_blobs) * n_t
    lam0 = rng.uniform(0.2, 0.8, n_blobs) * n_l
    t_ax, l_ax = np.arange(n_t), np.arange(n_l)
    full = np.zeros((n_t, n_l, n_z, xy, xy))
    for b in range(n_blobs):
        sp = np.exp(-((xx - cx[b]) ** 2 + (yy - cy[b]) ** 2) / (2 * sig[b] ** 2))
        zp = np.exp(-((np.arange(n_z) - cz[b]) ** 2) / (2 * (0.15 * n_z + 0.5) ** 2))
        blob = zp[:, None, None] * sp[None]
        vol += blob
        dec = np.exp(-t_ax / tau[b])
        spec = np.exp(-((l_ax - lam0[b]) ** 2) / (2 * (0.12 * n_l + 0.5) ** 2))
        full += dec[:, None, None, None, None] * spec[None, :, None, None, None] * blob[None, None]
    cmos = full.sum((0, 1)) + noise * vol.max() * rng.uniform(0, 1, vol.shape)
    spc = full.sum(2).reshape(n_t, n_l, xy // r, r, xy // r, r).sum((3, 5))
    spc = spc + noise * spc.max() * rng.uniform(0, 1, spc.shape)
    return spc, cmos
