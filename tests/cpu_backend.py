"""CPU stand-in for datafusion_amd.kernels.HipKernels - TEST INFRASTRUCTURE ONLY.

Implements the same small interface with plain torch-CPU ops so that the host logic of
datafusion_amd.fusion (call order, which scalars/partials are all-reduced when, z-sharding,
deferred x update, tolerance bookkeeping) can run under gloo in a GPU-less container.  It is
never importable from the product package."""
import math

import torch

from datafusion_amd import _cabi as cabi


class _CpuAdam(torch.optim.Adam):
    """Stand-in for datafusion_amd.optim.DfxAdam: torch.optim.Adam + the fused clamp and the two squared norms."""

    def __init__(self, params, lr, betas, eps, non_neg):
        super().__init__(params, lr=lr, betas=betas, eps=eps, amsgrad=False)
        self.non_neg, self.last_sq_norms = non_neg, None

    @torch.no_grad()
    def step(self, closure=None):
        p = self.param_groups[0]["params"][0]
        prev = p.detach().clone()
        out = super().step(closure)
        if self.non_neg:
            p.clamp_(min=0)
        self.last_sq_norms = torch.stack([((p - prev).double() ** 2).sum(), (p.double() ** 2).sum()])
        return out


class CpuStandIn:
    @staticmethod
    def adam_optimizer(params, lr, betas, eps, non_neg, moment_buffers=None, projectors=None):
        return _CpuAdam(params, lr, betas, eps, non_neg)      # no fused projection: `projected` stays absent

    def reserved_left(self):
        return 0

    def __init__(self, n_t, n_l, n_z_local, X, Y, r, device, target_blocks=0):
        self.device = torch.device("cpu")
        self.shape = (n_t, n_l, n_z_local, X, Y)
        self.r, self.Xs, self.Ys = r, X // r, Y // r
        self.M, self.S = n_t * n_l, n_z_local * X * Y
        self.N = self.M * self.S
        self.scal = torch.zeros(cabi.NSCAL, dtype=torch.float64)
        self.u = torch.zeros(n_z_local, X, Y)
        self.v = torch.zeros(n_t, n_l, self.Xs, self.Ys)
        self.ra = self.rb = self.zab = None
        self.timing = None

    def new_factored_residual(self):
        if self.ra is None:
            n_t, n_l, n_z, X, Y = self.shape
            self.ra, self.rb = torch.zeros(n_z, X, Y), torch.zeros(n_t, n_l, self.Xs, self.Ys)
            self.zab = torch.zeros(self.Xs, self.Ys)

    def _resid5(self):
        n_t, n_l, n_z, X, Y = self.shape
        return self.ra.reshape(1, 1, n_z, X, Y) + \
            self.rb.repeat_interleave(self.r, 2).repeat_interleave(self.r, 3).unsqueeze(2)

    def resid_update(self, A, B, sa, sb, coef_host, coef_slot, accumulate):
        c = coef_host * (1.0 if coef_slot is None else float(self.scal[coef_slot]))
        if not accumulate:
            self.ra.zero_(), self.rb.zero_()
        self.ra.add_(A.reshape(self.ra.shape), alpha=c * sa)
        self.rb.add_(B.reshape(self.rb.shape), alpha=c * sb)
        self.scal[cabi.S_RR] = (self._resid5().double() ** 2).sum()   # the plain 5-D norm (the HIP side uses the
                                                                       # small-space identity: cross-checks it)

    def diff_scale(self, a, b, mask, coef):
        out = torch.tensor(coef, dtype=torch.float32) * (a - b)
        return out * mask if mask is not None else out

    def new_vector_paired(self, ref, tries=3):
        return self.new_vector()

    def reserve_vectors(self, n, mode="rank", min_bytes=None):
        pass

    def new_vector(self):
        return torch.empty(self.shape)

    def _proj(self, a):
        n_t, n_l, n_z, X, Y = self.shape
        self.u.copy_(a.sum((0, 1)))
        self.v.copy_(a.sum(2).reshape(n_t, n_l, self.Xs, self.r, self.Ys, self.r).sum((3, 5)))
        self.scal[cabi.S_UU] = (self.u.double() ** 2).sum()

    def post_sweep(self, flags, c1=0.0, c2=0.0, eps=0.0):
        # u, UU and the scalar partials are already final here (the stand-in's sweep / adam_step computes them whole)
        if flags & cabi.POST_V:
            self.sqnorm_v()
        if flags & cabi.POST_ALPHA:
            self.cg_alpha(c1, c2, eps)

    def resid_cg_step(self, c1, c2, eps, deferred):
        self.resid_update(self.u, self.v, c1, c2, -1.0, cabi.S_ALPHA, True)
        self.cg_beta(eps, deferred)

    def adam_prep(self, cmos, cmask, spc, smask, A, B, m1a, m1b, w1, w2, lerp_weight):
        A.copy_(self.diff_scale(cmos, self.u, cmask, -2.0 * w1))
        B.copy_(self.diff_scale(spc, self.v, smask, -2.0 * w2))
        if m1a is not None:
            self.lerp(m1a, A, lerp_weight)
            self.lerp(m1b, B, lerp_weight)
        self.sqdiff(cmos, self.u, cmask, w1, cabi.S_LOSS1)
        self.sqdiff(spc, self.v, smask, w2, cabi.S_LOSS2)

    def sweep(self, mode, x, r=None, p=None, slices=1, after_slice=None, factored=False, fold=True):
        if factored:
            assert r is None
            r = self._resid5()
        self._sweep(mode, x, r, p)
        if after_slice is not None:           # hand v over in `slices` bands, like the HIP backend
            rows = self.v.view(self.M, -1)
            k = max(1, min(slices, self.M))
            for j in range(k):
                after_slice(rows[self.M * j // k: self.M * (j + 1) // k])

    def _sweep(self, mode, x, r=None, p=None):
        s = self.scal
        al, be = float(s[cabi.S_ALPHA]), float(s[cabi.S_BETA])
        if mode == cabi.SWEEP_PROJECT:
            self._proj(x)
            s[cabi.S_PP], s[cabi.S_XX], s[cabi.S_XP] = 0.0, (x.double() ** 2).sum(), 0.0
            return
        if mode == cabi.SWEEP_XP:
            x.add_(p, alpha=al)
            p.mul_(be).add_(r)
        elif mode == cabi.SWEEP_P:
            p.mul_(be).add_(r)
        else:
            p.copy_(r)
        self._proj(p)
        s[cabi.S_PP] = (p.double() ** 2).sum()
        if mode == cabi.SWEEP_P:
            s[cabi.S_XX], s[cabi.S_XP] = 0.0, 0.0
        else:
            s[cabi.S_XX] = (x.double() ** 2).sum()
            s[cabi.S_XP] = (x.double() * p.double()).sum()

    def lerp(self, m, g, weight):
        m.lerp_(g.reshape(m.shape), weight)

    def adam_step(self, x, m1, m2, A, B, mask, lr, beta1, beta2, eps, step, non_neg, m1a=None, m1b=None, fold=True,
                  slices=1, after_slice=None):
        n_z, X, Y = self.shape[2:]

        def expand(a, b):
            return a.reshape(1, 1, n_z, X, Y) + b.reshape(self.v.shape).repeat_interleave(self.r, 2) \
                .repeat_interleave(self.r, 3).unsqueeze(2)

        g = expand(A, B)
        if mask is not None:
            g = g * mask.reshape(1, 1, n_z, X, Y)
        x_old = x.clone()
        if m1 is not None:
            m1.lerp_(g, 1 - beta1)
        else:                                   # factored first moment, already advanced by lerp()
            m1 = expand(m1a, m1b)
            if mask is not None:
                m1 = m1 * mask.reshape(1, 1, n_z, X, Y)
        m2.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
        x.addcdiv_(m1, (m2.sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)
        if non_neg:
            x.clamp_(min=0)
        self._proj(x)
        self.scal[cabi.S_ADAM_DD] = ((x - x_old).double() ** 2).sum()
        self.scal[cabi.S_ADAM_XX] = (x.double() ** 2).sum()
        if after_slice is not None:           # hand v over in `slices` bands, like the HIP backend
            rows = self.v.view(self.M, -1)
            k = max(1, min(slices, self.M))
            for j in range(k):
                after_slice(rows[self.M * j // k: self.M * (j + 1) // k])

    def sum_axis(self, x, axis):
        return x.sum(axis)

    def sqnorm_v(self):
        self.scal[cabi.S_VV] = (self.v.double() ** 2).sum()

    def bcast(self, out, inp, A, B, mask, sa, sb, coef_host, coef_slot):
        coef = coef_host * (float(self.scal[coef_slot]) if coef_slot is not None else 1.0)
        n_z, X, Y = self.shape[2:]
        t = (coef * sa) * A.reshape(1, 1, n_z, X, Y) + (coef * sb) * \
            B.repeat_interleave(self.r, 2).repeat_interleave(self.r, 3).unsqueeze(2)
        if mask is not None:
            t = t * mask.reshape(1, 1, n_z, X, Y)
        if inp is not None:
            t = t + inp
        out.copy_(t)
        self.scal[cabi.S_RR] = (out.double() ** 2).sum()

    def axpy(self, y, x, coef_host, coef_slot):
        y.add_(x, alpha=coef_host * (float(self.scal[coef_slot]) if coef_slot is not None else 1.0))
        self.scal[cabi.S_XXNEW] = (y.double() ** 2).sum()

    def scale(self, y, mask, coef_host, coef_slot):
        if mask is not None:
            y.view(self.M, -1)[:, mask.reshape(-1) == 0] = 0.0
        y.mul_(coef_host * (float(self.scal[coef_slot]) if coef_slot is not None else 1.0))

    def sum(self, y):
        self.scal[cabi.S_SUM] = y.double().sum()

    def recip(self, dst, src, num):
        self.scal[dst] = num / self.scal[src]

    def sqdiff(self, a, b, mask, w, slot):
        d = (a.double() - b.double()) ** 2
        if mask is not None:
            d = d.reshape(-1, mask.numel()) * mask.reshape(1, -1).double()
        self.scal[slot] = w * d.sum()

    def baseline_init(self, spc, zw, x, coef_host, coef_slot):
        coef = coef_host * (float(self.scal[coef_slot]) if coef_slot is not None else 1.0)
        X, Y = self.shape[3:]
        up = torch.nn.functional.interpolate(spc, size=(X, Y), mode="bilinear", align_corners=False)
        full = coef * up.unsqueeze(2) * zw.unsqueeze(0).unsqueeze(0)
        self.scal[cabi.S_SUM] = full.double().sum()
        if x is not None:
            x.copy_(full)

    def cg_start(self):
        s = self.scal
        s[cabi.S_RSOLD] = s[cabi.S_RSNEW] = s[cabi.S_RR]
        s[cabi.S_ALPHA] = s[cabi.S_BETA] = s[cabi.S_TOL] = s[cabi.S_ITER] = 0.0
        s[cabi.S_NONFINITE] = 0.0 if math.isfinite(float(s[cabi.S_RR])) else 1.0

    def cg_alpha(self, c1, c2, eps):
        s = self.scal
        s[cabi.S_PAP] = c1 * s[cabi.S_UU] + c2 * s[cabi.S_VV]
        s[cabi.S_ALPHA] = s[cabi.S_RSOLD] / (s[cabi.S_PAP] + eps)

    def cg_beta(self, eps, deferred):
        s = self.scal
        rs = float(s[cabi.S_RR])
        s[cabi.S_RSNEW] = rs
        if not math.isfinite(rs):
            s[cabi.S_NONFINITE] += 1.0
        al, pp = float(s[cabi.S_ALPHA]), float(s[cabi.S_PP])
        xxn = float(s[cabi.S_XX]) + 2 * al * float(s[cabi.S_XP]) + al * al * pp if deferred else float(s[cabi.S_XXNEW])
        s[cabi.S_TOL] = abs(al) * math.sqrt(pp) / math.sqrt(xxn) if xxn > 0 else float("nan")
        s[cabi.S_BETA] = rs / (float(s[cabi.S_RSOLD]) + eps)
        s[cabi.S_RSOLD] = rs
        s[cabi.S_ALPHA_LAST] = al
        s[cabi.S_ITER] += 1.0
