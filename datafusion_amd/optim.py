"""DfxAdam: torch.optim.Adam's update (amsgrad=False, weight_decay=0, maximize=False) as ONE HIP launch per parameter.

For the autograd formulation of FusionAdam (fusion.py:324-346: loss() -> loss.backward() -> optimizer.step()), where the
gradient is an explicit 5-D tensor - e.g. because the caller added differentiable terms of their own to the loss.
torch's default (foreach) Adam makes seven full passes over x, grad, exp_avg, exp_avg_sq per step and its fused variant
52 launches on a 4 GiB parameter; dfx_adam_apply reads the four arrays and writes three in one pass, fuses the optional
non-negativity clamp (fusion.py:344-346) and returns ||x_new - x_old||^2 and ||x_new||^2, so the tolerance of
fusion.py:244-252 needs neither the clone of fusion.py:329 nor a subtraction pass.
State: exactly torch.optim.Adam's keys ("step", "exp_avg", "exp_avg_sq"), so state_dict() / load_state_dict()
round-trip with a torch.optim.Adam of the same parameters ("step" is kept as torch keeps it: a 0-dim float32 tensor;
a plain number is accepted).  The launch's scratch (scalars + per-workgroup partials, 576 KiB per device) lives
outside the state.  The update uses the IEEE square root and division like torch's (the 5-D kernels of the fused
FusionAdam path use the 1-ulp hardware approximations; this one is bound by its seven memory streams either way); the
bias correction of the denominator is a multiplication by 1 / sqrt(1 - beta2^t) where torch divides, so the last bit
can differ from torch.optim.Adam's.
`last_sq_norms` holds ||x_new - x_old||^2 and ||x_new||^2 SUMMED over every parameter stepped by the last step().
No CPU path: parameters must be float32, contiguous, on an AMD GPU.
"""
from __future__ import annotations

import torch

from . import _cabi as cabi


class DfxAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, non_neg: bool = False,
                 n_blocks: int = 0, moment_buffers=None, projectors=None):
        """moment_buffers: optional {parameter: (exp_avg, exp_avg_sq)} of caller-placed tensors shaped like the
        parameter (FusionAdam passes vectors laid out by memory class next to x, datafusion_amd/placement.py: the
        step's seven streams run at 0.78 of the HBM peak spread over two or more classes, 0.72 inside one); they are
        zeroed when first used.  Default: torch.zeros_like.
        projectors: optional {parameter: HipKernels of the 5-D volume the parameter is}: the step of such a parameter
        runs as dfx_adam_apply_project - the same update, in the sweep's lane-mapped structure, which also leaves
        u(x), v(x) of the STEPPED volume in the HipKernels object (FusionAdam's next forward then needs no projection
        pass); `projected` holds the parameters the last step() handled that way."""
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("DfxAdam: lr, eps >= 0 and betas in [0, 1) required")
        # the param-group keys of torch.optim.Adam as well (fixed to what this optimizer implements), so that a
        # state_dict() of either optimizer loads into the other
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, non_neg=non_neg, weight_decay=0, amsgrad=False,
                                      maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        self._non_neg = bool(non_neg)
        self._moment_buffers = dict(moment_buffers or {})
        self.n_blocks = int(n_blocks)
        self.last_sq_norms = None   # device tensor (2,) float64: ||dx||^2, ||x||^2 over the parameters of the last step()
        self._scratch = {}          # device -> (scal, part): launch scratch, not optimizer state
        self._projectors = dict(projectors or {})
        self.projected = set()      # parameters whose last step also projected the stepped volume

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = cabi.lib()
        total = None
        self.projected = set()
        for group in self.param_groups:
            if group.get("weight_decay", 0) or group.get("amsgrad", False) or group.get("maximize", False):
                raise cabi.DfxError("DfxAdam implements Adam with weight_decay=0, amsgrad=False, maximize=False only")
            non_neg = group.get("non_neg", self._non_neg)      # absent after loading a torch.optim.Adam checkpoint
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous():
                    raise cabi.DfxError("DfxAdam: parameters must be contiguous float32 tensors on an AMD GPU")
                if g.is_sparse:
                    raise cabi.DfxError("DfxAdam does not take sparse gradients")
                g = g.contiguous()
                st = self.state[p]
                if "exp_avg" not in st:                       # fresh, or a state that holds only a step count
                    st.setdefault("step", torch.zeros((), dtype=torch.float32))
                    bufs = self._moment_buffers.pop(p, None)
                    if bufs is not None:
                        st["exp_avg"], st["exp_avg_sq"] = bufs[0].view(p.shape).zero_(), bufs[1].view(p.shape).zero_()
                    else:
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                for key in ("exp_avg", "exp_avg_sq"):         # e.g. loaded from a checkpoint taken on another device
                    if st[key].device != p.device or st[key].dtype != torch.float32 or not st[key].is_contiguous():
                        st[key] = st[key].to(device=p.device, dtype=torch.float32).contiguous()
                step = int(st["step"]) + 1                    # tensor (torch.optim.Adam's) or number
                st["step"] = st["step"] + 1 if isinstance(st["step"], torch.Tensor) else step
                k = self._projectors.get(p)
                if k is not None and p.numel() == k.N and p.device == k.device and k.adam_apply_project(
                        p.data, g, st["exp_avg"], st["exp_avg_sq"], group["lr"], group["betas"][0], group["betas"][1],
                        group["eps"], step, non_neg):
                    self.projected.add(p)
                    norms = k.scal[cabi.S_ADAM_DD:cabi.S_ADAM_XX + 1]
                    total = norms.clone() if total is None else total + norms.to(total.device)
                    continue
                if p.device not in self._scratch:
                    self._scratch[p.device] = (torch.zeros(cabi.NSCAL, dtype=torch.float64, device=p.device),
                                               torch.empty(cabi.EW_MAXBLOCKS * cabi.NPART, dtype=torch.float64,
                                                           device=p.device))
                scal, part = self._scratch[p.device]
                with torch.cuda.device(p.device):
                    cabi.check(lib.dfx_adam_apply(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                                  st["exp_avg_sq"].data_ptr(), p.numel(), group["lr"], group["betas"][0],
                                                  group["betas"][1], group["eps"], step,
                                                  cabi.ADAM_NONNEG if non_neg else 0, self.n_blocks,
                                                  scal.data_ptr(), part.data_ptr(),
                                                  torch.cuda.current_stream(p.device).cuda_stream), "dfx_adam_apply")
                norms = scal[cabi.S_ADAM_DD:cabi.S_ADAM_XX + 1]
                total = norms.clone() if total is None else total + norms.to(total.device)
        if total is not None:
            self.last_sq_norms = total
        return loss
