"""DfxAdam: torch.optim.Adam's update (amsgrad=False, weight_decay=0, maximize=False) as ONE HIP launch per parameter.

For the autograd formulation of FusionAdam (fusion.py:324-346: loss() -> loss.backward() -> optimizer.step()), where the
gradient is an explicit 5-D tensor - e.g. because the caller added differentiable terms of their own to the loss.
torch's default (foreach) Adam makes seven full passes over x, grad, exp_avg, exp_avg_sq per step and its fused variant
52 launches on a 4 GiB parameter; dfx_adam_apply reads the four arrays and writes three in one pass, fuses the optional
non-negativity clamp (fusion.py:344-346) and returns ||x_new - x_old||^2 and ||x_new||^2, so the tolerance of
fusion.py:244-252 needs neither the clone of fusion.py:329 nor a subtraction pass.
State keys are torch.optim.Adam's ("step", "exp_avg", "exp_avg_sq").  No CPU path: parameters must be float32,
contiguous, on an AMD GPU.
"""
from __future__ import annotations

import torch

from . import _cabi as cabi


class DfxAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, non_neg: bool = False,
                 n_blocks: int = 0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("DfxAdam: lr, eps >= 0 and betas in [0, 1) required")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, non_neg=non_neg))
        self.n_blocks = int(n_blocks)
        self.last_sq_norms = None   # device tensor (2,) float64: ||dx||^2, ||x||^2 of the last parameter stepped

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = cabi.lib()
        for group in self.param_groups:
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
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["_scal"] = torch.zeros(cabi.NSCAL, dtype=torch.float64, device=p.device)
                    st["_part"] = torch.empty(cabi.EW_MAXBLOCKS * cabi.NPART, dtype=torch.float64, device=p.device)
                st["step"] += 1
                with torch.cuda.device(p.device):
                    cabi.check(lib.dfx_adam_apply(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                                  st["exp_avg_sq"].data_ptr(), p.numel(), group["lr"], group["betas"][0],
                                                  group["betas"][1], group["eps"], st["step"],
                                                  cabi.ADAM_NONNEG if group["non_neg"] else 0, self.n_blocks,
                                                  st["_scal"].data_ptr(), st["_part"].data_ptr(),
                                                  torch.cuda.current_stream(p.device).cuda_stream), "dfx_adam_apply")
                self.last_sq_norms = st["_scal"][cabi.S_ADAM_DD:cabi.S_ADAM_XX + 1]
        return loss
