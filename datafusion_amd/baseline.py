"""baseline() (datafusion/baseline.py:8-52) as one HIP pass instead of nT*nZ Python iterations."""
from __future__ import annotations

import numpy as np
import torch

from . import hostcopy
from .kernels import HipKernels


def baseline(cmos, spc, device: str = "cuda", return_numpy: bool = True):
    """x[t, l, z] = bilinear_upsample(spc[t, l]) * cmos[z] / sum_z cmos  -> (time, lambda, z, x, y).
    Same arguments as the reference (baseline.py:8-13); `device` must be an AMD GPU."""
    dev = torch.device(device)
    if isinstance(spc, np.ndarray):
        spc = torch.from_numpy(spc.astype(np.float32))       # baseline.py:24-28
    if isinstance(cmos, np.ndarray):
        cmos = torch.from_numpy(cmos.astype(np.float32))
    spc = spc.to(dev).float().contiguous()
    cmos = cmos.to(dev).float()
    r = cmos.shape[-1] // spc.shape[-1]
    if cmos.shape[1] != r * spc.shape[2] or cmos.shape[2] != r * spc.shape[3]:
        raise ValueError("cmos (z, x, y) must be an integer multiple of spc (t, l, x/r, y/r) in x and y")
    k = HipKernels(spc.shape[0], spc.shape[1], cmos.shape[0], cmos.shape[1], cmos.shape[2], r, dev)
    zw = (cmos / cmos.sum(dim=0, keepdim=True)).contiguous()  # baseline.py:46 (small array)
    x = k.new_vector()
    k.baseline_init(spc, zw, x, 1.0, None)                    # baseline.py:48-50
    return hostcopy.to_numpy(x) if return_numpy else x
