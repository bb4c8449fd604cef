"""Input stage of the README flow: `load_data` (datafusion/utils.py:411-427) without scikit-image.

The reference resizes the CMOS stack with `skimage.transform.resize(cmos_xyz, (S, S, nZ))` (defaults:
order 1, mode "reflect", anti-aliasing on when shrinking).  scikit-image is not part of this image, so
its published algorithm (skimage >= 0.19, `transform/_warps.py: resize`) is restated: a Gaussian pre-filter with
sigma = (factor - 1) / 2 per shrunk axis, then order-1 interpolation at the output grid's cell centres
(`ndimage.zoom(order=1, grid_mode=True)`), both with scipy's mode "mirror" - skimage's mode names are numpy.pad's and
its `_to_ndimage_mode` translates the default "reflect" (d c b | a b c d, edge not repeated) to ndimage's "mirror".
Integer stacks are rescaled like skimage's `img_as_float` (unsigned: / dtype max; signed: onto [-1, 1]; bool: 0 / 1)
before resizing, as `resize(preserve_range=False)` does.
  * `device=` given: the stack is resized ON THE GPU by dfx_resize_stack (hand-written kernels, float64), so the
    README flow needs neither scikit-image nor a host pass over the full-resolution acquisition;
  * `device=None`: the same restatement over scipy.ndimage on the host (the checker of the device path in the tests).
PARITY UNPINNED: neither scikit-image nor the Zenodo acquisitions are available offline, so this stage is tested
against the scipy calls skimage's resize is built from, plus shape / identity / conservation properties.
dtype: float64 in, float64 out; a float32 stack comes back as float32 like skimage's (computed in float64 and rounded
once at the end, where skimage rounds after each filter pass).
The wire format is the reference's: an .npz with keys spc (t, l, x~, y~), cmos, time_axis, spectral_axis.
"""
from __future__ import annotations

import numpy as np


def as_float_like_skimage(image) -> np.ndarray:
    """skimage.util.img_as_float as resize(preserve_range=False) applies it: floats pass through (as float64 here),
    unsigned integers are divided by their dtype's maximum, signed ones mapped onto [-1, 1], bool becomes 0 / 1."""
    a = np.asarray(image)
    if a.dtype.kind == "f":
        return a.astype(np.float64)
    if a.dtype.kind == "b":
        return a.astype(np.float64)
    if a.dtype.kind == "u":
        return a.astype(np.float64) / float(np.iinfo(a.dtype).max)
    if a.dtype.kind == "i":
        lo, hi = float(np.iinfo(a.dtype).min), float(np.iinfo(a.dtype).max)
        return (a.astype(np.float64) - lo) / (hi - lo) * 2.0 - 1.0
    raise TypeError(f"unsupported image dtype {a.dtype}")


def resize_like_skimage(image: np.ndarray, output_shape) -> np.ndarray:
    """skimage.transform.resize(image, output_shape) with its defaults, for float images (host, scipy.ndimage)."""
    from scipy import ndimage as ndi
    out_dtype = np.float32 if np.asarray(image).dtype == np.float32 else np.float64
    image = as_float_like_skimage(image)
    output_shape = tuple(int(s) for s in output_shape)
    if image.ndim != len(output_shape):
        raise ValueError("output_shape must have one entry per image axis")
    factors = np.divide(image.shape, output_shape)
    if np.any(factors > 1):  # anti_aliasing defaults to True when down-sampling
        sigma = np.maximum(0, (factors - 1) / 2)
        image = ndi.gaussian_filter(image, sigma, cval=0, mode="mirror")
    zoom = [1 / f for f in factors]
    return ndi.zoom(image, zoom, order=1, mode="mirror", cval=0, grid_mode=True).astype(out_dtype, copy=False)


def resize_stack_on_device(stack_zxy, size: int, device, transpose_out: bool = True):
    """(z, X, Y) stack -> (z, size, size) on the GPU (dfx_resize_stack): what utils.py:420-422 computes, i.e.
    resize(transpose(stack, (1, 2, 0)), (size, size, z)) transposed back with (2, 1, 0) - note the x/y swap, kept.
    Returns a float64 (float32 for float32 input) device tensor."""
    import torch
    from . import _cabi as cabi
    dev = torch.device(device)
    if dev.type != "cuda":
        raise cabi.DfxError("resize_stack_on_device runs on AMD GPUs only")
    if isinstance(stack_zxy, np.ndarray) and stack_zxy.dtype.kind in "uib":
        stack_zxy = as_float_like_skimage(stack_zxy)          # img_as_float's rescaling of integer stacks
    src = torch.as_tensor(stack_zxy)
    out_dtype = torch.float32 if src.dtype == torch.float32 else torch.float64
    src = src.to(dev, torch.float64).contiguous()
    if src.ndim != 3:
        raise ValueError("expected a (z, X, Y) stack")
    n_z, X, Y = (int(v) for v in src.shape)
    tmp_a, tmp_b = torch.empty_like(src), torch.empty_like(src)
    out = torch.empty((n_z, size, size), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        cabi.check(cabi.lib().dfx_resize_stack(cabi.ptr(src), cabi.ptr(tmp_a), cabi.ptr(tmp_b), cabi.ptr(out), n_z, X, Y,
                                               size, size, int(transpose_out),
                                               torch.cuda.current_stream(dev).cuda_stream), "dfx_resize_stack")
    return out.to(out_dtype)


def load_data(path, max_xy_size: int, device=None, return_torch: bool = False):
    """-> (spc (t, l, x~, y~), cmos (z, x, y) resized to max_xy_size, time_axis, spectral_axis);
    same transposes as utils.py:420-422 (note the x/y swap of the final (2, 1, 0)).
    device=None: host (scipy); device="cuda[:i]": the resize runs on that GPU; return_torch=True hands cmos (and spc)
    back as device tensors, ready for FusionCG(spc, cmos, ...)."""
    data = np.load(path, allow_pickle=False)
    spc = data["spc"]
    if device is None:
        cmos = np.transpose(data["cmos"], (1, 2, 0))
        cmos = resize_like_skimage(cmos, (max_xy_size, max_xy_size, cmos.shape[2]))
        cmos = np.transpose(cmos, (2, 1, 0))
    else:
        cmos = resize_stack_on_device(data["cmos"], int(max_xy_size), device, transpose_out=True)
        if return_torch:
            import torch
            spc = torch.from_numpy(np.ascontiguousarray(spc)).to(cmos.device)
        else:
            cmos = cmos.cpu().numpy()
    return spc, cmos, data["time_axis"], data["spectral_axis"]
