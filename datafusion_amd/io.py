"""Input stage of the README flow: `load_data` (datafusion/utils.py:411-427) without scikit-image.

The reference resizes the CMOS stack with `skimage.transform.resize(cmos_xyz, (S, S, nZ))` (defaults:
order 1, mode "reflect", anti-aliasing on when shrinking).  scikit-image is not part of this image, so
its published algorithm is restated with the scipy.ndimage calls it is built from (skimage >= 0.19,
`transform/_warps.py: resize`): a Gaussian pre-filter with sigma = (factor - 1) / 2 per shrunk axis, then
`ndimage.zoom(order=1, mode="reflect", grid_mode=True)`.  PARITY UNPINNED: neither scikit-image nor the
Zenodo acquisitions are available offline, so this stage has only shape / identity / conservation tests.
The wire format is the reference's: an .npz with keys spc (t, l, x~, y~), cmos, time_axis, spectral_axis.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi


def resize_like_skimage(image: np.ndarray, output_shape) -> np.ndarray:
    """skimage.transform.resize(image, output_shape) with its defaults, for float images."""
    image = np.asarray(image, dtype=np.float64)
    output_shape = tuple(int(s) for s in output_shape)
    if image.ndim != len(output_shape):
        raise ValueError("output_shape must have one entry per image axis")
    factors = np.divide(image.shape, output_shape)
    if np.any(factors > 1):  # anti_aliasing defaults to True when down-sampling
        sigma = np.maximum(0, (factors - 1) / 2)
        image = ndi.gaussian_filter(image, sigma, cval=0, mode="reflect")
    zoom = [1 / f for f in factors]
    return ndi.zoom(image, zoom, order=1, mode="reflect", cval=0, grid_mode=True)


def load_data(path, max_xy_size: int):
    """-> (spc (t, l, x~, y~), cmos (z, x, y) resized to max_xy_size, time_axis, spectral_axis);
    same transposes as utils.py:420-422 (note the x/y swap of the final (2, 1, 0))."""
    data = np.load(path, allow_pickle=False)
    spc = data["spc"]
    cmos = np.transpose(data["cmos"], (1, 2, 0))
    cmos = resize_like_skimage(cmos, (max_xy_size, max_xy_size, cmos.shape[2]))
    cmos = np.transpose(cmos, (2, 1, 0))
    return spc, cmos, data["time_axis"], data["spectral_axis"]
