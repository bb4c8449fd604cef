"""Minimal stand-in for the two torchvision entry points the reference hot path uses
(`Resize`, `resize`, `InterpolationMode`), forwarding to torch.nn.functional.interpolate,
which is what torchvision's tensor code path itself calls.  Test infrastructure only:
used by tools/make_golden.py in the build container to import /root/reference; never
shipped to, or imported on, the GPU box and never imported by the product package."""
import enum

import torch
from torch.nn.functional import interpolate


class InterpolationMode(enum.Enum):
    NEAREST = "nearest"
    BILINEAR = "bilinear"
    BICUBIC = "bicubic"


def _resize(img, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=True):
    size = list(size)
    if list(img.shape[-2:]) == size:
        return img
    squeeze = img.ndim < 4
    if squeeze:
        img = img.unsqueeze(0)
    out = interpolate(img, size=size, mode=interpolation.value, align_corners=False,
                      antialias=bool(antialias))
    return out.squeeze(0) if squeeze else out


class Resize(torch.nn.Module):
    def __init__(self, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=True):
        super().__init__()
        self.size, self.interpolation, self.antialias = size, interpolation, antialias

    def forward(self, img):
        return _resize(img, self.size, self.interpolation, None, self.antialias)
