"""Image preprocessing in front of the path with the arithmetic on the device.

DPNModel._preprocess (AdvancedModel.py:487-498) runs transforms.Resize((H, W)) -> ToTensor -> Normalize(ImageNet)
on the host for every image and DPNModel.fit resizes the annotated images with PIL's default (nearest) filter
(AdvancedModel.py:265).  Here only the raw 8-bit pixels go to the device; resizing, scaling and normalisation run
there (mfl_preprocess_rgb8, mfl_resize_labels) and the result is the [B, 3, H, W] tensor the model takes.

Resize on a PIL image is Pillow's resampling: `resample_tables` restates how Pillow derives its windows and
fixed-point coefficients (double precision, host side, a few kilobytes per axis), the kernels apply them in the same
integer arithmetic, so the resized 8-bit image is Pillow's bit for bit.
"""
from __future__ import annotations

import ctypes
import math
from typing import Iterable, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import lib, ptr

IMAGENET_MEAN = (0.485, 0.456, 0.406)       # AdvancedModel.py:492
IMAGENET_STD = (0.229, 0.224, 0.225)
PRECISION_BITS = 32 - 8 - 2                 # Pillow's fixed point for 8-bit channels


def resample_tables(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Windows and fixed-point coefficients of Pillow's bilinear resampling along one axis: (coef int32
    [out_size, ksize], bounds int32 [out_size, 2] = first input index and count).  The triangle filter is widened
    by the scale factor when shrinking (antialiasing), windows are clipped to the image and renormalised."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    coef = np.zeros((out_size, ksize), np.float64)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ww = 0.0
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - v if v < 1.0 else 0.0
            coef[xx, x] = w
            ww += w
        if ww != 0.0:
            coef[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    fixed = np.where(coef < 0, -0.5 + coef * (1 << PRECISION_BITS), 0.5 + coef * (1 << PRECISION_BITS))
    return np.trunc(fixed).astype(np.int32), bounds


def nearest_indices(in_size: int, out_size: int) -> np.ndarray:
    """Source index per output position of Pillow's nearest-neighbour resize (the running sum of the step, as
    Pillow accumulates it)."""
    step = in_size / out_size
    pos = step * 0.5
    out = np.empty(out_size, np.int32)
    for x in range(out_size):
        out[x] = min(int(pos), in_size - 1)
        pos += step
    return out


def _as_rgb8(image) -> np.ndarray:
    if hasattr(image, "convert"):                       # a PIL image
        image = np.asarray(image.convert("RGB"))
    a = np.ascontiguousarray(image)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("images must be PIL images or uint8 arrays [H, W, 3]")
    return a


def preprocess(images: Iterable, height: int, width: int, device="cuda", mean: Sequence[float] = IMAGENET_MEAN,
               std: Sequence[float] = IMAGENET_STD) -> torch.Tensor:
    """[B, 3, height, width] fp32 on `device`: Resize((height, width)) -> ToTensor -> Normalize(mean, std) of each
    image (AdvancedModel.py:487-498), computed on the device from the raw 8-bit pixels."""
    images = [_as_rgb8(im) for im in images]
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("semantic_segmentation_amd runs on a ROCm GPU only (there is no CPU fallback)")
    out = torch.empty(len(images), 3, height, width, dtype=torch.float32, device=device)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    tables = {}
    with torch.cuda.device(device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        for b, im in enumerate(images):
            H0, W0, _ = im.shape
            key = (H0, W0)
            if key not in tables:
                tx = None if W0 == width else [torch.from_numpy(t).to(device) for t in resample_tables(W0, width)]
                ty = None if H0 == height else [torch.from_numpy(t).to(device) for t in resample_tables(H0, height)]
                tables[key] = (tx, ty)
            tx, ty = tables[key]
            src = torch.from_numpy(im).to(device)
            n = int(lib.mfl_preprocess_workspace_bytes(H0, W0, height, width))
            ws = torch.empty(n, dtype=torch.uint8, device=device)
            _lib.check(lib.mfl_preprocess_rgb8(ptr(src), H0, W0, ptr(tx[0]) if tx else None, ptr(tx[1]) if tx else None,
                                               tx[0].shape[1] if tx else 0, ptr(ty[0]) if ty else None,
                                               ptr(ty[1]) if ty else None, ty[0].shape[1] if ty else 0, m, s, ptr(out[b]),
                                               height, width, ptr(ws), n, stream), "mfl_preprocess_rgb8")
    return out


def resize_labels(label_maps: Iterable, height: int, width: int, device="cuda") -> torch.Tensor:
    """[B, height, width] uint8 on `device`: the annotated images resized with the nearest filter
    (AdvancedModel.py:265; labels such as 255 = void pass through unchanged)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("semantic_segmentation_amd runs on a ROCm GPU only (there is no CPU fallback)")
    maps = [np.ascontiguousarray(np.asarray(m)) for m in label_maps]
    out = torch.empty(len(maps), height, width, dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        for b, a in enumerate(maps):
            if a.dtype != np.uint8 or a.ndim != 2:
                raise ValueError("label maps must be uint8 [H, W]")
            H0, W0 = a.shape
            yin = torch.from_numpy(nearest_indices(H0, height)).to(device)
            xin = torch.from_numpy(nearest_indices(W0, width)).to(device)
            src = torch.from_numpy(a).to(device)
            _lib.check(lib.mfl_resize_labels(ptr(src), H0, W0, ptr(yin), ptr(xin), ptr(out[b]), height, width, stream),
                       "mfl_resize_labels")
    return out
