"""Oracle (test infrastructure): sliding-window inference, fp32 torch-CPU.

Follows ``/root/reference/engine/utils.py:19-159`` (the reference's fork of
MONAI's ``sliding_window_inference`` that feeds ``(window, centers, affine)``
tuples to the predictor) and restates the MONAI helpers it imports
(``_get_scan_interval``, ``dense_patch_slices``, ``compute_importance_map``,
``get_valid_patch_size``).  MONAI absent -> those helpers are
parity unpinned vs MONAI; known answers (1000 windows / last start 416 at
512^3, symmetric importance map with max 1) are tested.
"""
from __future__ import annotations

import itertools
import math
from typing import Callable, List, Sequence

import torch
import torch.nn.functional as F


def _tup(v, n):
    if isinstance(v, (int, float)):
        return (v,) * n
    v = tuple(v)
    if len(v) != n:
        raise ValueError(f"expected {n} values, got {v}")
    return v


def fall_back_tuple(roi, image):
    roi = _tup(roi, len(image))
    return tuple(int(r) if (r is not None and r > 0) else int(i) for r, i in zip(roi, image))


def get_scan_interval(image_size, roi_size, num_spatial_dims, overlap):
    out = []
    for i in range(num_spatial_dims):
        if roi_size[i] == image_size[i]:
            out.append(int(roi_size[i]))
        else:
            interval = int(roi_size[i] * (1 - overlap))
            out.append(interval if interval > 0 else 1)
    return tuple(out)


def dense_patch_starts(image_size, patch_size, scan_interval):
    """Per-dim start lists; windows are their row-major product (first dim slowest)."""
    starts = []
    for d in range(len(image_size)):
        if scan_interval[d] == 0:
            num = 1
        else:
            num = int(math.ceil(float(image_size[d]) / scan_interval[d]))
            cnt = min(num, image_size[d] - patch_size[d] + 1)
            for x in range(num):
                if x * scan_interval[d] + patch_size[d] >= image_size[d]:
                    cnt = x + 1
                    break
            num = cnt
        dim_starts = []
        for x in range(num):
            s = x * scan_interval[d]
            s -= max(s + patch_size[d] - image_size[d], 0)
            dim_starts.append(s)
        starts.append(dim_starts)
    return starts


def dense_patch_slices(image_size, patch_size, scan_interval):
    starts = dense_patch_starts(image_size, patch_size, scan_interval)
    return [tuple(slice(s, s + patch_size[d]) for d, s in enumerate(st))
            for st in itertools.product(*starts)]


def _gauss_1d(sigma: float) -> torch.Tensor:
    """erf-integrated 1-D Gaussian, truncated=4, un-normalised sum (MONAI 'erf')."""
    tail = int(max(float(sigma) * 4.0, 0.5) + 0.5)
    x = torch.arange(-tail, tail + 1, dtype=torch.float32)
    t = 0.70710678 / abs(float(sigma))
    out = 0.5 * ((t * (x + 0.5)).erf() - (t * (x - 0.5)).erf())
    return out.clamp(min=0)


def compute_importance_map(patch_size, mode="constant", sigma_scale=0.125):
    patch_size = tuple(int(p) for p in patch_size)
    if mode == "constant":
        return torch.ones(patch_size, dtype=torch.float32)
    if mode != "gaussian":
        raise ValueError(f"unsupported blend mode {mode}")
    sig = _tup(sigma_scale, len(patch_size))
    sigmas = [p * s for p, s in zip(patch_size, sig)]
    imp = torch.zeros(patch_size, dtype=torch.float32)
    imp[tuple(p // 2 for p in patch_size)] = 1.0
    # separable zero-padded "same" convolution of the delta with each 1-D kernel
    for d, s in enumerate(sigmas):
        k = _gauss_1d(s)
        r = (k.numel() - 1) // 2
        x = imp.movedim(d, -1)
        shp = x.shape
        x = F.conv1d(x.reshape(-1, 1, shp[-1]), k.view(1, 1, -1), padding=r)
        imp = x.reshape(shp).movedim(-1, d)
    imp = imp / imp.max()
    min_non_zero = imp[imp != 0].min().item()
    return torch.clamp(imp, min=min_non_zero)


def get_valid_patch_size(image_size, patch_size):
    ps = fall_back_tuple(patch_size, image_size)
    return tuple(min(p, i) for p, i in zip(ps, image_size))


def sliding_window_inference(inputs: torch.Tensor, affine, roi_size, sw_batch_size: int,
                             predictor: Callable, overlap: float = 0.25, mode: str = "constant",
                             sigma_scale=0.125, padding_mode: str = "constant", cval: float = 0.0,
                             sw_device=None, device=None, *args, **kwargs) -> torch.Tensor:
    nsd = inputs.dim() - 2
    if overlap < 0 or overlap >= 1:
        raise AssertionError("overlap must be >= 0 and < 1.")
    image_size_ = list(inputs.shape[2:])
    batch_size = inputs.shape[0]
    device = device or inputs.device
    sw_device = sw_device or inputs.device
    roi_size = fall_back_tuple(roi_size, image_size_)
    image_size = tuple(max(image_size_[i], roi_size[i]) for i in range(nsd))
    pad_size: List[int] = []
    for k in range(inputs.dim() - 1, 1, -1):
        diff = max(roi_size[k - 2] - inputs.shape[k], 0)
        half = diff // 2
        pad_size.extend([half, diff - half])
    inputs = F.pad(inputs, pad=pad_size, mode=padding_mode, value=cval)
    scan_interval = get_scan_interval(image_size, roi_size, nsd, overlap)
    slices = dense_patch_slices(image_size, roi_size, scan_interval)
    num_win = len(slices)
    total = num_win * batch_size
    imp = compute_importance_map(get_valid_patch_size(image_size, roi_size), mode, sigma_scale).to(device)
    out = cnt = None
    for g in range(0, total, sw_batch_size):
        rng = range(g, min(g + sw_batch_size, total))
        uns = [[slice(idx // num_win, idx // num_win + 1), slice(None)] + list(slices[idx % num_win])
               for idx in rng]
        centers = torch.stack([torch.tensor([(ws[2 + d].stop - roi_size[d] // 2) / image_size[d]
                                             for d in range(3)]) for ws in uns]).float().to(sw_device)
        if sw_batch_size == 1:  # reference quirk, engine/utils.py:131-132
            centers = centers.unsqueeze(0)
        win = torch.cat([inputs[tuple(s)] for s in uns]).to(sw_device)
        seg = predictor((win, centers, affine), *args, **kwargs).to(device)
        if out is None:
            shape = [batch_size, seg.shape[1]] + list(image_size)
            out = torch.zeros(shape, dtype=torch.float32, device=device)
            cnt = torch.zeros(shape, dtype=torch.float32, device=device)
        for idx, s in zip(rng, uns):
            out[tuple(s)] += imp * seg[idx - g]
            cnt[tuple(s)] += imp
    out = out / cnt
    final: List[slice] = []
    for sp in range(nsd):
        final.insert(0, slice(pad_size[sp * 2], image_size_[nsd - sp - 1] + pad_size[sp * 2]))
    while len(final) < out.dim():
        final.insert(0, slice(None))
    return out[tuple(final)]
