"""Oracle (test infrastructure): the training crop + augmentation chain as plain numpy.

Restates, per patch and from an explicit parameter row (so that the random draws are shared with the product),
``RandCropByPosNegLabeld`` (centre -> crop, incl. MONAI's ``correct_crop_centers`` clamp), ``RandFlipd`` on axes 0/1/2,
``RandRotate90d`` (``np.rot90`` in the (0, 1) plane), ``RandShiftIntensityd`` (``img + offset``) and
``RandScaleIntensityd`` (``img * (1 + factor)``) as the reference chains them in
``/root/reference/data/dataset_builder.py:108-193`` (its crop fork records the centre, ``data/transforms.py:411``).
MONAI is absent: parity with MONAI's own random stream is unpinned; what is pinned is that the product's device kernel
reproduces this restatement bit for bit for the same rows.
"""
from __future__ import annotations

import numpy as np


def correct_crop_center(center, roi, img_size):
    """MONAI correct_crop_centers: keep the whole roi inside the image"""
    out = []
    for c, r, n in zip(center, roi, img_size):
        lo = r // 2
        hi = n + 1 - r / 2.0
        hi = int(np.floor(hi)) if hi == int(hi) else int(np.ceil(hi))   # first centre that would reach past the image
        if lo == hi:
            hi += 1
        c = int(c)
        c = lo if c < lo else c
        c = hi - 1 if c >= hi else c
        out.append(c)
    return tuple(out)


def crop_start(center, roi):
    return tuple(int(c) - r // 2 for c, r in zip(center, roi))


def apply_row(img: np.ndarray, lab, start, roi: int, flips, rotk: int, shift: float, scale: float):
    """img [C, D, H, W] fp32, lab [D, H, W] or None -> (patch_img [C, R, R, R], patch_lab [R, R, R] fp32 or None)"""
    z0, y0, x0 = start
    p = img[:, z0:z0 + roi, y0:y0 + roi, x0:x0 + roi].astype(np.float32)
    q = lab[z0:z0 + roi, y0:y0 + roi, x0:x0 + roi] if lab is not None else None
    for ax, f in enumerate(flips):
        if f:
            p = np.flip(p, axis=ax + 1)
            q = np.flip(q, axis=ax) if q is not None else None
    if rotk:
        p = np.rot90(p, rotk, axes=(1, 2))
        q = np.rot90(q, rotk, axes=(0, 1)) if q is not None else None
    p = (p + np.float32(shift)).astype(np.float32) * np.float32(scale)
    return np.ascontiguousarray(p.astype(np.float32)), (np.ascontiguousarray(q).astype(np.float32) if q is not None else None)
