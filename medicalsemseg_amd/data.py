"""Synthetic data with the batch-dict layout the engine consumes: keys ``image``, ``label``,
``image_meta_dict['original_affine' | 'affine' | 'filename_or_obj']``, ``label_meta_dict['affine']``,
``image_transforms`` (list of dicts with ``class``, ``orig_size``, ``extra_info['center']``).  The reference's MONAI
transform / CacheDataset pipeline (``/root/reference/data/*``) is CPU-side I/O outside the hot-path scope; benchmark
and plumbing runs use these generators (random volumes, nested-sphere labels so Dice is non-degenerate)."""
from __future__ import annotations

import torch


def sphere_labels(size, n_cls):
    sz = (size,) * 3 if isinstance(size, int) else tuple(size)
    ax = [torch.linspace(-1, 1, s) for s in sz]
    r = (ax[0][:, None, None] ** 2 + ax[1][None, :, None] ** 2 + ax[2][None, None, :] ** 2).sqrt()
    y = torch.zeros(sz)
    for c in range(1, n_cls):
        y[r < 0.9 * (n_cls - c) / max(n_cls - 1, 1)] = c
    return y


class SyntheticLoader:
    """len() batches of `batch` random volumes of `size`^3 with `in_chans` channels; deterministic per (seed, index)."""

    def __init__(self, n_batches, batch, size, in_chans, n_cls, seed=13, with_crop_info=True):
        self.n, self.batch, self.size, self.in_chans, self.n_cls, self.seed = n_batches, batch, size, in_chans, n_cls, seed
        self.with_crop_info = with_crop_info
        self.labels = sphere_labels(size, n_cls)

    def __len__(self):
        return self.n

    def __iter__(self):
        sz = (self.size,) * 3 if isinstance(self.size, int) else tuple(self.size)
        for i in range(self.n):
            g = torch.Generator().manual_seed(self.seed * 100003 + i)
            img = torch.randn(self.batch, self.in_chans, *sz, generator=g)
            lab = self.labels[None, None].repeat(self.batch, 1, 1, 1, 1)
            img = img + 0.5 * lab  # make the task learnable
            aff = torch.eye(4)[None].repeat(self.batch, 1, 1)
            b = {"image": img, "label": lab,
                 "image_meta_dict": {"original_affine": aff, "affine": aff.clone(),
                                     "filename_or_obj": [f"synthetic_{i}_{j}" for j in range(self.batch)]},
                 "label_meta_dict": {"affine": aff.clone()},
                 "image_transforms": []}
            if self.with_crop_info:
                b["image_transforms"].append({
                    "class": ["RandCropByPosNegLabeld"] * self.batch,
                    "orig_size": [torch.full((self.batch,), float(2 * s)) for s in sz],
                    "extra_info": {"center": [torch.full((self.batch,), float(s)) for s in sz]}})
            yield b
