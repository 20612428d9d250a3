"""Device-side training patches from volumes cached in HBM.

The reference crops and augments on the CPU with MONAI transforms and ``num_workers = 0``
(``/root/reference/data/dataset_builder.py:108-193``, hand-off ``run_training.py:59-66``); at hundreds of 96^3 patches per
second per GPU that chain is the bottleneck directly in front of the hot path.  Here the whole (normalised) volume and its
label map stay on the device (288 GB of HBM hold hundreds of CT volumes), the random draws of one batch -- crop centre
per patch (foreground / background voxel picked with the reference's pos : neg odds), three flip coins, a quarter-turn
count, intensity shift and scale -- are made on the host with a seeded ``numpy`` generator, and ONE gather kernel
(``msseg_aug_crop_batch``) writes the batch.  ``iter(DevicePatchLoader)`` yields the batch dict the engine consumes,
including the crop centre record the reference's transform fork adds (``data/transforms.py:411``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import hip


class AugRow(C.Structure):
    """mirror of msseg_aug_row (include/msseg.h)"""
    _fields_ = [("z0", C.c_int32), ("y0", C.c_int32), ("x0", C.c_int32), ("flips", C.c_int32), ("rotk", C.c_int32),
                ("pad0", C.c_int32), ("shift", C.c_float), ("scale", C.c_float)]


def correct_crop_center(center, roi, img_size):
    """keep the roi inside the image (MONAI correct_crop_centers semantics)"""
    out = []
    for c, r, n in zip(center, roi, img_size):
        lo = r // 2
        hi = n + 1 - r / 2.0
        hi = int(np.floor(hi)) if hi == int(hi) else int(np.ceil(hi))
        if lo == hi:
            hi += 1
        c = int(c)
        c = lo if c < lo else c
        c = hi - 1 if c >= hi else c
        out.append(c)
    return tuple(out)


def draw_rows(rng: np.random.Generator, n, fg_count, bg_count, cfg_like):
    """the random draws of n patches: (use_fg, index into the fg/bg voxel list, flips, rotk, shift, scale) per patch"""
    pos, neg = float(cfg_like["pos"]), float(cfg_like["neg"])
    rows = []
    for _ in range(n):
        use_fg = (rng.random() < pos / (pos + neg)) if fg_count > 0 and bg_count > 0 else fg_count > 0
        idx = int(rng.integers(0, fg_count if use_fg else max(bg_count, 1)))
        flips = tuple(bool(rng.random() < cfg_like["flip_prob"]) for _ in range(3))
        rotk = int(rng.integers(1, 4)) if rng.random() < cfg_like["rot_prob"] else 0
        shift = float(rng.uniform(-cfg_like["shift_os"], cfg_like["shift_os"])) if rng.random() < cfg_like["shift_prob"] else 0.0
        scale = 1.0 + (float(rng.uniform(-cfg_like["scale_f"], cfg_like["scale_f"])) if rng.random() < cfg_like["scale_prob"] else 0.0)
        rows.append((use_fg, idx, flips, rotk, shift, scale))
    return rows


class DevicePatchLoader:
    """len() batches of `batch` patches of roi^3 from ONE cached volume; deterministic per seed."""

    def __init__(self, image: torch.Tensor, label: torch.Tensor, roi: int, batch: int, n_batches: int, device, seed=13,
                 pos=1.0, neg=1.0, flip_prob=0.0, rot_prob=0.0, shift_os=0.1, shift_prob=0.0, scale_f=0.1, scale_prob=0.0,
                 image_threshold=0.0, out_dtype=torch.float32):
        if image.dim() != 4 or label.dim() != 3:
            raise ValueError("image [C, D, H, W] and label [D, H, W] expected")
        self.img = image.to(device=device, dtype=torch.float32).contiguous()
        self.lab = label.to(device=device, dtype=torch.uint8).contiguous()
        self.roi, self.batch, self.n, self.dev, self.out_dtype = int(roi), int(batch), int(n_batches), device, out_dtype
        self.cfg = dict(pos=pos, neg=neg, flip_prob=flip_prob, rot_prob=rot_prob, shift_os=shift_os, shift_prob=shift_prob,
                        scale_f=scale_f, scale_prob=scale_prob)
        self.rng = np.random.default_rng(seed)
        # voxel lists of the crop-centre candidates (once per cached volume; index work, stays on the device)
        flat_lab = self.lab.reshape(-1)
        self.fg = torch.nonzero(flat_lab > 0).reshape(-1)
        self.bg = torch.nonzero((flat_lab == 0) & (self.img[0].reshape(-1) > image_threshold)).reshape(-1)
        self.last_rows = None

    def __len__(self):
        return self.n

    def rows_for(self, draws):
        """draws -> AugRow list (+ the corrected centres); the centre lookup is one tiny gather from the voxel lists"""
        D, H, W = self.lab.shape
        sel = torch.stack([(self.fg if d[0] else self.bg)[d[1]] for d in draws]).cpu().tolist()
        rows, centers = [], []
        for flat, (use_fg, idx, flips, rotk, shift, scale) in zip(sel, draws):
            c = (flat // (H * W), (flat // W) % H, flat % W)
            c = correct_crop_center(c, (self.roi,) * 3, (D, H, W))
            z0, y0, x0 = (v - self.roi // 2 for v in c)
            rows.append(AugRow(z0, y0, x0, int(flips[0]) | int(flips[1]) << 1 | int(flips[2]) << 2, rotk, 0, shift, scale))
            centers.append(c)
        return rows, centers

    def __iter__(self):
        D, H, W = self.lab.shape
        Cc = self.img.shape[0]
        for _ in range(self.n):
            draws = draw_rows(self.rng, self.batch, int(self.fg.numel()), int(self.bg.numel()), self.cfg)
            rows, centers = self.rows_for(draws)
            self.last_rows = rows
            host = torch.frombuffer(bytearray(bytes((AugRow * len(rows))(*rows))), dtype=torch.uint8)
            table = host.to(self.dev, non_blocking=True)
            img = torch.empty(self.batch, Cc, self.roi, self.roi, self.roi, dtype=self.out_dtype, device=self.dev)
            lab = torch.empty(self.batch, 1, self.roi, self.roi, self.roi, dtype=torch.float32, device=self.dev)
            hip.aug_crop_batch(self.img, self.lab, table, img, lab, self.roi)
            aff = torch.eye(4)[None].repeat(self.batch, 1, 1)
            cen = torch.tensor(centers, dtype=torch.float32)
            yield {"image": img, "label": lab,
                   "image_meta_dict": {"original_affine": aff, "affine": aff.clone(),
                                       "filename_or_obj": [f"device_cache_{j}" for j in range(self.batch)]},
                   "label_meta_dict": {"affine": aff.clone()},
                   "image_transforms": [{"class": ["RandCropByPosNegLabeld"] * self.batch,
                                         "orig_size": [torch.full((self.batch,), float(s)) for s in (D, H, W)],
                                         "extra_info": {"center": [cen[:, 0], cen[:, 1], cen[:, 2]]}}]}
