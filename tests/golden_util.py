"""Deterministic parameter/input fills shared by ``oracle/gen_golden.py`` (which runs the
reference in the build container) and the tests (which run the oracle / HIP path).

Weights are a pure function of (state-dict key, shape), so fixtures only need to store
outputs -- the same fill is applied to the reference module when the golden vector is
generated and to the oracle / product module when it is checked.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rng(tag: str) -> np.random.Generator:
    return np.random.default_rng(zlib.crc32(tag.encode()) & 0xFFFFFFFF)


def det_tensor(tag: str, shape, scale: float = 1.0, shift: float = 0.0) -> torch.Tensor:
    a = _rng(tag).standard_normal(size=tuple(shape)).astype(np.float32) * scale + shift
    return torch.from_numpy(a)


def det_fill_(module: torch.nn.Module, salt: str = "") -> None:
    """In-place deterministic fill of every float parameter (buffers untouched).
    1-D 'weight' of norms ~ 1 + 0.1 n, biases ~ 0.1 n, matrices ~ n / sqrt(fan_in),
    bias tables ~ 0.5 n."""
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            tag = salt + name
            if name.endswith("relative_position_bias_table"):
                v = det_tensor(tag, p.shape, 0.5)
            elif p.dim() == 1 and name.endswith("weight"):
                v = det_tensor(tag, p.shape, 0.1, 1.0)
            elif p.dim() == 1:
                v = det_tensor(tag, p.shape, 0.1)
            else:
                fan_in = int(np.prod(p.shape[1:]))
                v = det_tensor(tag, p.shape, 1.0 / np.sqrt(fan_in))
            p.copy_(v)


def sw_predictor(model_in):
    """weight-free deterministic predictor for the sliding-window goldens: uses the window, the per-window relative
    centres (incl. the reference's unsqueeze quirk at sw_batch_size == 1) and the affine; 2 output classes"""
    win, centers, aff = model_in
    win = win.float()
    c = centers.reshape(-1, 3)[:win.shape[0]].to(win.device)
    c0 = win[:, 0] * 0.5 + c[:, 0].view(-1, 1, 1, 1)
    c1 = win[:, 0].abs() + (2.0 * c[:, 1] + 3.0 * c[:, 2]).view(-1, 1, 1, 1) + aff.to(win.device).sum() * 0.125
    return torch.stack([c0, c1], dim=1)


SW_CASES = [  # tag, volume, roi, sw_batch, overlap, mode, cval
    ("pad", (1, 1, 20, 30, 28), (24, 24, 24), 2, 0.5, "gaussian", -1.5),       # volume smaller than the roi in one dim
    ("noncubic", (1, 1, 40, 28, 52), (24, 16, 32), 4, 0.25, "constant", 0.0),
    ("sb1", (1, 1, 36, 36, 36), (24, 24, 24), 1, 0.5, "gaussian", 0.0),        # centers.unsqueeze(0) quirk
    ("batch2", (2, 1, 30, 26, 34), (16, 16, 16), 3, 0.5, "gaussian", 0.25),    # two volumes, short last batch
]
