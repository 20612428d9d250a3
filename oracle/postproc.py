"""Oracle (test infrastructure): post-inference steps, numpy on the CPU.

* ``argmax_labels``   -- ``/root/reference/engine/test.py:140-141``: ``softmax(outputs, 1)`` -> ``np.argmax(axis=1).astype(uint8)``.
* ``resample_nearest`` -- ``/root/reference/utils/misc.py:420-425``: ``scipy.ndimage.zoom(img, target/shape, order=0,
  prefilter=False)``; restated as index arithmetic (input coordinate = o * (in - 1) / (out - 1) in double, ``floor(c + 0.5)``),
  pinned by ``tests/golden/resample.npz`` (the reference's function run in the build container) and against scipy itself.
* ``majority_vote``   -- ``/root/reference/majority_vote.py:23-37`` (cannot be imported: module-level argparse + nibabel):
  votes of the foreground classes, background starts with one vote, ``np.argmax`` (first maximum).  Parity unpinned by a
  reference run; pinned by known answers.
"""
from __future__ import annotations

import numpy as np


def argmax_labels(logits: np.ndarray) -> np.ndarray:
    """logits [C, D, H, W] fp32 -> uint8 [D, H, W] through the reference's fp32 softmax"""
    x = logits.astype(np.float32)
    e = np.exp(x - x.max(axis=0, keepdims=True))
    p = e / e.sum(axis=0, keepdims=True)
    return np.argmax(p, axis=0).astype(np.uint8)


def resample_nearest(img: np.ndarray, target_size) -> np.ndarray:
    idx = []
    for n, t in zip(img.shape, target_size):
        o = np.arange(int(t), dtype=np.float64)
        c = o * (float(n - 1) / float(t - 1)) if t > 1 else np.zeros(1)
        idx.append(np.clip(np.floor(c + 0.5).astype(np.int64), 0, n - 1))
    return img[np.ix_(*idx)]


def majority_vote(fold_labels: np.ndarray, n_classes: int) -> np.ndarray:
    """fold_labels [F, D, H, W] -> [D, H, W]"""
    votes = np.zeros((n_classes,) + fold_labels.shape[1:], dtype=np.uint8)
    for f in range(fold_labels.shape[0]):
        for c in range(1, n_classes):
            votes[c] += (fold_labels[f] == c).astype(np.uint8)
    votes[0] = votes[0] + 1
    return np.argmax(votes, axis=0).astype(np.uint8)
