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


def hausdorff95(pred_onehot: np.ndarray, gt_onehot: np.ndarray, percentile: float = 95.0) -> np.ndarray:
    """``HausdorffDistanceMetric(include_background=True, percentile=95)`` as ``/root/reference/engine/test.py:31,48-50`` builds
    and calls it: pred / gt one-hot [B, C, D, H, W] -> hd[B, C].  MONAI (0.8.x) is not installed and the reference vendors
    none of it: restated from MONAI's published ``compute_hausdorff_distance`` (**parity unpinned** against MONAI) on the
    scipy.ndimage functions MONAI itself calls there:

    * ``get_mask_edges``: edges = ``binary_erosion(mask) ^ mask`` (scipy's default 6-neighbour structure, border value 0,
      after a crop to the bounding box of ``pred | gt``, which changes neither edge set nor distances);
    * ``get_surface_distance``: ``distance_transform_edt(~edges_other)`` (voxel units) read at this mask's edge voxels;
      ``inf`` for every voxel when either edge set is empty (``np.percentile`` of an all-inf array is NaN with numpy >= 1.22:
      ``inf - inf`` in its interpolation; older numpy returned inf), an EMPTY array when both are (-> NaN);
    * ``np.percentile(distances, 95)`` per direction, the larger of the two directions."""
    from scipy.ndimage import binary_erosion, distance_transform_edt

    def edges(m):
        return binary_erosion(m) ^ m

    def directed(e1, e2):
        if not e2.any():
            d = np.full(int(e1.sum()), np.inf)
        elif not e1.any():
            d = np.full(int(e2.sum()), np.inf)
        else:
            d = distance_transform_edt(~e2)[e1]
        if d.shape == (0,):
            return np.nan
        return float(np.percentile(d, percentile))

    B, C = pred_onehot.shape[:2]
    hd = np.empty((B, C), dtype=np.float64)
    for b in range(B):
        for c in range(C):
            p, g = pred_onehot[b, c].astype(bool), gt_onehot[b, c].astype(bool)
            if not (p | g).any():
                hd[b, c] = np.nan
                continue
            ep, eg = edges(p), edges(g)
            d1, d2 = directed(ep, eg), directed(eg, ep)
            hd[b, c] = np.nan if (np.isnan(d1) or np.isnan(d2)) else max(d1, d2)
    return hd


def hausdorff_mean(hd: np.ndarray):
    """MONAI ``do_metric_reduction(f, "mean")`` + ``get_not_nans``: mean over the classes that are not NaN, then over the
    batch entries that have any; returns (value, not_nans)"""
    f = hd.astype(np.float64).copy()
    nans = np.isnan(f)
    f[nans] = 0.0
    nn = (~nans).sum(1).astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        per = np.where(nn > 0, f.sum(1) / np.maximum(nn, 1), 0.0)
    nb = float((nn > 0).sum())
    return (float(per.sum() / nb) if nb > 0 else 0.0), nb
