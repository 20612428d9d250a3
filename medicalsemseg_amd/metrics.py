"""Evaluation metrics that the reference takes from MONAI, on the device.

``hausdorff95``: ``HausdorffDistanceMetric(include_background=True, percentile=95, reduction="mean", get_not_nans=True)``
as ``/root/reference/engine/test.py:31,48-51,64`` builds, feeds (arg-max one-hot prediction, one-hot label) and reduces it.
MONAI's route is CPU scipy per (sample, class): bounding-box crop, ``binary_erosion`` XOR for the surfaces,
``distance_transform_edt`` of the whole box, a fancy-index read, ``np.percentile``.  Here the label maps never leave the
GPU: one pass marks the surfaces of all classes of both maps (``msseg_hd_edges``), then per class and direction three integer
line passes give the exact squared distance of every surface voxel to the other surface and a histogram of them
(``msseg_hd_directed_hist``); the host reads the two order statistics the percentile interpolates between from the
histogram (a few KB) and finishes in double -- the same values scipy / numpy produce, bit for bit (tests/test_gpu_engine.py
compares with ``oracle/postproc.py``, which calls scipy).  No CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch

from . import hip


def _percentile_from_hist(counts: np.ndarray, q: float) -> float:
    """np.percentile(sqrt(d2) for every counted d2, q) (method 'linear', numpy's own arithmetic) from counts[d2]"""
    n = int(counts.sum())
    cum = np.cumsum(counts)
    quant = np.true_divide(q, 100.0)
    vi = (n - 1) * quant                      # numpy: get_virtual_index of the 'linear' method
    prev = int(np.floor(vi))
    nxt = min(prev + 1, n - 1)
    prev = max(min(prev, n - 1), 0)
    a = np.sqrt(np.float64(int(np.searchsorted(cum, prev + 1, side="left"))))    # order statistic `prev` (0-based)
    b = np.sqrt(np.float64(int(np.searchsorted(cum, nxt + 1, side="left"))))
    t = np.float64(vi - np.floor(vi))
    d = b - a
    r = a + d * t
    if t >= 0.5:                               # numpy's _lerp
        r = b - d * (1 - t)
    return float(r)


def hausdorff95(pred: torch.Tensor, label: torch.Tensor, n_classes: int, percentile: float = 95.0) -> np.ndarray:
    """pred, label: integer label maps [B, D, H, W] (or [B, 1, D, H, W]) on the GPU -> hd [B, n_classes] float64 (host):
    NaN where neither map has the class and, with numpy >= 1.22's percentile, also where only one has it (MONAI's
    get_surface_distance yields all-inf distances there; see below)."""
    if not pred.is_cuda or not label.is_cuda:
        raise RuntimeError("hausdorff95 runs on the GPU only (no CPU fallback; the scipy restatement is oracle/postproc.py)")
    if pred.dim() == 5:
        pred = pred[:, 0]
    if label.dim() == 5:
        label = label[:, 0]
    pred = pred.to(torch.uint8).contiguous()
    label = label.to(torch.uint8).contiguous()
    B, D, H, W = pred.shape
    hd = np.full((B, n_classes), np.nan, dtype=np.float64)
    hist = None
    for b in range(B):
        ep, eg, stats = hip.hd_edges(pred[b], label[b], n_classes)
        st = stats.cpu().numpy()
        for c in range(n_classes):
            n_p, n_g = int(st[c, 6]), int(st[c, 7])
            if n_p == 0 and n_g == 0:
                continue                                  # class absent from both maps: NaN
            if n_p == 0 or n_g == 0:
                # one surface empty: MONAI's get_surface_distance hands np.percentile an all-inf array in both directions.
                # numpy (>= 1.22, incl. the 2.2 here) interpolates as a + (b - a) * t: inf - inf -> NaN, so the class drops
                # out of the mean like an absent one (older numpy gave inf); the same arithmetic, not a constant:
                with np.errstate(invalid="ignore"):
                    hd[b, c] = float(np.float64(np.inf) + (np.float64(np.inf) - np.float64(np.inf)) * 0.5)
                continue
            box = (int(st[c, 0]), int(st[c, 1]), int(st[c, 2]), int(st[c, 3]) + 1, int(st[c, 4]) + 1, int(st[c, 5]) + 1)
            bz, by, bx = box[3] - box[0], box[4] - box[1], box[5] - box[2]
            nbins = (bz - 1) ** 2 + (by - 1) ** 2 + (bx - 1) ** 2 + 2
            if hist is None or hist.numel() < nbins:
                hist = torch.empty(max(nbins, 1 << 16), dtype=torch.int32, device=pred.device)
            d = []
            for src, tgt in ((eg, ep), (ep, eg)):         # pred -> gt, then gt -> pred (compute_hausdorff_distance)
                hip.hd_directed_hist(src, tgt, c, box, hist[:nbins])
                counts = hist[:nbins].cpu().numpy().astype(np.int64)
                assert counts[-1] == 0, "a surface voxel found no voxel of a non-empty surface"
                d.append(_percentile_from_hist(counts[:-1], percentile))
            hd[b, c] = max(d)
    return hd


def hausdorff_mean(hd: np.ndarray):
    """MONAI's do_metric_reduction(f, "mean") with get_not_nans: mean over the classes that are not NaN, then over the
    batch entries that have any -> (value, not_nans) as haus_dist_metric.aggregate() returns them"""
    f = np.asarray(hd, dtype=np.float64).copy()
    nans = np.isnan(f)
    f[nans] = 0.0
    nn = (~nans).sum(1).astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        per = np.where(nn > 0, f.sum(1) / np.maximum(nn, 1.0), 0.0)
    nb = float((nn > 0).sum())
    return (float(per.sum() / nb) if nb > 0 else 0.0), nb
