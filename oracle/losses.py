"""Oracle (test infrastructure): DiceCE loss and hard Dice metric, fp32 torch-CPU.

Restates MONAI ``DiceCELoss(to_onehot_y=True, softmax=True, squared_pred=True,
smooth_nr, smooth_dr)`` as constructed at
``/root/reference/run_training.py:103-105`` and called at
``/root/reference/engine/train.py:62``; and ``DiceMetric(include_background=True,
reduction="none", get_not_nans=True)`` + ``AsDiscrete`` as used at
``/root/reference/engine/train.py:29-31,89-111``.
MONAI itself is absent -> parity unpinned vs MONAI; known-answer tests in
``tests/test_host_logic.py`` (closed forms of SURVEY.md 8(c)).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def dice_ce_loss(logits: torch.Tensor, labels: torch.Tensor, smooth_nr: float = 1e-5,
                 smooth_dr: float = 1e-5) -> torch.Tensor:
    """logits [B,C,*sp] float, labels [B,1,*sp] (integer valued) -> 0-dim tensor."""
    n_cls = logits.shape[1]
    p = torch.softmax(logits.float(), dim=1)
    lab = labels.long().squeeze(1)
    t = F.one_hot(lab, n_cls).movedim(-1, 1).to(p.dtype)
    red = tuple(range(2, logits.dim()))
    inter = (p * t).sum(red)
    den = (p * p).sum(red) + (t * t).sum(red)
    dice = 1.0 - (2.0 * inter + smooth_nr) / (den + smooth_dr)
    ce = F.cross_entropy(logits.float(), lab)
    return dice.mean() + ce


class DiceCELoss(torch.nn.Module):
    def __init__(self, to_onehot_y=True, softmax=True, squared_pred=True, smooth_nr=1e-5, smooth_dr=1e-5):
        super().__init__()
        if not (to_onehot_y and softmax and squared_pred):
            raise ValueError("oracle restates only the reference's configuration "
                             "(to_onehot_y, softmax, squared_pred all True)")
        self.smooth_nr, self.smooth_dr = float(smooth_nr), float(smooth_dr)

    def forward(self, logits, labels):
        return dice_ce_loss(logits, labels, self.smooth_nr, self.smooth_dr)


def dice_metric(logits: torch.Tensor, labels: torch.Tensor):
    """Hard Dice per (n, c): 2|P&T| / (|P|+|T|), NaN where |T| == 0.

    Returns (scores[B,C], not_nans[B,C]) like ``DiceMetric.aggregate()`` with
    ``reduction="none", get_not_nans=True`` right after one call."""
    n_cls = logits.shape[1]
    pred = logits.argmax(dim=1)
    lab = labels.long().squeeze(1)
    P = F.one_hot(pred, n_cls).movedim(-1, 1).float()
    T = F.one_hot(lab, n_cls).movedim(-1, 1).float()
    red = tuple(range(2, logits.dim()))
    inter = (P * T).sum(red)
    y_o = T.sum(red)
    den = y_o + P.sum(red)
    score = torch.where(y_o > 0, 2.0 * inter / den, torch.full_like(inter, float("nan")))
    return score, (~torch.isnan(score)).float()


def class_means_and_mdice(scores: torch.Tensor, not_nans: torch.Tensor):
    """Per-class nanmean over batch + nanmean over classes,
    as ``/root/reference/engine/train.py:96-106``."""
    n_cls = scores.shape[1]
    class_means = torch.zeros(n_cls)
    for c in range(n_cls):
        if not_nans[:, c].sum() > 0:
            class_means[c] = scores[:, c].nanmean()
        else:
            class_means[c] = float("nan")
    return class_means, class_means.nanmean()
