"""Dice + cross-entropy loss and the hard Dice metric on the fused HIP reductions.

``DiceCELoss`` keeps the call signature of ``monai.losses.DiceCELoss`` as the reference constructs it
(``/root/reference/run_training.py:103-105``: ``to_onehot_y=True, softmax=True, squared_pred=True,
smooth_nr, smooth_dr``) and calls it (``/root/reference/engine/train.py:62``: ``criterion(logits[B,C,...],
labels[B,1,...]) -> 0-dim tensor``).  One pass over logits+labels produces every reduction of the loss AND of
the per-step hard Dice metric (``engine/train.py:89-111``); the backward is a second single pass.
"""
from __future__ import annotations

import torch

from . import hip


import weakref

# gradients this module wrote as channels-last rows [N, *spatial, ld] with ZERO padding channels, keyed by data_ptr:
# a model whose logits are a [B, C, ...] view of such rows (models/unet.py) takes them back without a layout pass
_CL_GRADS = weakref.WeakValueDictionary()


def _channels_last_rows(t: torch.Tensor):
    """ld if `t` [N, C, *spatial] is the channel-first VIEW of dense channels-last rows [N, *spatial, ld] (ld >= C,
    unit channel stride), else 0."""
    if t.dim() < 3 or t.stride(1) != 1 or t.shape[1] == 1:
        return 0
    ld = t.stride(-1)
    exp = ld
    for d in range(t.dim() - 1, 1, -1):
        if t.stride(d) != exp:
            return 0
        exp *= t.shape[d]
    if t.stride(0) != exp or ld < t.shape[1]:
        return 0
    return ld


def channels_last_grad(g: torch.Tensor, ld: int, dtype):
    """[N, *spatial, ld] tensor behind `g` if `g` is a gradient produced by _DiceCEFn.backward in that layout, else None"""
    base = _CL_GRADS.get(g.data_ptr())
    if base is None or base.dtype != dtype or base.shape[-1] != ld or _channels_last_rows(g) != ld:
        return None
    if base.shape[0] != g.shape[0] or tuple(base.shape[1:-1]) != tuple(g.shape[2:]) or base.device != g.device:
        return None
    return base


class _DiceCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, smooth_nr, smooth_dr, holder):
        if not logits.is_cuda:
            raise RuntimeError("DiceCELoss runs on the GPU only (no CPU fallback)")
        if logits.dtype not in (torch.float32, torch.bfloat16):
            logits = logits.float()
        ld = _channels_last_rows(logits)
        if ld == 0 or logits.data_ptr() % 16 or (ld * logits.element_size()) % 16:
            ld = 0
            logits = logits.contiguous()
        labels = labels.contiguous()
        if labels.dtype not in (torch.float32, torch.bfloat16, torch.uint8, torch.int64):
            labels = labels.long()
        N, C = logits.shape[0], logits.shape[1]
        S = logits.numel() // (N * C)
        if labels.numel() != N * S:
            raise ValueError(f"labels {tuple(labels.shape)} do not match logits {tuple(logits.shape)}")
        if N <= 8:   # deterministic two-step reduction (no atomics, no zero-filled outputs)
            partial, hard, loss3 = hip.dice_ce_fwd(logits, labels, C, smooth_nr, smooth_dr, ld, want_hard=True)
        else:
            partial, hard = hip.dice_ce_partials(logits, labels, C, ld, want_hard=True)
            loss3 = hip.dice_ce_finalize(partial, S, smooth_nr, smooth_dr)
        ctx.save_for_backward(logits, labels, partial)
        ctx.sm = (smooth_nr, smooth_dr)
        ctx.ld = ld
        if holder is not None:
            holder["hard"] = hard          # [N, C, 3] = (|P&T|, |P|, |T|) for the metric
            holder["parts"] = loss3        # (total, dice, ce)
            holder["of"] = (logits.data_ptr(), tuple(logits.shape))   # which logits these by-products belong to
        return loss3[0]

    @staticmethod
    def backward(ctx, g):
        logits, labels, partial = ctx.saved_tensors
        N, C = logits.shape[0], logits.shape[1]
        gs = g.reshape(1).to(torch.float32).contiguous()
        if ctx.ld:
            # logits are a view of channels-last rows: the gradient goes out in the same layout (padding channels zeroed
            # by the kernel) and is registered so that the producer of the logits can take the rows as they are
            rows = torch.empty((N,) + tuple(logits.shape[2:]) + (ctx.ld,), dtype=logits.dtype, device=logits.device)
            hip.dice_ce_bwd(logits, labels, partial, gs, rows, C, ctx.sm[0], ctx.sm[1], ctx.ld, ctx.ld)
            _CL_GRADS[rows.data_ptr()] = rows
            perm = (0, logits.dim() - 1) + tuple(range(1, logits.dim() - 1))
            return rows[..., :C].permute(*perm), None, None, None, None
        dl = torch.empty_like(logits)
        hip.dice_ce_bwd(logits, labels, partial, gs, dl, C, ctx.sm[0], ctx.sm[1], 0, 0)
        return dl, None, None, None, None


class DiceCELoss(torch.nn.Module):
    def __init__(self, to_onehot_y=True, softmax=True, squared_pred=True, smooth_nr=1e-5, smooth_dr=1e-5,
                 include_background=True, lambda_dice=1.0, lambda_ce=1.0):
        super().__init__()
        if not (to_onehot_y and softmax and squared_pred and include_background) or lambda_dice != 1.0 or lambda_ce != 1.0:
            raise ValueError("only the reference's configuration is implemented: to_onehot_y=True, softmax=True, "
                             "squared_pred=True, include_background=True, lambda_dice=lambda_ce=1")
        self.smooth_nr, self.smooth_dr = float(smooth_nr), float(smooth_dr)
        self.last = {}   # by-products of the last call: 'hard' counts and (total, dice, ce)

    def forward(self, logits, labels):
        if logits.shape[1] > 16:
            raise ValueError("at most 16 classes are supported")
        return _DiceCEFn.apply(logits, labels, self.smooth_nr, self.smooth_dr, self.last)


def dice_from_counts(hard: torch.Tensor):
    """hard [N,C,3] -> (scores[N,C] with NaN where |T| == 0, not_nans[N,C]) like MONAI
    ``DiceMetric(include_background=True, reduction='none', get_not_nans=True).aggregate()``."""
    inter, p, t = hard[..., 0], hard[..., 1], hard[..., 2]
    score = torch.where(t > 0, 2.0 * inter / (p + t), torch.full_like(inter, float("nan")))
    return score, (~torch.isnan(score)).float()


def dice_metric(logits: torch.Tensor, labels: torch.Tensor):
    """argmax one-hot hard Dice per (n, c) in one fused pass (replaces decollate + AsDiscrete + DiceMetric,
    ``/root/reference/engine/train.py:89-94``)."""
    if not logits.is_cuda:
        raise RuntimeError("dice_metric runs on the GPU only (no CPU fallback)")
    if logits.dtype not in (torch.float32, torch.bfloat16):
        logits = logits.float()
    labels = labels.contiguous()
    if labels.dtype not in (torch.float32, torch.bfloat16, torch.uint8, torch.int64):
        labels = labels.long()
    ld = _channels_last_rows(logits)
    if ld == 0 or logits.data_ptr() % 16 or (ld * logits.element_size()) % 16:
        ld, logits = 0, logits.contiguous()
    _, hard = hip.dice_ce_partials(logits, labels, logits.shape[1], ld, want_hard=True)
    return dice_from_counts(hard)
