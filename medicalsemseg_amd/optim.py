"""Flat-buffer AdamW + gradient clipping + LR schedule for the hot path.

* ``FlatAdamW``: every parameter becomes a view into ONE fp32 buffer, every gradient a view into another;
  the optimiser step is a single fused HIP kernel (``msseg_adamw_step``) and -- under data parallelism --
  the gradient exchange is ONE RCCL all-reduce of the flat gradient buffer (``parallel.py``).
  Semantics follow ``torch.optim.AdamW(param_groups, lr, betas=(0.9, 0.95), eps=1e-6)`` with timm's
  ``add_weight_decay`` grouping as wired at ``/root/reference/run_training.py:92-93`` (no decay for 1-D
  parameters and ``.bias``).
* ``LinearWarmupCosineAnnealingLR``: closed form of ``/root/reference/models/optimizers/lr_scheduler.py:93-168``
  (stepped once per epoch, ``run_training.py:174``); pinned by ``tests/golden/lr_misc.npz``.
"""
from __future__ import annotations

import math
import os
from typing import Iterable, List

import torch

from . import hip, layers


def add_weight_decay(model: torch.nn.Module, weight_decay=1e-5, skip_list=()):
    """timm.optim.optim_factory.add_weight_decay: 1-D params and *.bias get no decay."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.ndim <= 1 or name.endswith(".bias") or name in skip_list:
            no_decay.append(p)
        else:
            decay.append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.95), eps=1e-6, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        ps: List[torch.nn.Parameter] = [p for g in self.param_groups for p in g["params"]]
        if not ps:
            raise ValueError("no parameters")
        dev = ps[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdamW runs on the GPU only (no CPU fallback)")
        wds = {g["weight_decay"] for g in self.param_groups if g["weight_decay"] != 0.0}
        if len(wds) > 1:
            raise ValueError("FlatAdamW supports one non-zero weight_decay value")
        self._wd = wds.pop() if wds else 0.0
        n = sum(p.numel() for p in ps)
        pad = (-n) % 4
        self.flat_param = torch.zeros(n + pad, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n + pad, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.decay_mask = torch.zeros(n + pad, dtype=torch.uint8, device=dev)
        self._views = []
        off = 0
        with torch.no_grad():
            for g in self.param_groups:
                for p in g["params"]:
                    k = p.numel()
                    self.flat_param[off:off + k].copy_(p.detach().reshape(-1))
                    p.data = self.flat_param[off:off + k].view(p.shape)
                    gv = self.flat_grad[off:off + k].view(p.shape)
                    self._views.append((p, gv))
                    if g["weight_decay"] != 0.0:
                        self.decay_mask[off:off + k] = 1
                    off += k
        self._n = n
        self._step = 0
        # lazy zero_grad (layers._grad_buf): gradients live in flat_grad for good, zero_grad() only opens a new epoch
        # (MSSEG_EAGER_ZERO_GRAD=1: zero_grad() fills the buffer and every kernel accumulates, as torch's set_to_none=False does.
        #  Between a lazy zero_grad() and the next backward the gradient views hold the previous step's values, not zeros.)
        self._gepoch = 0
        if not os.environ.get("MSSEG_EAGER_ZERO_GRAD"):
            for p, _ in self._views:
                p._msseg_gowner = self
                p._msseg_gepoch = -1
        self._gscale = torch.ones(1, dtype=torch.float32, device=dev)
        self._sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._sq_part = torch.zeros(4096, dtype=torch.float32, device=dev)   # per-block sums of grad_norm (fixed-order reduction)
        # device-resident (lr, step): the fused kernel reads them, so a captured hipGraph replays correctly
        self._hyper = torch.tensor([lr, 0.0], dtype=torch.float32, device=dev)
        self._lr_on_device = lr
        self.attach_grads()

    def attach_grads(self):
        """(re)bind every .grad to its slice of the flat gradient buffer (kernels then write in place)."""
        for p, gv in self._views:
            if p.grad is None or p.grad.data_ptr() != gv.data_ptr():
                if p.grad is not None:
                    gv.copy_(p.grad)
                    p._msseg_gepoch = getattr(self, "_gepoch", 0)   # a gradient made elsewhere: counts as written this epoch
                p.grad = gv

    def zero_grad(self, set_to_none: bool = False):
        # gradients are overwritten / accumulated in place in the flat buffer and the views stay bound.  No fill: a new epoch --
        # the first kernel that writes a parameter's gradient in it overwrites (layers._grad_buf); whatever no kernel touched is
        # zeroed before the buffer is read (_zero_untouched)
        self._gepoch += 1
        self.attach_grads()
        if os.environ.get("MSSEG_EAGER_ZERO_GRAD"):
            self.flat_grad.zero_()
            return
        # gradients that torch autograd accumulates into (parameters no kernel of this package has ever written: a torch encoder
        # in front of the UNETR decoder, ...) still need their zeros: contiguous runs of such slices, none for the models here
        for a, b in self._autograd_ranges():
            self.flat_grad[a:b].zero_()

    def _autograd_ranges(self):
        n_k = sum(1 for p, _ in self._views if getattr(p, "_msseg_kgrad", False))
        if getattr(self, "_ar_key", None) != n_k:
            runs, off, start = [], 0, None
            for p, _ in self._views:
                k = p.numel()
                if not getattr(p, "_msseg_kgrad", False):
                    start = off if start is None else start
                elif start is not None:
                    runs.append((start, off)); start = None
                off += k
            if start is not None:
                runs.append((start, off))
            self._ar_key, self._ar = n_k, runs
        for p, _ in self._views:
            if not getattr(p, "_msseg_kgrad", False):
                p._msseg_gepoch = self._gepoch      # zeroed just now; autograd adds into it
        return self._ar

    def written_params(self):
        """parameters whose gradient a kernel wrote since the last zero_grad() (for callers that replay captured launches)"""
        return [p for p, _ in self._views if getattr(p, "_msseg_gepoch", -1) == self._gepoch]

    def mark_written(self, params):
        """a replayed hipGraph wrote these parameters' gradients (the capture's launches overwrite them; no Python ran)"""
        for p in params:
            p._msseg_gepoch = self._gepoch

    def _zero_untouched(self):
        """gradients no kernel wrote since the last zero_grad(): zero them now (normally none: no launch)"""
        for p, gv in self._views:
            # only gradients that this package's kernels write (they overwrite on first touch, so nothing has cleared the slice);
            # everything else -- torch autograd's in-place accumulation, a caller filling .grad by hand -- was zero-filled by
            # zero_grad() and must be left alone
            if getattr(p, "_msseg_kgrad", False) and getattr(p, "_msseg_gepoch", -1) != self._gepoch:
                gv.zero_()
                p._msseg_gepoch = self._gepoch

    def grad_norm(self) -> torch.Tensor:
        self._zero_untouched()
        hip.sumsq(self.flat_grad, self._sq, self._sq_part)
        return self._sq.sqrt()

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ semantics, folded into the step as a gradient scale (no extra pass)."""
        total = self.grad_norm() * self._gscale   # norm of the gradient the step will apply (after 1/world averaging)
        self._gscale.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total

    def early_suffix_offset(self, late_params) -> int:
        """Smallest offset o such that flat_grad[o:] holds no gradient of `late_params` (the ones a split backward
        produces last): flat_grad[o:] can be all-reduced while those are still being computed."""
        late = {id(p) for p in late_params}
        off, o = 0, 0
        for p, _ in self._views:
            off += p.numel()
            if id(p) in late:
                o = off
        return o

    def state_dict(self):
        return {"flat": True, "step": self._step, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def _load_torch_adamw(self, sd):
        """state dict written by torch.optim.AdamW over the same param groups (the reference's checkpoints,
        /root/reference/utils/misc.py:268-283): per-parameter exp_avg / exp_avg_sq go to their slices of the flat
        moments, in param-group order (timm add_weight_decay: [no_decay, decay], as `add_weight_decay` here)."""
        groups = sd["param_groups"]
        if [len(g["params"]) for g in groups] != [len(g["params"]) for g in self.param_groups]:
            raise ValueError("optimizer state does not match the parameter groups of this model")
        ids = [i for g in groups for i in g["params"]]
        steps = set()
        off = 0
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        for (p, _), i in zip(self._views, ids):
            k = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} != parameter {tuple(p.shape)}")
                self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
            off += k
        if len(steps) > 1:
            raise ValueError("parameters with different step counts cannot share the fused optimiser step")
        return steps.pop() if steps else 0, groups

    def load_state_dict(self, sd):
        if not sd.get("flat"):
            if "state" not in sd or "param_groups" not in sd:
                raise ValueError("neither a FlatAdamW nor a torch.optim.AdamW state dict")
            self._step, groups = self._load_torch_adamw(sd)
            self._hyper[1:2].fill_(float(self._step))
            for g, s_ in zip(self.param_groups, groups):
                g.update({k: v for k, v in s_.items() if k in ("lr", "betas", "eps", "weight_decay", "initial_lr")})
            layers.bump_weights_epoch()
            return
        self._step = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self._hyper[1:2].fill_(float(self._step))
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
        # parameters may have been re-loaded (load_state_dict copies into the flat views in place)
        layers.bump_weights_epoch()

    def sync_lr(self):
        """push the host learning rate to the device copy (call between graph replays when the schedule moves)."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_device:
            self._hyper[0:1].fill_(lr)
            self._lr_on_device = lr

    @torch.no_grad()
    def step(self, closure=None):
        self.attach_grads()
        self._zero_untouched()
        self._step += 1
        self.sync_lr()
        lr = self.param_groups[0]["lr"]
        b1, b2 = self.param_groups[0]["betas"]
        eps = self.param_groups[0]["eps"]
        self._hyper[1:2].add_(1.0)
        hip.adamw_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.decay_mask, lr, b1, b2, eps,
                       self._wd, self._step, self._gscale, self._hyper)
        self._gscale.fill_(1.0)
        layers.bump_weights_epoch()
        return None


class LinearWarmupCosineAnnealingLR(torch.optim.lr_scheduler._LRScheduler):
    def __init__(self, optimizer, warmup_epochs: int, max_epochs: int, warmup_start_lr: float = 0.0,
                 eta_min: float = 0.0, last_epoch: int = -1):
        self.warmup_epochs, self.max_epochs = warmup_epochs, max_epochs
        self.warmup_start_lr, self.eta_min = warmup_start_lr, eta_min
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        e = self.last_epoch
        if e < self.warmup_epochs:
            return [self.warmup_start_lr + e * (b - self.warmup_start_lr) / max(self.warmup_epochs - 1, 1)
                    for b in self.base_lrs]
        return [self.eta_min + 0.5 * (b - self.eta_min) *
                (1 + math.cos(math.pi * (e - self.warmup_epochs) / (self.max_epochs - self.warmup_epochs)))
                for b in self.base_lrs]
