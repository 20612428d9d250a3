"""One training epoch.  Signature, meter names and control flow of ``/root/reference/engine/train.py:14-128``;
the per-step hard-Dice metric re-uses the reductions the fused DiceCE pass already produced (no second pass
over the logits), and nothing here assumes CUDA-only APIs beyond the device the tensors live on."""
from __future__ import annotations

import math
import sys

import torch

from .. import losses as L
from ..utils import misc


def _metric_update(metric_logger, criterion, outputs, labels, n_cls):
    # by-products of the fused DiceCE pass are used only when they were produced from THESE logits (a validation call or
    # another cached graph may have run the criterion in between)
    last = getattr(criterion, "last", None) or {}
    hard = last.get("hard") if last.get("of") == (outputs.data_ptr(), tuple(outputs.shape)) else None
    custom = getattr(criterion, "hard_dice", None)   # injection point (tests drive the loop with the CPU oracle)
    if custom is not None:
        scores, not_nans = custom(outputs.detach(), labels)
    elif hard is not None:
        scores, not_nans = L.dice_from_counts(hard)
    else:
        scores, not_nans = L.dice_metric(outputs.detach(), labels)
    scores, not_nans = scores.cpu(), not_nans.cpu()
    class_means = torch.zeros(n_cls)
    for c in range(n_cls):
        cd = scores[:, c].nanmean() if not_nans[:, c].sum() > 0 else torch.tensor(float("nan"))
        class_means[c] = cd
        metric_logger.update(**{"class" + str(c) + "Dice": float(cd)})
    return class_means.nanmean()


class _GraphedFwdBwd:
    """forward + loss + backward of one static-shape batch, replayed from a captured hipGraph.

    A UNet step is ~200 kernel launches of 5-150 us; issued from Python they cost ~8 ms of host time per step against
    ~5 ms of GPU time.  Models that declare `graph_safe` (static shapes, no host synchronisation inside the step) and
    ignore (crop_loc, affine) take this path; the optimiser step, gradient clipping, the data-parallel all-reduce and
    the metrics stay outside the graph, exactly where the reference loop has them.  With a two-phase backward
    (parallel.GradSync) the step is two graphs and the first all-reduce starts between them."""

    _cache = {}

    @classmethod
    def get(cls, model, criterion, optimizer, inputs, labels):
        key = (id(model), id(criterion), tuple(inputs.shape), inputs.dtype, tuple(labels.shape), labels.dtype, inputs.device,
               optimizer.flat_grad.data_ptr(), optimizer.flat_param.data_ptr())   # a new FlatAdamW re-points p.data / p.grad
        g = cls._cache.get(key)
        if g is None or g.model() is not model:
            g = cls._cache[key] = cls(model, criterion, optimizer, inputs, labels)
        return g

    def __init__(self, model, criterion, optimizer, inputs, labels):
        import weakref
        from .. import layers
        self.model = weakref.ref(model)
        self.x, self.y = inputs.clone(), labels.clone()
        side = torch.cuda.Stream(device=inputs.device)
        side.wait_stream(torch.cuda.current_stream())
        net = getattr(model, "module", model)
        tail = getattr(net, "backward_tail", None) if getattr(net, "_defer_tail", False) else None
        with torch.cuda.stream(side):   # builds the packed-weight images, workspaces and gradient views
            criterion(model((self.x, None, None)), self.y).backward()
            if tail is not None:
                tail()
        torch.cuda.current_stream().wait_stream(side)
        optimizer.zero_grad()           # the warm-up pass must not count
        layers.PACK_REGISTRY.prepare()
        layers.bump_weights_epoch()     # the capture then starts with the batched weight re-packing
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = model((self.x, None, None))
            self.loss = criterion(self.out, self.y)
            self.loss.backward()
        # the criterion's by-products (hard-Dice counts of THIS graph's static logits): a replay does not run
        # _DiceCEFn.forward in Python, so they are re-bound after every replay
        self.last = dict(getattr(criterion, "last", {}))
        self.criterion = weakref.ref(criterion)
        # two-phase backward (parallel.GradSync): the tail of the backward is its own graph, so the all-reduce of the
        # gradients the head finished can be started between the two replays
        self.graph_tail = None
        if tail is not None:
            self.graph_tail = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_tail, pool=self.graph.pool()):
                tail()
        # FlatAdamW's lazy zero_grad: the captured kernels overwrite the gradients they produce; a replay runs no Python, so
        # the parameters written during the capture are marked as written after every replay (optim.FlatAdamW.mark_written)
        self.optimizer = weakref.ref(optimizer)
        self.written = optimizer.written_params() if hasattr(optimizer, "written_params") else None

    def __call__(self, inputs, labels, between=None):
        self.x.copy_(inputs)
        self.y.copy_(labels)
        self.graph.replay()
        if self.graph_tail is not None:
            if between is not None:
                between()
            self.graph_tail.replay()
        crit = self.criterion()
        if crit is not None and hasattr(crit, "last"):
            crit.last.update(self.last)
        opt = self.optimizer()
        if opt is not None and self.written is not None:
            opt.mark_written(self.written)
        return self.out, self.loss


def _graph_ok(model, criterion, optimizer, loss_scaler, inputs, cfg):
    import os
    net = getattr(model, "module", model)
    return (inputs.is_cuda and getattr(net, "graph_safe", False) and isinstance(criterion, L.DiceCELoss)
            and hasattr(optimizer, "flat_grad") and not getattr(loss_scaler, "is_enabled", lambda: False)()
            and not bool(getattr(cfg, "anomaly_detection", False)) and not os.environ.get("MSSEG_NO_TRAIN_GRAPH"))


def train_one_epoch(model, data_loader, optimizer, criterion, device, epoch, loss_scaler, cfg, log_writer=None):
    model.train()
    metric_logger = misc.MetricLogger(delimiter="  ")
    for name in ["lr", "loss", "mDice"] + ["class" + str(c) + "Dice" for c in range(cfg.output_dim)]:
        metric_logger.add_meter(name, misc.SmoothedValue(window_size=100, fmt="{value:.6f}"))
    header = "Epoch: [{}]".format(epoch)
    iters = len(data_loader)
    optimizer.zero_grad()
    amp_dtype = torch.bfloat16  # bf16 on MI355X: no loss scaling needed (the reference used fp16 + GradScaler)

    for data_iter_step, batch in enumerate(metric_logger.log_every(data_loader, 20, header)):
        torch.autograd.set_detect_anomaly(bool(getattr(cfg, "anomaly_detection", False)))
        inputs = batch["image"].to(device, non_blocking=True)
        labels = batch["label"].to(device, non_blocking=True)
        aff_xyz = misc.get_affine_xyz(batch["image_meta_dict"]["original_affine"]).float().to(device, non_blocking=True)
        crop_loc = None
        for t in batch.get("image_transforms", []):
            if t["class"][0] in ("RandCropByPosNegLabeld", "RandCropByClassesd"):
                crop_loc = misc.get_rel_crop_loc(t)

        graphed = _graph_ok(model, criterion, optimizer, loss_scaler, inputs, cfg)
        gsync = getattr(optimizer, "grad_sync", None)          # parallel.GradSync (data parallel), else None
        net = getattr(model, "module", model)
        if graphed:
            outputs, loss = _GraphedFwdBwd.get(model, criterion, optimizer, inputs, labels)(
                inputs, labels, gsync.start if gsync is not None else None)
        else:
            outputs = model((inputs, crop_loc, aff_xyz))   # compute dtype is a property of the model (bf16 / fp32)
            loss = criterion(outputs, labels)
        loss_value = loss.item()
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            sys.exit(1)

        if not graphed:
            loss_scaler.scale(loss).backward()
            if getattr(net, "_defer_tail", False):
                if gsync is not None:
                    gsync.start()
                net.backward_tail()
        if gsync is not None:
            gsync.finish()   # gradients are averaged over the ranks before clipping, as under DDP
        if cfg.gradient_clipping is not None:
            loss_scaler.unscale_(optimizer)
            if hasattr(optimizer, "clip_grad_norm_"):
                optimizer.clip_grad_norm_(cfg.gradient_clipping)
            else:
                torch.nn.utils.clip_grad_norm_(model.parameters(), cfg.gradient_clipping)
        loss_scaler.step(optimizer)
        loss_scaler.update()
        optimizer.zero_grad()

        mDice = _metric_update(metric_logger, criterion, outputs, labels, cfg.output_dim)
        metric_logger.update(loss=loss_value)
        metric_logger.update(mDice=mDice.item())
        lr = optimizer.param_groups[0]["lr"]
        metric_logger.update(lr=lr)
        loss_value_reduce = misc.all_reduce_mean(loss_value)
        if log_writer is not None:
            epoch_1000x = int((data_iter_step / iters + epoch) * 1000)
            log_writer.add_scalar("train_loss", loss_value_reduce, epoch_1000x)
            log_writer.add_scalar("lr", lr, epoch_1000x)

    metric_logger.synchronize_between_processes()
    print("Training averaged stats:", metric_logger.log_all_average())
    return {"train/" + k: meter.global_avg for k, meter in metric_logger.meters.items()}
