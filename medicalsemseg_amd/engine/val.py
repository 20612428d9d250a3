"""Validation with sliding-window inference: mirror of ``/root/reference/engine/val.py:15-110``."""
from __future__ import annotations

import math
import sys

import torch

from ..utils import misc
from .train import _metric_update
from .utils import sliding_window_inference


def run_validation(model, data_loader, criterion, device, epoch, cfg, log_writer=None, inferer=sliding_window_inference):
    model.eval()
    metric_logger = misc.MetricLogger(delimiter="  ")
    for name in ["loss", "mDice"] + ["class" + str(c) + "Dice" for c in range(cfg.output_dim)]:
        metric_logger.add_meter(name, misc.SmoothedValue(window_size=100, fmt="{value:.6f}"))
    header = "Validation for epoch: [{}]".format(epoch)
    air_cval = (0.0 - cfg.t_norm_mean) / cfg.t_norm_std if cfg.t_normalize else 0.0

    for data_iter_step, batch in enumerate(metric_logger.log_every(data_loader, 1, header)):
        inputs = batch["image"].to(device, non_blocking=True)
        labels = batch["label"].to(device, non_blocking=True)
        aff_xyz = misc.get_affine_xyz(batch["image_meta_dict"]["original_affine"]).float().to(device, non_blocking=True)
        with torch.no_grad():
            outputs = inferer(inputs=inputs, affine=aff_xyz, predictor=model, roi_size=cfg.vol_size,
                              sw_batch_size=cfg.batch_size_val, overlap=cfg.val_infer_overlap, mode="gaussian",
                              device=device, sw_device=device, cval=air_cval)
            loss = criterion(outputs, labels)
        loss_value = loss.item()
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping validation".format(loss_value))
            sys.exit(1)
        mDice = _metric_update(metric_logger, criterion, outputs, labels, cfg.output_dim)
        metric_logger.update(loss=loss_value)
        metric_logger.update(mDice=mDice.item())
        loss_value_reduce = misc.all_reduce_mean(loss_value)
        if log_writer is not None:
            epoch_1000x = int((data_iter_step / len(data_loader) + epoch) * 1000)
            log_writer.add_scalar("val_loss", loss_value_reduce, epoch_1000x)

    if torch.cuda.is_available():
        torch.cuda.synchronize()
    metric_logger.synchronize_between_processes()
    print("Validation averaged stats:", metric_logger.log_all_average())
    return {"val/" + k: meter.global_avg for k, meter in metric_logger.meters.items()}
