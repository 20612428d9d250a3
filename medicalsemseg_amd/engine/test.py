"""Evaluation / test-time inference: mirrors ``/root/reference/engine/test.py`` (``eval_model`` :15-94,
``test_model`` :96-173).  Hausdorff95 and the NIfTI dump are outside the hot path: results
are returned / saved as ``.npy`` label maps instead of NIfTI (nibabel is not available here)."""
from __future__ import annotations

import os

import numpy as np
import torch

from ..utils import misc
from .train import _metric_update
from .utils import sliding_window_inference


def eval_model(inferer, model, data_loader, criterion, device, cfg, log_writer=None):
    """`inferer(inputs, network)` -> logits; pass ``functools.partial``-style callables or None for the built-in
    sliding window with the validation settings."""
    model.eval()
    metric_logger = misc.MetricLogger(delimiter="  ")
    header = "Evaluation:"
    for data_iter_step, batch in enumerate(metric_logger.log_every(data_loader, 1, header)):
        inputs = batch["image"].to(device, non_blocking=True)
        labels = batch["label"].to(device, non_blocking=True)
        aff_xyz = misc.get_affine_xyz(batch["image_meta_dict"]["original_affine"]).float().to(device)
        with torch.no_grad():
            if inferer is None:
                outputs = sliding_window_inference(inputs, aff_xyz, cfg.vol_size, cfg.batch_size_val, model,
                                                   overlap=cfg.val_infer_overlap, mode="gaussian")
            else:
                outputs = inferer(inputs=inputs, network=lambda w: model((w, None, aff_xyz)))
            loss = criterion(outputs, labels)
        mDice = _metric_update(metric_logger, criterion, outputs, labels, cfg.output_dim)
        metric_logger.update(loss=loss.item(), mDice=mDice.item())
        if getattr(cfg, "save_eval_output", False) and cfg.output_dir:
            os.makedirs(cfg.output_dir, exist_ok=True)
            np.save(os.path.join(cfg.output_dir, f"eval_{data_iter_step}.npy"),
                    outputs.argmax(1).to(torch.uint8).cpu().numpy())
    metric_logger.synchronize_between_processes()
    print("Evaluation averaged stats:", metric_logger.log_all_average())
    return {"eval/" + k: meter.global_avg for k, meter in metric_logger.meters.items()}


def test_model(model, data_loader, device, cfg, log_writer=None):
    model.eval()
    air_cval = (0.0 - cfg.t_norm_mean) / cfg.t_norm_std if cfg.t_normalize else 0.0
    for i, batch in enumerate(data_loader):
        inputs = batch["image"].to(device, non_blocking=True)
        aff_xyz = misc.get_affine_xyz(batch["image_meta_dict"]["original_affine"]).float().to(device)
        with torch.no_grad():
            outputs = sliding_window_inference(inputs, aff_xyz, cfg.vol_size, cfg.batch_size_val, model,
                                               overlap=cfg.val_infer_overlap, mode="gaussian", cval=air_cval)
        seg = outputs.softmax(1).argmax(1).to(torch.uint8).cpu().numpy()
        if cfg.output_dir:
            os.makedirs(cfg.output_dir, exist_ok=True)
            np.save(os.path.join(cfg.output_dir, f"test_{i}.npy"), seg)
    return None
