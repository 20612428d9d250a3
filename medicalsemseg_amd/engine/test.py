"""Evaluation / test-time inference: mirrors ``/root/reference/engine/test.py`` (``eval_model`` :15-94,
``test_model`` :96-173).  The label map is formed on the device (``msseg_argmax_u8``: the arg max of the blended logits,
the reference's softmax is monotonic) and resampled to the original grid with ``msseg_resample_nearest_u8`` (the
reference's ``resample_3d`` = scipy order-0 zoom, ``utils/misc.py:420-425``); only uint8 maps cross PCIe.  Outputs go
to the reference's directory layout as NIfTI-1 files (``utils/nifti.py``: nibabel is not available) when the input names
carry a NIfTI extension, as ``.npy`` otherwise (synthetic loaders).  The Hausdorff-95 meter (``mHdorffDist``, :20,31,48-51,64)
comes from ``metrics.hausdorff95``: surfaces, exact distances and their histogram on the device."""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import hip, metrics
from ..utils import misc
from .train import _metric_update
from .utils import sliding_window_inference


def label_map(outputs: torch.Tensor) -> torch.Tensor:
    """logits [1, C, D, H, W] -> uint8 [D, H, W] on the device (engine/test.py:140-141)"""
    return hip.argmax_u8(outputs[0].float().contiguous())


def eval_model(inferer, model, data_loader, criterion, device, cfg, log_writer=None):
    """`inferer(inputs=..., network=...)` -> logits (MONAI SlidingWindowInferer call convention, engine/test.py:47), or
    None for the built-in sliding window with the validation settings.  Returns {'eval/<meter>': global average}."""
    model.eval()
    metric_logger = misc.MetricLogger(delimiter="  ")
    for name in ["loss", "mHdorffDist", "mDice"] + ["class" + str(c) + "Dice" for c in range(cfg.output_dim)]:
        metric_logger.add_meter(name, misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = "Evaluation starting"
    for data_iter_step, batch in enumerate(metric_logger.log_every(data_loader, 1, header)):
        inputs = batch["image"].to(device, non_blocking=True)
        labels = batch["label"].to(device, non_blocking=True)
        aff_xyz = misc.get_affine_xyz(batch["image_meta_dict"]["original_affine"]).float().to(device)
        img_name = os.path.split(str(batch["image_meta_dict"]["filename_or_obj"][0]))[-1]
        with torch.no_grad():
            if inferer is None:
                outputs = sliding_window_inference(inputs, aff_xyz, cfg.vol_size, cfg.batch_size_val, model,
                                                   overlap=cfg.val_infer_overlap, mode="gaussian")
            else:
                # the reference hands the bare module to MONAI's inferer, which feeds it a bare window tensor (the models
                # want the (window, centers, affine) tuple: SURVEY.md M4); an inferer that already builds the tuple passes
                outputs = inferer(inputs=inputs, network=lambda w, *a, **k: model(
                    w if isinstance(w, (tuple, list)) else (w, None, aff_xyz)))
            loss = criterion(outputs, labels)
        mDice = _metric_update(metric_logger, criterion, outputs, labels, cfg.output_dim)
        # engine/test.py:31,48-51: Hausdorff-95 of the arg-max map against the labels, all classes, MONAI's "mean" reduction
        pred_maps = torch.stack([label_map(outputs[b:b + 1]) for b in range(outputs.shape[0])])
        hdorf, _ = metrics.hausdorff_mean(metrics.hausdorff95(pred_maps, labels, cfg.output_dim))
        metric_logger.update(loss=loss.item(), mHdorffDist=hdorf, mDice=mDice.item())
        if getattr(cfg, "save_eval_output", False) and cfg.output_dir:
            os.makedirs(cfg.output_dir, exist_ok=True)
            np.save(os.path.join(cfg.output_dir, "pred_" + img_name + ".npy"), label_map(outputs).cpu().numpy())
    metric_logger.synchronize_between_processes()
    print("Evaluation averaged stats:", metric_logger.log_all_average())
    return {"eval/" + k: meter.global_avg for k, meter in metric_logger.meters.items()}


def test_model(model, data_loader, device, cfg, log_writer=None):
    model.eval()
    air_cval = (0.0 - cfg.t_norm_mean) / cfg.t_norm_std if cfg.t_normalize else 0.0
    for i, batch in enumerate(data_loader):
        inputs = batch["image"].to(device, non_blocking=True)
        aff_xyz = misc.get_affine_xyz(batch["image_meta_dict"]["original_affine"]).float().to(device)
        img_name = os.path.split(str(batch["image_meta_dict"]["filename_or_obj"][0]))[-1].split("img")[-1]
        with torch.no_grad():
            outputs = sliding_window_inference(inputs, aff_xyz, cfg.vol_size, cfg.batch_size_val, model,
                                               overlap=cfg.val_infer_overlap, mode="gaussian", cval=air_cval)
        seg = label_map(outputs)
        seg_rs = None
        if getattr(cfg, "t_voxel_spacings", None):
            target = None
            for t in batch.get("image_transforms", []):
                if t["class"][0] == "Spacingd":
                    target = [int(v[0]) if hasattr(v, "__len__") else int(v) for v in t["orig_size"]]
            if target is not None:
                seg_rs = hip.resample_nearest_u8(seg, target)
        if getattr(cfg, "save_eval_output", False) and cfg.output_dir:
            out_dir = os.path.join(cfg.output_dir, "test_output", "Fold" + str(getattr(cfg, "cv_fold", 0)))
            meta = batch["image_meta_dict"]
            aff = _affine0(meta.get("affine"))                       # translation zeroed, as engine/test.py:151-152 does
            _save_volume(os.path.join(out_dir, "pred"), img_name, seg.cpu().numpy(), aff)
            if img_name.endswith((".nii", ".nii.gz")):
                _save_volume(os.path.join(out_dir, "img"), img_name, inputs.squeeze().float().cpu().numpy(), aff)
            if seg_rs is not None:
                _save_volume(os.path.join(out_dir, "rs"), img_name, seg_rs.cpu().numpy(), _affine0(meta.get("original_affine")))
    return None


def _affine0(a):
    """first affine of the batch with its translation set to zero (/root/reference/engine/test.py:151-152); identity when
    the loader carries none"""
    if a is None:
        return np.eye(4)
    a = np.array(torch.as_tensor(a).detach().cpu().numpy(), dtype=np.float64).reshape(-1, 4, 4)[0]
    a[0:3, 3] = 0
    return a


def _save_volume(folder, name, arr, affine):
    """NIfTI-1 (what nib.save(nib.Nifti1Image(arr, affine), path) writes, /root/reference/engine/test.py:160-170) for
    names that carry a NIfTI extension, .npy otherwise (synthetic loaders)"""
    os.makedirs(folder, exist_ok=True)
    if name.endswith((".nii", ".nii.gz")):
        from ..utils.nifti import save_nifti
        save_nifti(os.path.join(folder, name), arr, affine)
    else:
        np.save(os.path.join(folder, name + ".npy"), arr)


def majority_vote(fold_maps, n_classes: int) -> torch.Tensor:
    """fold_maps: sequence of uint8 [D, H, W] label maps (one per fold) -> voted uint8 map on the device
    (/root/reference/majority_vote.py:23-37)"""
    stack = torch.stack([torch.as_tensor(m, dtype=torch.uint8) for m in fold_maps]).contiguous()
    if not stack.is_cuda:
        stack = stack.cuda()
    return hip.majority_vote_u8(stack, n_classes)
