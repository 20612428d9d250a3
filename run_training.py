#!/usr/bin/env python
"""Training driver with the reference's flow and flags (``/root/reference/run_training.py:27-190``): distributed
init -> seeds -> data -> build_model -> AdamW(+timm-style weight-decay groups) -> cosine schedule with linear warm-up
-> resume -> DiceCE -> epochs of train_one_epoch / run_validation with best-mDice and periodic checkpoints.

Differences: runs on MI355X through ``medicalsemseg_amd`` (no CPU fallback); ``--synthetic`` replaces the MONAI
data pipeline (out of the hot-path scope); logging is stdout + ``log.txt`` JSON lines (tensorboardX / Neptune are not
available here); bf16 compute instead of fp16 autocast, so the GradScaler is a disabled pass-through;
gradients are exchanged with one flat RCCL all-reduce instead of DDP buckets.
"""
from __future__ import annotations

import datetime
import json
import os
import time

import numpy as np
import torch

from medicalsemseg_amd import parallel
from medicalsemseg_amd.data import SyntheticLoader
from medicalsemseg_amd.engine.train import train_one_epoch
from medicalsemseg_amd.engine.val import run_validation
from medicalsemseg_amd.losses import DiceCELoss
from medicalsemseg_amd.models.model_builder import build_model
from medicalsemseg_amd.optim import FlatAdamW, LinearWarmupCosineAnnealingLR, add_weight_decay
from medicalsemseg_amd.utils import misc
from medicalsemseg_amd.utils.arguments import get_args


def main(cfg):
    misc.init_distributed_mode(cfg)
    if not torch.cuda.is_available():
        raise SystemExit("run_training.py needs an MI355X: medicalsemseg_amd has no CPU fallback "
                         "(the CPU oracle under oracle/ is test infrastructure)")
    # MSSEG_BENCH_ONE_DEVICE: several gloo ranks on one GPU, to rehearse the data-parallel control flow on a 1-GPU box
    device = torch.device("cuda", 0 if os.environ.get("MSSEG_BENCH_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    seed = cfg.seed + misc.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)

    if not cfg.synthetic:
        raise SystemExit("only --synthetic data is available in this build: the MONAI/Decathlon pipeline of the "
                         "reference (data/*) is outside the hot-path scope (SURVEY.md section 2)")
    vol = cfg.vol_size if isinstance(cfg.vol_size, int) else cfg.vol_size[0]
    vval = cfg.synthetic_val_size if isinstance(cfg.synthetic_val_size, int) else cfg.synthetic_val_size[0]
    if cfg.t_rand_crop_fgbg:
        # device-side data path: one synthetic "CT" of 2x the patch size cached in HBM, patches cropped
        # (fg / bg centres with the reference's pos : neg odds) and augmented by one gather kernel per batch
        from medicalsemseg_amd.data import sphere_labels
        from medicalsemseg_amd.data_device import DevicePatchLoader
        g = torch.Generator().manual_seed(seed)
        lab = sphere_labels(2 * vol, cfg.output_dim)
        img = torch.randn(cfg.in_chans, 2 * vol, 2 * vol, 2 * vol, generator=g) + 0.5 * lab[None]
        loader_train = DevicePatchLoader(img, lab, vol, cfg.n_images_per_batch, cfg.synthetic_steps, device, seed=seed,
                                         pos=cfg.t_rand_crop_pos_weight or 1.0, neg=cfg.t_rand_crop_neg_weight or 1.0,
                                         flip_prob=cfg.t_flip_prob, rot_prob=cfg.t_rot_prob,
                                         shift_os=cfg.t_intensity_shift_os, shift_prob=cfg.t_intensity_shift_prob,
                                         scale_f=cfg.t_intensity_scale_factors, scale_prob=cfg.t_intensity_scale_prob,
                                         image_threshold=-1e9)
    else:
        loader_train = SyntheticLoader(cfg.synthetic_steps, cfg.n_images_per_batch, vol, cfg.in_chans, cfg.output_dim, seed)
    loader_val = SyntheticLoader(1, 1, vval, cfg.in_chans, cfg.output_dim, seed + 7, with_crop_info=False)

    model = build_model(cfg).to(device)
    if cfg.distributed:
        # /root/reference/run_training.py:83: every BatchNorm uses the statistics of the global batch under data
        # parallelism (SwinDepth / SwInception / SegFormer3D carry BatchNorms; the InstanceNorm models have none)
        n_bn = parallel.convert_sync_batchnorm(model)
        if n_bn and misc.is_main_process():
            print(f"SyncBatchNorm: {n_bn} BatchNorm holder(s) switched to cross-rank statistics")
    print("parameters:", misc.count_parameters(model))
    groups = add_weight_decay(model, cfg.weight_decay)
    optimizer = FlatAdamW(groups, lr=cfg.lr, betas=(0.9, 0.95), eps=1e-6)
    if cfg.distributed:
        # gradient averaging over the ranks, overlapped with the tail of the backward where the model splits it
        optimizer.grad_sync = parallel.GradSync(optimizer, model)
        # identical initial weights on every rank
        torch.distributed.broadcast(optimizer.flat_param, src=0)
    loss_scaler = torch.amp.GradScaler("cuda", enabled=False)
    scheduler = LinearWarmupCosineAnnealingLR(optimizer, warmup_epochs=cfg.warmup_epochs, max_epochs=cfg.epochs)
    misc.load_model(cfg, model, optimizer, loss_scaler, scheduler)
    if cfg.loss_fn != "DiceCE":
        raise RuntimeError("Could not parse loss function argument (only DiceCE is on the hot path).")
    criterion = DiceCELoss(to_onehot_y=True, softmax=True, squared_pred=True, smooth_nr=cfg.smooth_nr, smooth_dr=cfg.smooth_dr)

    best, best_epoch = 0.0, 0
    start = time.time()
    for epoch in range(cfg.start_epoch, cfg.epochs):
        if cfg.distributed:
            torch.distributed.barrier()
        stats = train_one_epoch(model, loader_train, optimizer, criterion, device, epoch, loss_scaler, cfg)
        if (epoch + 1) % cfg.val_interval == 0 or epoch + 1 == cfg.epochs:
            if cfg.distributed:
                torch.distributed.barrier()
            vstats = run_validation(model, loader_val, criterion, device, epoch, cfg)
            stats.update(vstats)
            if vstats["val/mDice"] > best and cfg.output_dir:
                best, best_epoch = vstats["val/mDice"], epoch
                misc.save_model(cfg, epoch, model, optimizer, loss_scaler, scheduler, filename="best_model.pth")
        if cfg.output_dir and ((epoch + 1) % cfg.save_ckpt_freq == 0 or epoch + 1 == cfg.epochs):
            misc.save_model(cfg, epoch, model, optimizer, loss_scaler, scheduler)
        if cfg.output_dir and misc.is_main_process():
            os.makedirs(cfg.output_dir, exist_ok=True)
            with open(os.path.join(cfg.output_dir, "log.txt"), "a") as f:
                f.write(json.dumps({"epoch": epoch, **{k: float(v) for k, v in stats.items()}}) + "\n")
        scheduler.step()
    print("Training time", str(datetime.timedelta(seconds=int(time.time() - start))), "best val mDice", best, "at", best_epoch)
    if parallel.is_dist():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(get_args())
