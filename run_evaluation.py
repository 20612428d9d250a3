#!/usr/bin/env python
"""Evaluation driver: mirror of ``/root/reference/run_evaluation.py`` (build the model, load ``cfg.resume``, run
``engine.test.eval_model`` over the validation volumes) on the MI355X hot path.  The reference hands the model to MONAI's
``SlidingWindowInferer`` (``run_evaluation.py:68-74``); here ``inferer=None`` selects the built-in sliding window with the
same settings (roi ``cfg.vol_size``, ``cfg.batch_size_val`` windows per forward, overlap ``cfg.val_infer_overlap``,
gaussian blending).  Data: ``--synthetic`` volumes (the MONAI / Decathlon pipeline is outside the hot-path scope).

    python run_evaluation.py --synthetic --model UNet --output_dim 3 --vol_size 96 --resume out/best_model.pth
"""
from __future__ import annotations

import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from medicalsemseg_amd.data import SyntheticLoader
from medicalsemseg_amd.engine.test import eval_model
from medicalsemseg_amd.losses import DiceCELoss
from medicalsemseg_amd.models.model_builder import build_model
from medicalsemseg_amd.utils import misc
from medicalsemseg_amd.utils.arguments import get_args


def main(cfg):
    misc.init_distributed_mode(cfg)
    if not torch.cuda.is_available():
        raise SystemExit("run_evaluation.py needs an MI355X: medicalsemseg_amd has no CPU fallback")
    device = torch.device("cuda", 0 if os.environ.get("MSSEG_BENCH_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    if not cfg.synthetic:
        raise SystemExit("only --synthetic data is available in this build (SURVEY.md section 2)")
    torch.manual_seed(cfg.seed)
    model = build_model(cfg).to(device)
    cfg.eval = True
    misc.load_model(cfg, model)                       # weights only: load_state_dict(torch.load(cfg.resume)['model'])
    criterion = DiceCELoss(to_onehot_y=True, softmax=True, squared_pred=True, smooth_nr=cfg.smooth_nr, smooth_dr=cfg.smooth_dr)
    vval = cfg.synthetic_val_size if isinstance(cfg.synthetic_val_size, int) else cfg.synthetic_val_size[0]
    loader = SyntheticLoader(cfg.synthetic_steps, 1, vval, cfg.in_chans, cfg.output_dim, cfg.seed + 7 + misc.get_rank(),
                             with_crop_info=False)
    stats = eval_model(None, model, loader, criterion, device, cfg)
    if misc.is_main_process():
        print(json.dumps(stats))
        if cfg.output_dir:
            os.makedirs(cfg.output_dir, exist_ok=True)
            with open(os.path.join(cfg.output_dir, "eval.json"), "w") as fh:
                json.dump(stats, fh)
    if cfg.distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(get_args())
