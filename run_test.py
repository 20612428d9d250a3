#!/usr/bin/env python
"""Test-time inference driver: mirror of ``/root/reference/run_test.py`` (build the model, load ``cfg.resume``, run
``engine.test.test_model``: sliding window -> arg-max label map on the device -> nearest resample to the original grid
-> saved maps).  Data: ``--synthetic`` volumes.

    python run_test.py --synthetic --model UNet --output_dim 3 --vol_size 96 --resume out/best_model.pth \
        --save_eval_output --output_dir out
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from medicalsemseg_amd.data import SyntheticLoader
from medicalsemseg_amd.engine.test import test_model
from medicalsemseg_amd.models.model_builder import build_model
from medicalsemseg_amd.utils import misc
from medicalsemseg_amd.utils.arguments import get_args


def main(cfg):
    misc.init_distributed_mode(cfg)
    if not torch.cuda.is_available():
        raise SystemExit("run_test.py needs an MI355X: medicalsemseg_amd has no CPU fallback")
    device = torch.device("cuda", 0 if os.environ.get("MSSEG_BENCH_ONE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    if not cfg.synthetic:
        raise SystemExit("only --synthetic data is available in this build (SURVEY.md section 2)")
    torch.manual_seed(cfg.seed)
    model = build_model(cfg).to(device)
    cfg.eval = True
    misc.load_model(cfg, model)
    vval = cfg.synthetic_val_size if isinstance(cfg.synthetic_val_size, int) else cfg.synthetic_val_size[0]
    loader = SyntheticLoader(cfg.synthetic_steps, 1, vval, cfg.in_chans, cfg.output_dim, cfg.seed + 11 + misc.get_rank(),
                             with_crop_info=False)
    test_model(model, loader, device, cfg)
    if cfg.distributed:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(get_args())
