"""The small helpers the engine calls: meters, distributed reductions of scalars, affine / crop helpers,
checkpoint I/O.  Behavioural mirror of ``/root/reference/utils/misc.py`` (SmoothedValue :16-86, MetricLogger
:89-195, save/load :268-305, all_reduce_mean :307-315, get_affine_xyz :427-432, get_rel_crop_loc :434-441),
re-written; device-agnostic (meters reduce on whatever device the process group uses)."""
from __future__ import annotations

import datetime
import math
import os
import time
from collections import defaultdict, deque
from pathlib import Path

import torch
import torch.distributed as dist

from .. import parallel


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return parallel.world_size()


def get_rank():
    return parallel.rank()


def is_main_process():
    return get_rank() == 0


def _reduce_device():
    if is_dist_avail_and_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


class SmoothedValue:
    """Windowed median/mean + global average of a scalar series."""

    def __init__(self, window_size=100, fmt=None):
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """sums (count, total) over ranks; the window is left per-rank (as the reference)."""
        if not is_dist_avail_and_initialized():
            return
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=_reduce_device())
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), float(t[1].item())

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count if self.count > 0 else 0.0

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        if not self.deque:
            return self.fmt.format(median=0.0, avg=0.0, global_avg=0.0, max=0.0, value=0.0)
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max,
                               value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            if isinstance(v, float) and math.isnan(v):
                continue  # a class absent from the batch: the reference skips `np.nan`
            assert isinstance(v, (float, int))
            self.meters[k].update(v)

    def __getattr__(self, attr):
        meters = self.__dict__.get("meters", {})
        if attr in meters:
            return meters[attr]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{attr}'")

    def __str__(self):
        return self.delimiter.join(f"{n}: {m}" for n, m in self.meters.items())

    def log_all_average(self):
        return self.delimiter.join(f"{n}: {m.global_avg:.4f}" for n, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None):
        header = header or ""
        start = end = time.time()
        iter_time, data_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        n = len(iterable)
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0 or i == n - 1:
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i))))
                msg = [header, f"[{i}/{n}]", f"eta: {eta}", str(self), f"time: {iter_time}", f"data: {data_time}"]
                if torch.cuda.is_available():
                    msg.append(f"max mem: {torch.cuda.max_memory_allocated() / 2 ** 20:.0f}")
                print(self.delimiter.join(msg))
            end = time.time()
        total = time.time() - start
        print(f"{header} Total time: {datetime.timedelta(seconds=int(total))} ({total / max(n, 1):.4f} s / it)")


def all_reduce_mean(x: float) -> float:
    return parallel.all_reduce_mean(x)


def get_affine_xyz(full_affine: torch.Tensor) -> torch.Tensor:
    """diag of the 4x4 affine -> [B,3]"""
    return torch.stack([full_affine[:, 0, 0], full_affine[:, 1, 1], full_affine[:, 2, 2]], dim=1)


def get_rel_crop_loc(rand_crop_transform_dict) -> torch.Tensor:
    os_, c = rand_crop_transform_dict["orig_size"], rand_crop_transform_dict["extra_info"]["center"]
    return torch.stack([c[0] / os_[0], c[1] / os_[1], c[2] / os_[2]], dim=1)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def save_model(cfg, epoch, model_without_ddp, optimizer, loss_scaler, scheduler=None, filename=None):
    out = Path(cfg.output_dir)
    out.mkdir(parents=True, exist_ok=True)
    path = out / (filename or f"checkpoint-{epoch}.pth")
    to_save = {"model": model_without_ddp.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch,
               "scaler": loss_scaler.state_dict() if loss_scaler is not None else None,
               "scheduler": scheduler.state_dict() if scheduler is not None else None, "cfg": vars(cfg)}
    save_on_master(to_save, str(path))
    return str(path)


def load_model(cfg, model_without_ddp, optimizer=None, loss_scaler=None, scheduler=None):
    if not getattr(cfg, "resume", ""):
        return
    if str(cfg.resume).startswith("http"):
        raise RuntimeError("remote checkpoints are not supported (no network)")
    # weights-only load; the reference stores `cfg` as an argparse.Namespace (/root/reference/utils/misc.py:268-283),
    # an inert container that is allow-listed instead of unpickling arbitrary globals
    import argparse
    with torch.serialization.safe_globals([argparse.Namespace]):
        ck = torch.load(cfg.resume, map_location="cpu", weights_only=True)
    model_without_ddp.load_state_dict(ck["model"])
    print(f"Resume checkpoint {cfg.resume}")
    if optimizer is not None and "optimizer" in ck and "epoch" in ck and not getattr(cfg, "eval", False):
        # FlatAdamW takes its own layout or a torch.optim.AdamW state dict (converted in param-group order); anything
        # else fails loudly rather than silently restarting the moments mid-schedule
        optimizer.load_state_dict(ck["optimizer"])
        cfg.start_epoch = ck["epoch"] + 1
        if loss_scaler is not None and ck.get("scaler") is not None:
            loss_scaler.load_state_dict(ck["scaler"])
        if scheduler is not None and ck.get("scheduler") is not None:
            scheduler.load_state_dict(ck["scheduler"])


def init_distributed_mode(cfg):
    """torchrun env:// rendezvous (/root/reference/utils/misc.py:227-266); single process otherwise."""
    parallel.init_from_env(getattr(cfg, "backend", None) if torch.cuda.is_available() else "gloo")
    cfg.distributed = parallel.world_size() > 1
    cfg.rank, cfg.world_size = parallel.rank(), parallel.world_size()
    cfg.gpu = int(os.environ.get("LOCAL_RANK", "0"))
    if cfg.rank != 0:  # rank-0-only printing, as the reference's print monkey-patch
        import builtins
        _p = builtins.print

        def _quiet(*a, force=False, **k):
            if force:
                _p(*a, **k)
        builtins.print = _quiet
