"""Single-node data parallelism for the hot path: one process per GPU, RCCL over xGMI.

The reference wraps the model in ``DistributedDataParallel(find_unused_parameters=True)``
(``/root/reference/run_training.py:82-85``): bucketed all-reduce(sum)/world of fp32 gradients every step.
Here all gradients already live in ONE flat fp32 buffer (``optim.FlatAdamW``), so the exchange is a single
all-reduce of that buffer (23 MB for the UNet: one message, all 7 xGMI links busy, < 0.3 ms) issued on a side
stream right after backward; the 1/world scaling is folded into the optimiser's gradient scale.
Works on CPU tensors with the ``gloo`` backend too (tests, world_size 2).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def rank() -> int:
    return dist.get_rank() if is_dist() else 0


def init_from_env(backend: str | None = None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or is_dist():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        # MSSEG_DIST_BACKEND=gloo: rehearse the multi-rank control flow where RCCL cannot run (several ranks on one GPU)
        backend = os.environ.get("MSSEG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, init_method="env://")


def all_reduce_flat_grads(flat_grad: torch.Tensor, async_op: bool = False):
    """sum over ranks in place; caller folds 1/world into the optimiser step.  Returns the work handle."""
    if world_size() == 1:
        return None
    return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=async_op)


class GradSync:
    """Gradient averaging over the data-parallel ranks for a FlatAdamW buffer.

    The reference wraps the model in DistributedDataParallel (/root/reference/run_training.py:82-85), whose bucketed
    all-reduce overlaps the backward pass.  Here the gradients live in one flat fp32 buffer; a model with a two-phase
    backward (`defer_backward_tail` / `backward_tail` / `tail_parameters`, models/unet.py) lets the buffer go out in two
    pieces: `start()` after the head of the backward sends everything the head finished (the suffix of the buffer
    that holds no tail gradient), the tail of the backward runs under that collective, `finish()` sends the rest and
    folds 1/world into the optimiser's gradient scale.  Models without a split backward get one all-reduce in
    `finish()`.  No collective is issued on a single rank."""

    def __init__(self, optimizer, model=None):
        import os
        self.opt = optimizer
        self.ws = world_size()
        self.split = 0
        self.net = None
        self._work = []
        net = getattr(model, "module", model)
        if (self.ws > 1 and net is not None and hasattr(net, "backward_tail")
                and not os.environ.get("MSSEG_NO_GRAD_OVERLAP")):
            o = optimizer.early_suffix_offset(net.tail_parameters())
            if 0 < o < optimizer.flat_grad.numel():
                self.split, self.net = o, net
                net.defer_backward_tail(True)

    @property
    def overlapped(self) -> bool:
        return self.net is not None

    def start(self):
        """after the head of the backward: all-reduce the finished suffix asynchronously"""
        if self.overlapped:
            self._work.append(dist.all_reduce(self.opt.flat_grad[self.split:], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """after the whole backward: all-reduce what is left, wait, and fold the 1/world average into the step"""
        if self.ws == 1:
            return
        g = self.opt.flat_grad
        self._work.append(dist.all_reduce(g[:self.split] if self.overlapped else g, op=dist.ReduceOp.SUM, async_op=True))
        for w in self._work:
            w.wait()
        self._work = []
        self.opt._gscale.mul_(1.0 / self.ws)


def convert_sync_batchnorm(model, group=True) -> int:
    """What the reference does to every model under DDP (/root/reference/run_training.py:83,
    ``SyncBatchNorm.convert_sync_batchnorm``): each BatchNorm of the build's models normalises with the statistics of the
    GLOBAL batch.  Here a BatchNorm lives inside an op record (`layers.BatchNormAct`) or behind a module attribute
    `sync_group` (SwinDepth / SwInception MLPs, the SegFormer head); this walks the module tree and switches all of them.
    `group`: a process group, True = the default group, None = back to per-rank statistics.  Returns the number of
    switched holders (0 for the InstanceNorm / LayerNorm models, which need no exchange)."""
    from .layers import BatchNormAct
    net = getattr(model, "module", model)
    n = 0
    for m in net.modules():
        if hasattr(m, "sync_group"):
            m.sync_group = group
            n += 1
        for v in vars(m).values():          # op records are plain attributes (lists / dicts of them) of their module
            stack = [v]
            while stack:
                o = stack.pop()
                if isinstance(o, BatchNormAct):
                    o.group = group
                    n += 1
                elif isinstance(o, (list, tuple)):
                    stack.extend(o)
                elif isinstance(o, dict):
                    stack.extend(o.values())
                elif hasattr(o, "op") and not isinstance(o, torch.nn.Module):
                    stack.append(o.op)
                elif hasattr(o, "norm") and not isinstance(o, torch.nn.Module):
                    stack.append(o.norm)
    return n


def all_reduce_mean(x: float) -> float:
    """/root/reference/utils/misc.py:307-315"""
    if world_size() == 1:
        return x
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    return float(t.item()) / world_size()


def shard_windows(num_windows: int, ws: int | None = None, rk: int | None = None):
    """contiguous window ranges per rank for sliding-window inference (balanced to +-1)."""
    ws = world_size() if ws is None else ws
    rk = rank() if rk is None else rk
    base, rem = divmod(num_windows, ws)
    start = rk * base + min(rk, rem)
    return start, start + base + (1 if rk < rem else 0)
