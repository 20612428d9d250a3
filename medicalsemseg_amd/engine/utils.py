"""Sliding-window inference on the GPU.

Same signature and semantics as ``/root/reference/engine/utils.py:19-159`` (the reference's fork of MONAI's
function that feeds ``(window, centers, affine)`` tuples to the predictor): constant padding up to the ROI,
``scan_interval = int(roi * (1 - overlap))``, dense windows in row-major order with the last start clamped,
Gaussian (sigma = 0.125 * roi) or constant importance map, blend of raw LOGITS ``out += w * logit; cnt += w``,
``out / cnt``, crop of the padding.  Differences, all result-preserving: the window gather, blend and
normalise are HIP kernels (``msseg_sw_gather/_blend/_normalize``); the count map has one channel instead of
``classes`` identical ones; under ``torch.distributed`` the accumulators are updated batch by batch (memory O(volume), like the reference); with
``shard_ranks=True`` window batches are dealt round-robin to the ranks and each step's logits are exchanged with one
all-gather, after which every rank blends the step's windows in the reference order (bit-identical to the single-GPU
result).
"""
from __future__ import annotations

import os

import itertools
import math
from typing import Callable, List, Sequence, Tuple

import torch
import torch.nn.functional as F

from .. import hip, parallel


def _tup(v, n):
    if isinstance(v, (int, float)):
        return (v,) * n
    v = tuple(v)
    if len(v) != n:
        raise ValueError(f"expected {n} values, got {v}")
    return v


def fall_back_tuple(roi, image):
    roi = _tup(roi, len(image))
    return tuple(int(r) if (r is not None and r > 0) else int(i) for r, i in zip(roi, image))


def get_scan_interval(image_size, roi_size, num_spatial_dims, overlap):
    return tuple(int(roi_size[i]) if roi_size[i] == image_size[i] else max(int(roi_size[i] * (1 - overlap)), 1)
                 for i in range(num_spatial_dims))


def dense_patch_starts(image_size, patch_size, scan_interval) -> List[List[int]]:
    out = []
    for d in range(len(image_size)):
        if scan_interval[d] == 0:
            num = 1
        else:
            num = int(math.ceil(float(image_size[d]) / scan_interval[d]))
            cnt = min(num, image_size[d] - patch_size[d] + 1)
            for x in range(num):
                if x * scan_interval[d] + patch_size[d] >= image_size[d]:
                    cnt = x + 1
                    break
            num = cnt
        out.append([x * scan_interval[d] - max(x * scan_interval[d] + patch_size[d] - image_size[d], 0)
                    for x in range(num)])
    return out


def window_starts(image_size, roi_size, scan_interval) -> List[Tuple[int, ...]]:
    """all window start corners, first spatial dim slowest (1000 windows for 512^3 / 96^3 / 48)."""
    return list(itertools.product(*dense_patch_starts(image_size, roi_size, scan_interval)))


_imp_cache = {}


def importance_map(patch_size: Sequence[int], mode="constant", sigma_scale=0.125, device=None) -> torch.Tensor:
    """window weights: ones, or a separable erf-integrated Gaussian (truncated at 4 sigma, zero padded) of a
    centre delta, divided by its max, floored at its smallest non-zero value."""
    patch_size = tuple(int(p) for p in patch_size)
    key = (patch_size, mode, str(sigma_scale))
    imp = _imp_cache.get(key)
    if imp is None:
        if mode == "constant":
            imp = torch.ones(patch_size, dtype=torch.float32)
        elif mode == "gaussian":
            sig = _tup(sigma_scale, len(patch_size))
            imp = torch.zeros(patch_size, dtype=torch.float32)
            imp[tuple(p // 2 for p in patch_size)] = 1.0
            for d, (p, s) in enumerate(zip(patch_size, sig)):
                sigma = p * s
                tail = int(max(float(sigma) * 4.0, 0.5) + 0.5)
                xs = torch.arange(-tail, tail + 1, dtype=torch.float32)
                t = 0.70710678 / abs(float(sigma))
                k = (0.5 * ((t * (xs + 0.5)).erf() - (t * (xs - 0.5)).erf())).clamp(min=0)
                x = imp.movedim(d, -1)
                shp = x.shape
                x = F.conv1d(x.reshape(-1, 1, shp[-1]), k.view(1, 1, -1), padding=tail)
                imp = x.reshape(shp).movedim(-1, d)
            imp = imp / imp.max()
            imp = torch.clamp(imp, min=imp[imp != 0].min().item())
        else:
            raise ValueError(f"unsupported blend mode {mode}")
        _imp_cache[key] = imp
    return imp.to(device) if device is not None else imp


class _GraphedInfer:
    """`seg = predictor.infer_cl(win)` on a static batch of channels-last windows, replayed from a captured hipGraph
    (models that declare `graph_safe` -- static shapes, no host synchronisation -- and offer `infer_cl`: channels-last
    in, channels-last logits out, no layout passes, nothing retained).  A window batch is ~70 launches of 20-500 us;
    issued from Python the small ones are launch-bound.  The packed weight images are NOT rebuilt inside the graph (an
    inference loop replays it hundreds of times on unchanged weights): `sliding_window_inference` checks them once per
    call, eagerly (`layers.PACK_REGISTRY.refresh_if_stale`)."""

    _cache = {}

    @classmethod
    def get(cls, predictor, nb, cin, roi, dev):
        key = (id(predictor), nb, cin, tuple(roi), dev)
        g = cls._cache.get(key)
        if g is None or g.predictor() is not predictor:
            g = cls._cache[key] = cls(predictor, nb, cin, roi, dev)
        return g

    def __init__(self, predictor, nb, cin, roi, dev):
        import weakref
        from .. import layers
        self.predictor = weakref.ref(predictor)
        self.win = torch.zeros(nb, *roi, cin, dtype=predictor.compute_dtype, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            predictor.infer_cl(self.win)    # warm-up: packed-weight images, workspaces
        torch.cuda.current_stream().wait_stream(side)
        layers.PACK_REGISTRY.prepare()
        layers.PACK_REGISTRY.refresh_if_stale(dev)    # images are current: the capture contains no repack
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.seg = predictor.infer_cl(self.win)

    def __call__(self):
        self.graph.replay()
        return self.seg


def sliding_window_inference(inputs: torch.Tensor, affine, roi_size, sw_batch_size: int, predictor: Callable,
                             overlap: float = 0.25, mode: str = "constant", sigma_scale=0.125,
                             padding_mode: str = "constant", cval: float = 0.0, sw_device=None, device=None,
                             *args, shard_ranks: bool = False, **kwargs) -> torch.Tensor:
    """`shard_ranks=True` (opt-in; every rank must pass the SAME `inputs`, in lock-step): window batches are dealt
    round-robin to the ranks, each step's logits are exchanged with one all-gather and blended by every rank in the
    single-rank order, so all ranks return the bit-identical single-rank result.  Default: every rank works on its own
    volume, as the reference does (validation files are partitioned per rank, data/dataset_builder.py:455-464)."""
    if inputs.dim() != 5:
        raise ValueError("expects NCDHW volumes")
    if overlap < 0 or overlap >= 1:
        raise AssertionError("overlap must be >= 0 and < 1.")
    if not inputs.is_cuda:
        raise RuntimeError("sliding_window_inference runs on the GPU only (no CPU fallback)")
    if padding_mode != "constant":
        raise NotImplementedError("only constant padding (the reference's call) is implemented")
    mode = getattr(mode, "value", mode)
    dev = inputs.device
    image_size_ = list(inputs.shape[2:])
    B, Cin = inputs.shape[0], inputs.shape[1]
    roi = fall_back_tuple(roi_size, image_size_)
    image_size = tuple(max(image_size_[i], roi[i]) for i in range(3))
    # symmetric constant padding when the volume is smaller than the ROI; realised by the gather kernel's cval
    pad_lo = [max(roi[d] - image_size_[d], 0) // 2 for d in range(3)]
    interval = get_scan_interval(image_size, roi, 3, overlap)
    starts = window_starts(image_size, roi, interval)
    num_win = len(starts)
    total = num_win * B
    nb = int(sw_batch_size)
    imp = importance_map(tuple(min(r, i) for r, i in zip(roi, image_size)), mode, sigma_scale, dev)
    vol = inputs.float().contiguous()
    ws, rk = (parallel.world_size(), parallel.rank()) if shard_ranks else (1, 0)

    # all window starts as (b, z0, y0, x0) rows, padded with unused slots to whole steps of ws * nb windows
    nsteps = -(-total // (nb * ws))
    rows_h = torch.full((nsteps * ws * nb, 4), -1, dtype=torch.int32)
    st = torch.tensor(starts, dtype=torch.int32)
    rows_h[:total, 0] = torch.arange(total, dtype=torch.int32) // num_win
    rows_h[:total, 1:] = st.repeat(B, 1)
    rows_gather = rows_h.clone()
    rows_gather[:total, 1:] -= torch.tensor(pad_lo, dtype=torch.int32)    # gather reads the unpadded volume
    rows_blend = rows_h.to(dev)
    rows_gather = rows_gather.to(dev)

    fast = (getattr(predictor, "graph_safe", False) and hasattr(predictor, "infer_cl") and not torch.is_grad_enabled()
            and not args and not kwargs and not os.environ.get("MSSEG_NO_SW_GRAPH"))
    out = cnt = gathered = None

    def accumulators(ncls):
        return (torch.zeros(B, ncls, *image_size, dtype=torch.float32, device=dev),
                torch.zeros(B, *image_size, dtype=torch.float32, device=dev))

    if fast:
        # gather -> graph replay of the forward -> [all-gather] -> blend, per step; the accumulators are updated step by
        # step (memory O(volume), as the reference), windows and logits never leave the channels-last compute dtype
        g = _GraphedInfer.get(predictor, nb, Cin, roi, dev)
        from .. import layers
        layers.PACK_REGISTRY.refresh_if_stale(dev)    # parameters may have changed since the graph was captured
        ncls = predictor.out_channels
        out, cnt = accumulators(ncls)
        # Sharded: the step's logits are exchanged as COMPACT channels-first rows (ncls of the 8 channels of a 16-byte
        # logits row are real: 6 instead of 16 bytes per voxel on the links at 3 classes) and the all-gather of step i
        # runs under the forward of step i + 1 (two buffer pairs; the blend of a step follows its exchange, one step
        # late, in the same global window order -> still bit-identical to the single-rank result).
        comp = gath = None
        pending = None                                    # (work handle, gathered buffer, step) of the exchange in flight

        def blend_step(item):
            work, buf, step = item
            work.wait()
            hip.sw_blend_batch(buf, imp, out, cnt, rows_blend[step * ws * nb:(step + 1) * ws * nb], ws * nb)

        for i in range(nsteps):
            g0 = (i * ws + rk) * nb                       # this rank's batch of the step (all slots unused: idle replay)
            hip.sw_gather_batch(vol, g.win, rows_gather[g0:g0 + nb], nb, cval, channels_last_ld=Cin)
            seg = g()
            if ws == 1:
                hip.sw_blend_batch(seg, imp, out, cnt, rows_blend[g0:g0 + nb], nb, channels_last_ld=seg.shape[-1])
                continue
            if comp is None:
                comp = [torch.empty((nb, ncls) + tuple(seg.shape[1:4]), dtype=seg.dtype, device=dev) for _ in range(2)]
                gath = [torch.empty((ws * nb, ncls) + tuple(seg.shape[1:4]), dtype=seg.dtype, device=dev) for _ in range(2)]
            hip.to_channels_first(seg[..., :ncls], comp[i & 1])
            work = torch.distributed.all_gather_into_tensor(gath[i & 1], comp[i & 1], async_op=True)
            if pending is not None:
                blend_step(pending)
            pending = (work, gath[i & 1], i)
        if pending is not None:
            blend_step(pending)
    else:
        win = None
        ncls = int(getattr(predictor, "out_channels", 0) or 0)
        for i in range(nsteps):
            g0 = (i * ws + rk) * nb                       # this rank's batch of the step
            idxs = [g for g in range(g0, g0 + nb) if g < total]
            seg = None
            if idxs:
                centers = torch.tensor([[(starts[g % num_win][d] + roi[d] - roi[d] // 2) / image_size[d] for d in range(3)]
                                        for g in idxs], dtype=torch.float32, device=dev)
                if sw_batch_size == 1:
                    centers = centers.unsqueeze(0)  # reference quirk (engine/utils.py:131-132)
                if win is None:
                    win = torch.zeros(nb, Cin, *roi, dtype=torch.float32, device=dev)
                hip.sw_gather_batch(vol, win, rows_gather[g0:g0 + nb], nb, cval)
                seg = predictor((win[:len(idxs)], centers, affine), *args, **kwargs)
                if seg.dtype not in (torch.float32, torch.bfloat16):
                    seg = seg.float()
                ncls = seg.shape[1]
            if out is None:
                if ws > 1:   # a rank without a window in step 0 (fewer windows than ranks) learns the class count here
                    t = torch.tensor([ncls], dtype=torch.int64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
                    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                    ncls = int(t.item())
                out, cnt = accumulators(ncls)
            if ws == 1:
                hip.sw_blend_batch(seg.contiguous(), imp, out, cnt, rows_blend[g0:g0 + nb], len(idxs))
                continue
            # sharded: one all-gather per step, then every rank blends the step's ws * nb windows in global order
            mine = torch.zeros(nb, ncls, *roi, dtype=torch.float32, device=dev)
            if seg is not None:
                mine[:len(idxs)] = seg.float()
            if gathered is None:
                gathered = torch.empty(ws * nb, ncls, *roi, dtype=torch.float32, device=dev)
            torch.distributed.all_gather_into_tensor(gathered, mine)
            hip.sw_blend_batch(gathered, imp, out, cnt, rows_blend[i * ws * nb:(i + 1) * ws * nb], ws * nb)
    for b in range(B):
        hip.sw_normalize(out[b], cnt[b])
    sl = [slice(None), slice(None)] + [slice(pad_lo[d], pad_lo[d] + image_size_[d]) for d in range(3)]
    return out[tuple(sl)]
