"""Sliding-window inference on the GPU.

Same signature and semantics as ``/root/reference/engine/utils.py:19-159`` (the reference's fork of MONAI's
function that feeds ``(window, centers, affine)`` tuples to the predictor): constant padding up to the ROI,
``scan_interval = int(roi * (1 - overlap))``, dense windows in row-major order with the last start clamped,
Gaussian (sigma = 0.125 * roi) or constant importance map, blend of raw LOGITS ``out += w * logit; cnt += w``,
``out / cnt``, crop of the padding.  Differences, all result-preserving: the window gather, blend and
normalise are HIP kernels (``msseg_sw_gather/_blend/_normalize``); the count map has one channel instead of
``classes`` identical ones; under ``torch.distributed`` the windows are sharded across ranks and their logits
exchanged with ONE all-gather, after which every rank blends all windows in the reference order (bit-identical
to the single-GPU result).
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


class _GraphedPredictor:
    """Replays `predictor((win, None, None))` on a static window batch from a captured hipGraph (models that declare
    `graph_safe`: static shapes, no host synchronisation, no autograd).  A window batch is ~70 kernel launches of
    20-100 us each: issued from Python they are launch-bound.  The captured graph starts with the batched weight
    re-packing, so a replay always sees the current parameters."""

    _cache = {}

    @classmethod
    def get(cls, predictor, win: torch.Tensor):
        key = (id(predictor), tuple(win.shape), win.dtype, win.device)
        g = cls._cache.get(key)
        if g is None or g.predictor() is not predictor:
            g = cls._cache[key] = cls(predictor, win)
        return g

    def __init__(self, predictor, win: torch.Tensor):
        import weakref
        from .. import layers
        self.predictor = weakref.ref(predictor)
        self.win = torch.zeros_like(win)
        side = torch.cuda.Stream(device=win.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            predictor((self.win, None, None))
        torch.cuda.current_stream().wait_stream(side)
        layers.PACK_REGISTRY.prepare()
        layers.bump_weights_epoch()   # the capture then starts with the batched weight re-packing
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = predictor((self.win, None, None))

    def __call__(self):
        self.graph.replay()
        return self.out


def sliding_window_inference(inputs: torch.Tensor, affine, roi_size, sw_batch_size: int, predictor: Callable,
                             overlap: float = 0.25, mode: str = "constant", sigma_scale=0.125,
                             padding_mode: str = "constant", cval: float = 0.0, sw_device=None, device=None,
                             *args, **kwargs) -> torch.Tensor:
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
    imp = importance_map(tuple(min(r, i) for r, i in zip(roi, image_size)), mode, sigma_scale, dev)
    vol = inputs.float().contiguous()

    ws, rk = parallel.world_size(), parallel.rank()
    lo, hi = parallel.shard_windows(total, ws, rk)
    per_rank = -(-total // ws)
    my_logits = None
    # graph replay of the window forward: models that ignore (centers, affine) and declare themselves graph-safe
    graphed = None
    if (getattr(predictor, "graph_safe", False) and not torch.is_grad_enabled() and not args and not kwargs
            and not os.environ.get("MSSEG_NO_SW_GRAPH") and hi - lo >= sw_batch_size):
        graphed = _GraphedPredictor.get(predictor, torch.empty(sw_batch_size, Cin, *roi, dtype=torch.float32, device=dev))
    for g in range(lo, hi, sw_batch_size):
        idxs = list(range(g, min(g + sw_batch_size, hi)))
        if graphed is not None:
            win = graphed.win   # a short last batch keeps the previous batch's windows in the unused slots
        else:
            win = torch.empty(len(idxs), Cin, *roi, dtype=torch.float32, device=dev)
        centers = []
        for j, idx in enumerate(idxs):
            b, st = idx // num_win, starts[idx % num_win]
            hip.sw_gather(vol[b], win[j], tuple(st[d] - pad_lo[d] for d in range(3)), cval)
            centers.append([(st[d] + roi[d] - roi[d] // 2) / image_size[d] for d in range(3)])
        if graphed is not None:
            seg = graphed()[:len(idxs)]
        else:
            centers = torch.tensor(centers, dtype=torch.float32, device=dev)
            if sw_batch_size == 1:
                centers = centers.unsqueeze(0)  # reference quirk (engine/utils.py:131-132)
            seg = predictor((win, centers, affine), *args, **kwargs)
        if my_logits is None:
            ncls = seg.shape[1]
            my_logits = torch.zeros(per_rank, ncls, *roi, dtype=torch.float32, device=dev)
        my_logits[g - lo:g - lo + len(idxs)] = seg.float()
    if my_logits is None:
        raise RuntimeError("a rank received no window")
    if ws > 1:
        gathered = torch.empty(ws * per_rank, *my_logits.shape[1:], dtype=torch.float32, device=dev)
        torch.distributed.all_gather_into_tensor(gathered, my_logits)
    else:
        gathered = my_logits
    ncls = gathered.shape[1]
    out = torch.zeros(B, ncls, *image_size, dtype=torch.float32, device=dev)
    cnt = torch.zeros(B, *image_size, dtype=torch.float32, device=dev)
    for r in range(ws):
        rlo, rhi = parallel.shard_windows(total, ws, r)
        for idx in range(rlo, rhi):
            b, st = idx // num_win, starts[idx % num_win]
            hip.sw_blend(gathered[r * per_rank + idx - rlo], imp, out[b], cnt[b], st)
    for b in range(B):
        hip.sw_normalize(out[b], cnt[b])
    sl = [slice(None), slice(None)] + [slice(pad_lo[d], pad_lo[d] + image_size_[d]) for d in range(3)]
    return out[tuple(sl)]
