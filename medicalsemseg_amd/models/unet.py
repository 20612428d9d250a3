"""3-D UNet (MONAI ``BasicUNet`` topology) on the HIP kernels.

BASELINE.json configs 2-3 name "UNet base (MONAI BasicUNet) 1->3cls": features (32,32,64,128,256,32),
LeakyReLU(0.1), InstanceNorm3d(affine=True), bias=True, deconv upsampling, skip FIRST in the concat
(SURVEY.md 8(a) row A15).  The reference itself ships no UNet; ``build_model`` gains ``cfg.model in
{'UNet','UNetSmall'}`` branches.  Module / parameter names follow MONAI's
state-dict layout (``conv_0.conv_0.conv.weight``, ``upcat_4.upsample.deconv.weight``, ``final_conv.bias`` ...).

The sub-modules below only HOLD parameters (with torch's default initialisation, which is what MONAI's
layers use); the arithmetic is one ``torch.autograd.Function`` that runs the whole network forward and
backward through ``layers.py`` -- concat buffers are written in place by their producers, the max-pool
gradient is accumulated into the skip gradient, and parameter gradients land directly in ``.grad``.
"""
from __future__ import annotations

import os

from typing import Sequence

import torch
import torch.nn as nn

from .. import hip, layers
from ..layers import Conv1, Conv3, ConvNormAct, Deconv2, InstNormAct, maxpool_fwd

UNET_FEATURES = {"UNet": (32, 32, 64, 128, 256, 32), "UNetSmall": (16, 16, 32, 64, 128, 16)}
LOGIT_LD = 8  # channel stride of the internal logits / dlogits buffers (16-byte rows for both dtypes)


class _ADN(nn.Sequential):
    def __init__(self, ch):
        super().__init__()
        self.add_module("N", nn.InstanceNorm3d(ch, affine=True))


class _ConvADN(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.add_module("conv", nn.Conv3d(cin, cout, 3, padding=1, bias=True))
        self.add_module("adn", _ADN(cout))


class _TwoConv(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.add_module("conv_0", _ConvADN(cin, cout))
        self.add_module("conv_1", _ConvADN(cout, cout))


class _Down(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.add_module("convs", _TwoConv(cin, cout))


class _Up(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.add_module("deconv", nn.ConvTranspose3d(cin, cout, kernel_size=2, stride=2, bias=True))


class _UpCat(nn.Module):
    def __init__(self, in_chns, cat_chns, out_chns, halves=True):
        super().__init__()
        up = in_chns // 2 if halves else in_chns
        self.upsample = _Up(in_chns, up)
        self.convs = _TwoConv(cat_chns + up, out_chns)


def _cna(m: _ConvADN, slope):
    return ConvNormAct(Conv3(m.conv.weight, m.conv.bias), InstNormAct(m.adn.N.weight, m.adn.N.bias, slope))


class UNet(nn.Module):
    """``model((vol[B,C,D,H,W], crop_loc, affine)) -> logits[B,out,D,H,W]`` (fp32, NCDHW).

    compute_dtype: torch.bfloat16 (speed) or torch.float32 (exact-fp32 MFMA path used for parity)."""

    graph_safe = True   # static shapes, no host synchronisation: engine.utils may replay the forward from a hipGraph

    def __init__(self, in_channels=1, out_channels=2, features: Sequence[int] = UNET_FEATURES["UNet"],
                 compute_dtype=torch.bfloat16, slope=0.1):
        super().__init__()
        f = tuple(features)
        self.in_channels, self.out_channels, self.features = in_channels, out_channels, f
        self.compute_dtype = compute_dtype
        self.slope = slope
        self.conv_0 = _TwoConv(in_channels, f[0])
        self.down_1 = _Down(f[0], f[1])
        self.down_2 = _Down(f[1], f[2])
        self.down_3 = _Down(f[2], f[3])
        self.down_4 = _Down(f[3], f[4])
        self.upcat_4 = _UpCat(f[4], f[3], f[3])
        self.upcat_3 = _UpCat(f[3], f[2], f[2])
        self.upcat_2 = _UpCat(f[2], f[1], f[1])
        self.upcat_1 = _UpCat(f[1], f[0], f[5], halves=False)
        self.final_conv = nn.Conv3d(f[5], out_channels, kernel_size=1)
        self._build_ops()

    # ---- two-phase backward (data-parallel overlap) ----
    # With `defer_backward_tail(True)` the autograd backward stops after the decoder and the bottom encoder level
    # (86 % of the weights, about 2/3 of the backward time) and `backward_tail()` runs encoder levels 3..0.  The caller
    # starts the all-reduce of the finished gradients in between (parallel.GradSync), which is what DDP's bucketed
    # overlap does for the reference (/root/reference/run_training.py:82-85).
    def defer_backward_tail(self, on: bool = True):
        self._defer_tail = bool(on)
        self._pending_tail = None
        return self

    def tail_parameters(self):
        """parameters whose gradients `backward_tail()` produces"""
        mods = (self.conv_0, self.down_1, self.down_2, self.down_3)
        return [p for m in mods for p in m.parameters()]

    def backward_tail(self):
        pend, self._pending_tail = getattr(self, "_pending_tail", None), None
        if pend is not None:
            with torch.no_grad():
                _UNetFn._run_tail(self, *pend)

    def _build_ops(self):
        s = self.slope
        self._enc = [(_cna(self.conv_0.conv_0, s), _cna(self.conv_0.conv_1, s))]
        for d in (self.down_1, self.down_2, self.down_3, self.down_4):
            self._enc.append((_cna(d.convs.conv_0, s), _cna(d.convs.conv_1, s)))
        self._dec = []
        for u in (self.upcat_4, self.upcat_3, self.upcat_2, self.upcat_1):
            self._dec.append((Deconv2(u.upsample.deconv.weight, u.upsample.deconv.bias),
                              _cna(u.convs.conv_0, s), _cna(u.convs.conv_1, s)))
        self._final = Conv1(self.final_conv.weight, self.final_conv.bias)

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._build_ops()  # parameters were replaced (.to / .cuda): re-bind the op objects
        return r

    def infer_cl(self, x_cl):
        """inference entry for callers that already hold channels-last windows in the compute dtype (sliding-window
        inference): [N, D, H, W, Cin] -> logits [N, D, H, W, LOGIT_LD] (the first out_channels are valid); no autograd,
        nothing retained, no layout conversion."""
        if x_cl.dtype != self.compute_dtype or not x_cl.is_cuda or not x_cl.is_contiguous():
            raise ValueError("infer_cl expects a contiguous channels-last GPU tensor in the compute dtype")
        if any(int(d) % 16 for d in x_cl.shape[1:4]):
            raise ValueError(f"UNet needs spatial dims divisible by 16, got {tuple(x_cl.shape[1:4])}")
        with torch.no_grad():
            return _forward_cl(self, x_cl, False)[0]

    def forward(self, x_in):
        vol = x_in[0] if isinstance(x_in, (tuple, list)) else x_in
        if not vol.is_cuda:
            raise RuntimeError("medicalsemseg_amd.UNet runs on the GPU only (no CPU fallback); "
                               "the CPU oracle lives in oracle/ and is test infrastructure")
        if any(int(d) % 16 for d in vol.shape[2:]):
            raise ValueError(f"UNet needs spatial dims divisible by 16, got {tuple(vol.shape[2:])}")
        params = [p for p in self.parameters()]
        return _UNetFn.apply(self, vol, *params)


def _forward_cl(net: "UNet", x, keep: bool):
    """x: channels-last [N, D, H, W, Cin] in the compute dtype -> (logits_cl [N, D, H, W, LOGIT_LD], saved, last, skips).
    keep=False (inference): nothing is retained, every intermediate is released as soon as its consumer is issued."""
    prev = layers.INFERENCE_FORWARD
    layers.INFERENCE_FORWARD = not keep
    try:
        return _forward_cl_body(net, x, keep)
    finally:
        layers.INFERENCE_FORWARD = prev


def _forward_cl_body(net: "UNet", x, keep: bool):
    T = net.compute_dtype
    N, D, H, W, _ = x.shape
    dev = x.device
    f = net.features
    saved = {"enc": [], "dec": []}
    # concat buffers [skip | up] for the 4 decoder levels (level i uses encoder output i).  Inference (nothing kept): at the
    # 32 + 32-channel levels the first decoder conv runs as two launches of the ping-pong kernel on the skip and on the
    # upsampled tensor (Conv3.fwd_split): those levels get two dense tensors instead of the concat buffer.
    cat, split = [], [False] * 4
    for i in range(4):
        sh = (N, D >> i, H >> i, W >> i)
        up_ch = f[i + 1] // 2 if i > 0 else f[1]
        split[i] = (not keep) and net._dec[3 - i][1].conv.split_ok(sh, T, f[i], up_ch)
        if split[i]:
            cat.append((torch.empty(sh + (f[i],), dtype=T, device=dev), torch.empty(sh + (up_ch,), dtype=T, device=dev)))
        else:
            cat.append(torch.empty(sh + (f[i] + up_ch,), dtype=T, device=dev))
    cur = x
    skips = []
    for lvl, (c0, c1) in enumerate(net._enc):
        a0, s0 = c0.fwd(cur)
        out = (cat[lvl][0] if split[lvl] else cat[lvl][..., :f[lvl]]) if lvl < 4 else None
        # the level's output goes into the decoder's concat buffer and, max-pooled by the same kernel, to the next level
        pooled = torch.empty(N, D >> (lvl + 1), H >> (lvl + 1), W >> (lvl + 1), f[lvl], dtype=T, device=dev) if lvl < 4 else None
        a1, s1 = c1.fwd(a0, out, pooled=pooled)
        if keep:
            saved["enc"].append((s0, s1))
            skips.append(a1)
        cur = pooled if lvl < 4 else a1
        del a0, s0, s1
    # decoder: levels 3..0
    for j, (up, c0, c1) in enumerate(net._dec):
        lvl = 3 - j
        up_in = cur
        if split[lvl]:
            up.fwd(up_in, cat[lvl][1])
            y0, st0 = c0.conv.fwd_split(cat[lvl][0], cat[lvl][1])
            a0, st0 = c0.norm.fwd(y0, stats=st0)
            s0 = None
        else:
            up.fwd(up_in, cat[lvl][..., f[lvl]:])
            a0, s0 = c0.fwd(cat[lvl])
        if lvl == 0:
            # last unit: only its raw conv output + statistics; the head normalises on load (no activation tensor)
            y1, st1 = c1.conv.fwd(a0, want_stats=True)
            if net._final.head_norm_ok(y1):
                a1, s1 = None, (a0, y1, st1, None)
            else:
                a1, st1 = c1.norm.fwd(y1, stats=st1)
                s1 = (a0, y1, st1, a1)
        else:
            a1, s1 = c1.fwd(a0)
        if keep:
            saved["dec"].append((up_in, s0, s1))
        else:
            cat[lvl] = None
        cur = a1
        last_unit = s1
        del a0, s0, s1, up_in
    logits_cl = torch.empty(N, D, H, W, LOGIT_LD, dtype=T, device=dev)
    if cur is None:
        net._final.fwd_norm(last_unit[1], last_unit[2], net._dec[3][2].norm, logits_cl[..., :net.out_channels])
    else:
        net._final.fwd(cur, logits_cl[..., :net.out_channels])
    return logits_cl, saved, cur, skips


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net: UNet, vol, *params):
        T = net.compute_dtype
        N, Cin, D, H, W = vol.shape
        dev = vol.device
        need_grad = any(ctx.needs_input_grad)
        x = torch.empty(N, D, H, W, Cin, dtype=T, device=dev)
        hip.to_channels_last(vol.float() if vol.dtype not in (torch.float32, torch.bfloat16) else vol, x)
        logits_cl, saved, cur, skips = _forward_cl(net, x, need_grad)
        if need_grad:
            ctx.net, ctx.saved, ctx.last, ctx.skips = net, saved, cur, skips
        ctx.set_materialize_grads(False)
        # The logits stay in their channels-last buffer (16-byte rows, compute dtype); what the caller gets is the
        # [B, C, D, H, W] view of it, so the fp32 NCDHW round trip between the network and the loss never happens
        # (losses.DiceCELoss reads / writes the channels-last buffers directly; any other consumer sees an ordinary
        # strided tensor).
        return logits_cl[..., :net.out_channels].permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, dlogits):
        net, saved = ctx.net, ctx.saved
        n_in = 2 + len(list(net.parameters()))
        if dlogits is None:
            return (None,) * n_in
        T = net.compute_dtype
        f = net.features
        N, C, D, H, W = dlogits.shape
        dev = dlogits.device
        from ..losses import channels_last_grad
        dl = channels_last_grad(dlogits, LOGIT_LD, T)   # the loss wrote [N, D, H, W, LOGIT_LD] rows (zero padded) itself
        if dl is None:
            dl = torch.zeros(N, D, H, W, LOGIT_LD, dtype=T, device=dev)
            hip.to_channels_last(dlogits.contiguous(), dl[..., :C])
        layers.WGRAD_SIDE.begin(dev)
        try:
            return _UNetFn._backward_body(ctx, net, saved, dl, f, n_in)
        finally:
            layers.WGRAD_SIDE.join()

    @staticmethod
    def _backward_body(ctx, net, saved, dl, f, n_in):
        side = layers.WGRAD_SIDE
        side.set_mode(side.DEFER if os.environ.get("MSSEG_WGRAD_STREAM") == "defer" else side.INLINE)
        # Every gradient that reaches the second conv+norm unit of a level comes from a flat input-gradient kernel (the
        # 1x1x1 output conv, a transposed conv): those kernels also produce the unit's InstanceNorm-backward sums
        # (`red_c1`), which saves its separate reduction pass over three full tensors.
        def unit_norm(cna, sv):
            if os.environ.get("MSSEG_NO_FLAT_INBWD"):    # A/B switch: separate reduction pass
                return None
            return (cna.norm, sv[1], sv[2], sv[3])      # (InstNormAct, yraw, stats, act) of a ConvNormAct's saved tuple

        def pair(r):                                     # (dx, red) with or without the fused sums
            return r if isinstance(r, tuple) else (r, None)

        if ctx.last is None:   # head fused with the last unit's InstanceNorm + LeakyReLU (forward never wrote its activation)
            lu = saved["dec"][3][2]
            g, red_c1 = net._final.bwd_norm(dl, LOGIT_LD, net._dec[3][2].norm, lu[1], lu[2])
        else:
            g, red_c1 = pair(net._final.bwd(ctx.last, dl, True, dy_channels=LOGIT_LD,
                                            next_norm=unit_norm(net._dec[3][2], saved["dec"][3][2])))
        skip_grads = [None] * 4
        for j in range(3, -1, -1):  # decoder levels 0..3 in reverse order of execution
            up, c0, c1 = net._dec[j]
            lvl = 3 - j
            if lvl == 2:   # the deep levels begin: their small kernels run under the deferred full-chip wgrads
                side.flush()
                side.set_mode(side.SIDE)
            up_in, s0, s1 = saved["dec"][j]
            g, red = c1.bwd(s1, g, True, red=red_c1, next_saved=s0, next_cna=c0)
            dcat = c0.bwd(s0, g, True, red=red)
            skip_grads[lvl] = dcat[..., :f[lvl]]
            below = unit_norm(net._dec[j - 1][2], saved["dec"][j - 1][2]) if j > 0 else unit_norm(net._enc[4][1], saved["enc"][4][1])
            g, red_c1 = pair(up.bwd(up_in, dcat[..., f[lvl]:], True, next_norm=below))
        # encoder, bottom-up
        g = _UNetFn._encoder_bwd(net, saved, g, red_c1, skip_grads, (4,))
        if getattr(net, "_defer_tail", False):
            net._pending_tail = (saved, g, skip_grads)     # levels 3..0 run in net.backward_tail()
        else:
            _UNetFn._encoder_bwd(net, saved, g, None, skip_grads, (3, 2, 1, 0))
        ctx.saved = None
        return (None,) * n_in

    @staticmethod
    def _run_tail(net, saved, g, skip_grads):
        side = layers.WGRAD_SIDE
        side.begin(g.device)
        try:
            side.set_mode(side.SIDE)
            _UNetFn._encoder_bwd(net, saved, g, None, skip_grads, (3, 2, 1, 0))
        finally:
            side.join()

    @staticmethod
    def _encoder_bwd(net, saved, g, red_c1, skip_grads, levels):
        side = layers.WGRAD_SIDE
        for lvl in levels:
            c0, c1 = net._enc[lvl]
            s0, s1 = saved["enc"][lvl]
            if lvl == 1:
                side.set_mode(side.INLINE)
            red_in = red_c1 if lvl == 4 else None
            if lvl < 4:
                # gradient of the skip (written by the decoder) + max-pool path from the level below: formed by the kernel
                # that also computes the second unit's InstanceNorm-backward sums
                fused = c1.norm.pool_bwd_reduce(s1[1], s1[2], skip_grads[lvl], g)
                if fused is not None:
                    g, red_in = fused
                else:
                    hip.maxpool2_bwd(s1[3], g, skip_grads[lvl], accumulate=True)
                    g = skip_grads[lvl]
            g, red = c1.bwd(s1, g, True, red=red_in, next_saved=s0, next_cna=c0)
            g = c0.bwd(s0, g, need_dx=(lvl > 0), red=red)
        return g
