"""Oracle (test infrastructure): conv encoder/decoder blocks, fp32 torch-CPU.

Restates, with stock torch ops, the MONAI blocks the reference instantiates:

* ``UnetResBlock`` / ``UnetrBasicBlock`` / ``UnetrUpBlock`` / ``UnetOutBlock``
  -- call sites ``/root/reference/models/segmentors/swin_unetr.py:73-130``
  (SURVEY.md 8(a) rows A3-A5).
* ``BasicUNet`` (``TwoConv``/``Down``/``UpCat``) -- named by BASELINE.json
  configs 1-3.

MONAI is absent from the container: these are **parity unpinned** against
MONAI; module/parameter names follow MONAI's state-dict layout so that
checkpoints keep loading.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


class _ConvOnly(nn.Sequential):
    """MONAI ``Convolution(conv_only=True)``: a Sequential holding ``conv``."""

    def __init__(self, cin, cout, k, stride=1, bias=False, transposed=False):
        super().__init__()
        if transposed:
            conv = nn.ConvTranspose3d(cin, cout, kernel_size=k, stride=stride, bias=bias)
        else:
            pad = (k - 1) // 2 if isinstance(k, int) else tuple((kk - 1) // 2 for kk in k)
            conv = nn.Conv3d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=bias)
        self.add_module("conv", conv)


class UnetResBlock(nn.Module):
    """y = lrelu(norm2(conv2(lrelu(norm1(conv1(x))))) + r),
    r = norm3(conv3(x)) iff in != out (or stride != 1) else x.
    convs bias-free, InstanceNorm3d affine=False eps 1e-5, LeakyReLU(0.01)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1):
        super().__init__()
        self.conv1 = _ConvOnly(in_channels, out_channels, kernel_size, stride)
        self.conv2 = _ConvOnly(out_channels, out_channels, kernel_size, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=0.01, inplace=False)
        self.norm1 = nn.InstanceNorm3d(out_channels)
        self.norm2 = nn.InstanceNorm3d(out_channels)
        self.downsample = in_channels != out_channels or stride != 1
        if self.downsample:
            self.conv3 = _ConvOnly(in_channels, out_channels, 1, stride)
            self.norm3 = nn.InstanceNorm3d(out_channels)

    def forward(self, inp):
        residual = inp
        out = self.lrelu(self.norm1(self.conv1(inp)))
        out = self.norm2(self.conv2(out))
        if self.downsample:
            residual = self.norm3(self.conv3(residual))
        return self.lrelu(out + residual)


class UnetrBasicBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1):
        super().__init__()
        self.layer = UnetResBlock(in_channels, out_channels, kernel_size, stride)

    def forward(self, x):
        return self.layer(x)


class UnetrUpBlock(nn.Module):
    """up = ConvT(k=s=upsample_kernel_size, bias=False)(x);
    y = UnetResBlock(2*out -> out)(cat([up, skip], 1))."""

    def __init__(self, in_channels, out_channels, kernel_size=3, upsample_kernel_size=2):
        super().__init__()
        self.transp_conv = _ConvOnly(in_channels, out_channels, upsample_kernel_size,
                                     upsample_kernel_size, transposed=True)
        self.conv_block = UnetResBlock(out_channels + out_channels, out_channels, kernel_size, 1)

    def forward(self, inp, skip):
        out = self.transp_conv(inp)
        out = torch.cat((out, skip), dim=1)
        return self.conv_block(out)


class UnetOutBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = _ConvOnly(in_channels, out_channels, 1, 1, bias=True)

    def forward(self, x):
        return self.conv(x)


# --------------------------------------------------------------------------
# BasicUNet (MONAI): features=(32,32,64,128,256,32), LeakyReLU(0.1),
# InstanceNorm3d(affine=True), bias=True, deconv upsampling.
# --------------------------------------------------------------------------
class _ADN(nn.Sequential):
    """MONAI ADN with ordering "NDA" and dropout 0: ``N`` then ``A``."""

    def __init__(self, ch, slope):
        super().__init__()
        self.add_module("N", nn.InstanceNorm3d(ch, affine=True))
        self.add_module("A", nn.LeakyReLU(negative_slope=slope, inplace=False))


class _ConvADN(nn.Sequential):
    def __init__(self, cin, cout, slope, bias):
        super().__init__()
        self.add_module("conv", nn.Conv3d(cin, cout, 3, padding=1, bias=bias))
        self.add_module("adn", _ADN(cout, slope))


class TwoConv(nn.Sequential):
    def __init__(self, cin, cout, slope=0.1, bias=True):
        super().__init__()
        self.add_module("conv_0", _ConvADN(cin, cout, slope, bias))
        self.add_module("conv_1", _ConvADN(cout, cout, slope, bias))


class Down(nn.Sequential):
    def __init__(self, cin, cout, slope=0.1, bias=True):
        super().__init__()
        self.add_module("max_pooling", nn.MaxPool3d(2))
        self.add_module("convs", TwoConv(cin, cout, slope, bias))


class _UpSampleDeconv(nn.Sequential):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.add_module("deconv", nn.ConvTranspose3d(cin, cout, kernel_size=2, stride=2, bias=bias))


class UpCat(nn.Module):
    def __init__(self, in_chns, cat_chns, out_chns, slope=0.1, bias=True, halves=True):
        super().__init__()
        up_chns = in_chns // 2 if halves else in_chns
        self.upsample = _UpSampleDeconv(in_chns, up_chns, bias)
        self.convs = TwoConv(cat_chns + up_chns, out_chns, slope, bias)

    def forward(self, x, x_e):
        x_0 = self.upsample(x)
        # MONAI pads odd encoder dims with replicate padding
        dims = x.dim() - 2
        sp = [0] * (dims * 2)
        for i in range(dims):
            if x_e.shape[-i - 1] != x_0.shape[-i - 1]:
                sp[i * 2 + 1] = 1
        if any(sp):
            x_0 = F.pad(x_0, sp, "replicate")
        return self.convs(torch.cat([x_e, x_0], dim=1))  # skip first


class BasicUNet(nn.Module):
    def __init__(self, in_channels=1, out_channels=2,
                 features: Sequence[int] = (32, 32, 64, 128, 256, 32), slope=0.1, bias=True):
        super().__init__()
        fea = tuple(features)
        self.conv_0 = TwoConv(in_channels, fea[0], slope, bias)
        self.down_1 = Down(fea[0], fea[1], slope, bias)
        self.down_2 = Down(fea[1], fea[2], slope, bias)
        self.down_3 = Down(fea[2], fea[3], slope, bias)
        self.down_4 = Down(fea[3], fea[4], slope, bias)
        self.upcat_4 = UpCat(fea[4], fea[3], fea[3], slope, bias)
        self.upcat_3 = UpCat(fea[3], fea[2], fea[2], slope, bias)
        self.upcat_2 = UpCat(fea[2], fea[1], fea[1], slope, bias)
        self.upcat_1 = UpCat(fea[1], fea[0], fea[5], slope, bias, halves=False)
        self.final_conv = nn.Conv3d(fea[5], out_channels, kernel_size=1)

    def forward(self, x):
        # the engine's model contract: model((vol, crop_loc, affine))
        # (/root/reference/engine/train.py:58-61)
        if isinstance(x, (tuple, list)):
            x = x[0]
        x0 = self.conv_0(x)
        x1 = self.down_1(x0)
        x2 = self.down_2(x1)
        x3 = self.down_3(x2)
        x4 = self.down_4(x3)
        u4 = self.upcat_4(x4, x3)
        u3 = self.upcat_3(u4, x2)
        u2 = self.upcat_2(u3, x1)
        u1 = self.upcat_1(u2, x0)
        return self.final_conv(u1)


UNET_FEATURES = {"UNet": (32, 32, 64, 128, 256, 32), "UNetSmall": (16, 16, 32, 64, 128, 16)}
