"""Oracle (test infrastructure): the reference's UNETR conv decoder, fp32 torch-CPU.

Restates ``/root/reference/models/segmentors/unetr.py:9-52`` (conv / transposed-conv / BatchNorm blocks) and ``:195-289``
(``UNETRC``: U-shaped fusion of four token feature maps of a ViT-style encoder with the raw input).  SURVEY.md 8(a) row
A13.  Pinned by ``tests/golden/unetrc_ref.npz``, which ``oracle/gen_golden.py`` produced by running the reference's own
class on the same deterministic weights (``tests/test_oracle_golden.py``).

Layout (state-dict keys equal the reference's):
  conv unit   ``cbr(i, o)``   = Conv3d k3 p1 (+bias) -> BatchNorm3d -> ReLU                     keys ``block.0.block``, ``block.1``
  deconv unit ``dbr(i, o)``   = ConvTranspose3d k2 s2 (+bias) -> conv unit (o -> o)             keys ``block.0.block``, ``block.1.block``, ``block.2``
"""
from __future__ import annotations

import torch
import torch.nn as nn


class _Wrap(nn.Module):
    """a module kept under the attribute name ``block`` (the reference wraps single convs that way)"""

    def __init__(self, inner):
        super().__init__()
        self.block = inner

    def forward(self, x):
        return self.block(x)


def up2(i, o):
    return _Wrap(nn.ConvTranspose3d(i, o, kernel_size=2, stride=2))


def conv(i, o, k=3):
    return _Wrap(nn.Conv3d(i, o, kernel_size=k, stride=1, padding=(k - 1) // 2))


def cbr(i, o):
    return _Wrap(nn.Sequential(conv(i, o), nn.BatchNorm3d(o), nn.ReLU(True)))


def dbr(i, o):
    return _Wrap(nn.Sequential(up2(i, o), conv(o, o), nn.BatchNorm3d(o), nn.ReLU(True)))


class UNETRC(nn.Module):
    """encoder(x) -> four token maps [B, L, E] (depths 3, 6, 9, 12 of the ViT); L = prod(vol / patch)"""

    def __init__(self, encoder, in_chans=1, output_dim=3):
        super().__init__()
        self.encoder = encoder
        E = encoder.embed_dim
        self.embed_dim = E
        self.grid = [int(v // p) for v, p in zip(encoder.vol_size, encoder.patch_size)]
        self.decoder0 = nn.Sequential(cbr(in_chans, 32), cbr(32, 64))
        self.decoder3 = nn.Sequential(dbr(E, 512), dbr(512, 256), dbr(256, 128))
        self.decoder6 = nn.Sequential(dbr(E, 512), dbr(512, 256))
        self.decoder9 = dbr(E, 512)
        self.decoder12_upsampler = up2(E, 512)
        self.decoder9_upsampler = nn.Sequential(cbr(1024, 512), cbr(512, 512), cbr(512, 512), up2(512, 256))
        self.decoder6_upsampler = nn.Sequential(cbr(512, 256), cbr(256, 256), up2(256, 128))
        self.decoder3_upsampler = nn.Sequential(cbr(256, 128), cbr(128, 128), up2(128, 64))
        self.decoder0_header = nn.Sequential(cbr(128, 64), cbr(64, 64), conv(64, output_dim, 1))

    def forward(self, x):
        vol = lambda z: z.transpose(-1, -2).reshape(-1, self.embed_dim, *self.grid)
        z3, z6, z9, z12 = (vol(z) for z in self.encoder(x))
        y = self.decoder9_upsampler(torch.cat([self.decoder9(z9), self.decoder12_upsampler(z12)], 1))
        y = self.decoder6_upsampler(torch.cat([self.decoder6(z6), y], 1))
        y = self.decoder3_upsampler(torch.cat([self.decoder3(z3), y], 1))
        return self.decoder0_header(torch.cat([self.decoder0(x), y], 1))
