"""Oracle (test infrastructure): the MONAI / "official" Swin-UNETR variant, fp32 torch-CPU.

Restates ``/root/reference/models/segmentors/swin_unetr_official.py`` (a vendored copy of MONAI's ``SwinUNETR``):
``SwinUNETR`` :32-295, ``window_partition / window_reverse`` :297-368, ``get_window_size`` :371-388, ``WindowAttention``
:390-496, ``SwinTransformerBlock`` :499-662, ``PatchMerging`` :665-723 (incl. the duplicated sub-grids x2/x5 and x3/x6),
``compute_mask`` :726-763, ``BasicLayer`` :766-863, ``SwinTransformer`` :866-981 (un-affine ``proj_out`` layer norm), with
``PatchEmbed`` of ``models/blocks/patch_embeddings.py:11-84`` and ``MLPBlock`` of ``models/blocks/mlp.py``.  The conv
encoder / decoder blocks are MONAI's (``oracle/blocks.py``).  Parameter names equal the reference's state-dict keys.
Pinned by ``tests/golden/swin_official_*.npz``: the reference's own file run in the build container (MONAI block names
bound to ``oracle/blocks.py``), see ``oracle/gen_golden.py``.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .blocks import UnetOutBlock, UnetrBasicBlock, UnetrUpBlock


def window_partition(x, ws):
    b, d, h, w, c = x.shape
    x = x.view(b, d // ws[0], ws[0], h // ws[1], ws[1], w // ws[2], ws[2], c)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous().view(-1, ws[0] * ws[1] * ws[2], c)


def window_reverse(windows, ws, dims):
    b, d, h, w = dims
    x = windows.view(b, d // ws[0], h // ws[1], w // ws[2], ws[0], ws[1], ws[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).contiguous().view(b, d, h, w, -1)


def get_window_size(x_size, window_size, shift_size=None):
    use_ws = list(window_size)
    use_ss = list(shift_size) if shift_size is not None else None
    for i in range(len(x_size)):
        if x_size[i] <= window_size[i]:
            use_ws[i] = x_size[i]
            if use_ss is not None:
                use_ss[i] = 0
    return tuple(use_ws) if use_ss is None else (tuple(use_ws), tuple(use_ss))


def compute_mask(dims, window_size, shift_size):
    d, h, w = dims
    img_mask = torch.zeros((1, d, h, w, 1))
    cnt = 0
    for ds in (slice(-window_size[0]), slice(-window_size[0], -shift_size[0]), slice(-shift_size[0], None)):
        for hs in (slice(-window_size[1]), slice(-window_size[1], -shift_size[1]), slice(-shift_size[1], None)):
            for wsl in (slice(-window_size[2]), slice(-window_size[2], -shift_size[2]), slice(-shift_size[2], None)):
                img_mask[:, ds, hs, wsl, :] = cnt
                cnt += 1
    mw = window_partition(img_mask, window_size).squeeze(-1)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


class WindowAttention(nn.Module):
    def __init__(self, dim, num_heads, window_size, qkv_bias=True):
        super().__init__()
        self.num_heads, self.scale = num_heads, (dim // num_heads) ** -0.5
        ws = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws[0] - 1) * (2 * ws[1] - 1) * (2 * ws[2] - 1), num_heads))
        coords = torch.stack(torch.meshgrid(torch.arange(ws[0]), torch.arange(ws[1]), torch.arange(ws[2]), indexing="ij"))
        cf = torch.flatten(coords, 1)
        rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws[0] - 1
        rel[:, :, 1] += ws[1] - 1
        rel[:, :, 2] += ws[2] - 1
        rel[:, :, 0] *= (2 * ws[1] - 1) * (2 * ws[2] - 1)
        rel[:, :, 1] *= 2 * ws[2] - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def forward(self, x, mask):
        b, n, c = x.shape
        qkv = self.qkv(x).reshape(b, n, 3, self.num_heads, c // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * self.scale, qkv[1], qkv[2]
        attn = q @ k.transpose(-2, -1)
        # the index is built for the FULL window and sliced [:n, :n] when the window was clamped (faithful quirk)
        bias = self.relative_position_bias_table[self.relative_position_index[:n, :n].reshape(-1)].reshape(n, n, -1)
        attn = attn + bias.permute(2, 0, 1).contiguous().unsqueeze(0)
        if mask is not None:
            nw = mask.shape[0]
            attn = attn.view(b // nw, nw, self.num_heads, n, n) + mask.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, self.num_heads, n, n)
        attn = attn.softmax(-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(b, n, c))


class MLPBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim):
        super().__init__()
        self.linear1 = nn.Linear(hidden_size, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)

    def forward(self, x):
        return self.linear2(F.gelu(self.linear1(x)))


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.window_size, self.shift_size = window_size, shift_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, num_heads, window_size, qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = MLPBlock(dim, int(dim * mlp_ratio))

    def forward(self, x, mask_matrix):
        shortcut = x
        x = self.norm1(x)
        b, d, h, w, c = x.shape
        ws, ss = get_window_size((d, h, w), self.window_size, self.shift_size)
        pd, pb, pr = (ws[0] - d % ws[0]) % ws[0], (ws[1] - h % ws[1]) % ws[1], (ws[2] - w % ws[2]) % ws[2]
        x = F.pad(x, (0, 0, 0, pr, 0, pb, 0, pd))
        _, dp, hp, wp, _ = x.shape
        if any(i > 0 for i in ss):
            sx = torch.roll(x, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
            am = mask_matrix
        else:
            sx, am = x, None
        aw = self.attn(window_partition(sx, ws), am).view(-1, *(ws + (c,)))
        sx = window_reverse(aw, ws, (b, dp, hp, wp))
        x = torch.roll(sx, shifts=ss, dims=(1, 2, 3)) if any(i > 0 for i in ss) else sx
        if pd > 0 or pr > 0 or pb > 0:
            x = x[:, :d, :h, :w, :].contiguous()
        x = shortcut + x
        return x + self.mlp(self.norm2(x))


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(8 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(8 * dim)

    def forward(self, x):
        b, d, h, w, c = x.shape
        if (h % 2 == 1) or (w % 2 == 1) or (d % 2 == 1):
            x = F.pad(x, (0, 0, 0, d % 2, 0, w % 2, 0, h % 2))     # (the reference's swapped pad order)
        x0 = x[:, 0::2, 0::2, 0::2, :]
        x1 = x[:, 1::2, 0::2, 0::2, :]
        x2 = x[:, 0::2, 1::2, 0::2, :]
        x3 = x[:, 0::2, 0::2, 1::2, :]
        x4 = x[:, 1::2, 0::2, 1::2, :]
        x5 = x[:, 0::2, 1::2, 0::2, :]      # == x2 (the duplicated sub-grid of the reference)
        x6 = x[:, 0::2, 0::2, 1::2, :]      # == x3
        x7 = x[:, 1::2, 1::2, 1::2, :]
        return self.reduction(self.norm(torch.cat([x0, x1, x2, x3, x4, x5, x6, x7], -1)))


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.window_size = window_size
        self.shift_size = tuple(i // 2 for i in window_size)
        self.no_shift = tuple(0 for _ in window_size)
        self.blocks = nn.ModuleList([SwinTransformerBlock(dim, num_heads, window_size,
                                                          self.no_shift if i % 2 == 0 else self.shift_size, mlp_ratio, qkv_bias)
                                     for i in range(depth)])
        self.downsample = PatchMerging(dim)

    def forward(self, x):
        b, c, d, h, w = x.shape
        ws, ss = get_window_size((d, h, w), self.window_size, self.shift_size)
        x = x.permute(0, 2, 3, 4, 1)
        dp, hp, wp = (int(np.ceil(v / s)) * s for v, s in zip((d, h, w), ws))
        mask = compute_mask([dp, hp, wp], ws, ss)
        for blk in self.blocks:
            x = blk(x, mask)
        x = self.downsample(x.reshape(b, d, h, w, -1))
        return x.permute(0, 4, 1, 2, 3)


class PatchEmbed(nn.Module):
    def __init__(self, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_size = tuple(patch_size)
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)

    def forward(self, x):
        _, _, d, h, w = x.shape
        p = self.patch_size
        x = F.pad(x, (0, (-w) % p[2], 0, (-h) % p[1], 0, (-d) % p[0]))
        return self.proj(x)


class SwinTransformer(nn.Module):
    def __init__(self, in_chans, embed_dim, window_size, patch_size, depths, num_heads, mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim)
        for i in range(4):
            layer = BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], tuple(window_size), mlp_ratio, qkv_bias)
            setattr(self, f"layers{i + 1}", nn.ModuleList([layer]))

    @staticmethod
    def proj_out(x, normalize):
        if not normalize:
            return x
        ch = x.shape[1]
        return F.layer_norm(x.permute(0, 2, 3, 4, 1), [ch]).permute(0, 4, 1, 2, 3)

    def forward(self, x, normalize=True):
        x0 = self.patch_embed(x)
        outs = [self.proj_out(x0, normalize)]
        cur = x0
        for i in range(4):
            cur = getattr(self, f"layers{i + 1}")[0](cur.contiguous())
            outs.append(self.proj_out(cur, normalize))
        return outs


class SwinUNETR(nn.Module):
    def __init__(self, img_size, in_channels, out_channels, depths: Sequence[int] = (2, 2, 2, 2),
                 num_heads: Sequence[int] = (3, 6, 12, 24), feature_size: int = 24, normalize: bool = True,
                 window_size: int = 7):
        super().__init__()
        fs = feature_size
        self.normalize = normalize
        self.swinViT = SwinTransformer(in_channels, fs, (window_size,) * 3, (2, 2, 2), depths, num_heads)
        self.encoder1 = UnetrBasicBlock(in_channels, fs)
        self.encoder2 = UnetrBasicBlock(fs, fs)
        self.encoder3 = UnetrBasicBlock(2 * fs, 2 * fs)
        self.encoder4 = UnetrBasicBlock(4 * fs, 4 * fs)
        self.encoder10 = UnetrBasicBlock(16 * fs, 16 * fs)
        self.decoder5 = UnetrUpBlock(16 * fs, 8 * fs)
        self.decoder4 = UnetrUpBlock(8 * fs, 4 * fs)
        self.decoder3 = UnetrUpBlock(4 * fs, 2 * fs)
        self.decoder2 = UnetrUpBlock(2 * fs, fs)
        self.decoder1 = UnetrUpBlock(fs, fs)
        self.out = UnetOutBlock(fs, out_channels)

    def forward(self, x_in):
        if isinstance(x_in, (tuple, list)):
            x_in = x_in[0]
        hs = self.swinViT(x_in, self.normalize)
        enc0 = self.encoder1(x_in)
        enc1 = self.encoder2(hs[0])
        enc2 = self.encoder3(hs[1])
        enc3 = self.encoder4(hs[2])
        dec4 = self.encoder10(hs[4])
        dec3 = self.decoder5(dec4, hs[3])
        dec2 = self.decoder4(dec3, enc3)
        dec1 = self.decoder3(dec2, enc2)
        dec0 = self.decoder2(dec1, enc1)
        out = self.decoder1(dec0, enc0)
        return self.out(out)
