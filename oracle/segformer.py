"""Oracle (test infrastructure): SegFormer3D, fp32 torch-CPU.

Restates ``/root/reference/models/backbones/segformer_backbone.py`` (MixVisionTransformer: overlapping patch embeddings
:163-197, spatial-reduction attention :51-117, depthwise-conv MLP :13-47 / :346-357, blocks :120-160, stages :200-343)
and ``/root/reference/models/segmentors/segformer_head_official.py:38-90`` with stock torch ops (dropout / stochastic
depth as parameters, initialisation left to the caller).  SURVEY.md 8(f) row N3.  Pinned by
``tests/golden/segformer3d_ref.npz`` (``oracle/gen_golden.py`` runs the reference's own classes on the same
deterministic weights; ``tests/test_oracle_golden.py``).  State-dict keys equal the reference's.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class DWConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv3d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, x, grid):
        B, N, C = x.shape
        return self.dwconv(x.transpose(1, 2).reshape(B, C, *grid)).flatten(2).transpose(1, 2)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.dwconv, self.fc2 = nn.Linear(dim, hidden), DWConv(hidden), nn.Linear(hidden, dim)

    def forward(self, x, grid):
        return self.fc2(F.gelu(self.dwconv(self.fc1(x), grid)))


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias, sr_ratio):
        super().__init__()
        self.num_heads, self.sr_ratio, self.scale = num_heads, sr_ratio, (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        if sr_ratio > 1:
            self.sr = nn.Conv3d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)

    def forward(self, x, grid):
        B, N, C = x.shape
        h = self.num_heads
        q = self.q(x).reshape(B, N, h, C // h).permute(0, 2, 1, 3)
        if self.sr_ratio > 1:
            r = self.sr(x.permute(0, 2, 1).reshape(B, C, *grid)).reshape(B, C, -1).permute(0, 2, 1)
            r = self.norm(r)
        else:
            r = x
        kv = self.kv(r).reshape(B, -1, 2, h, C // h).permute(2, 0, 3, 1, 4)
        attn = ((q @ kv[0].transpose(-2, -1)) * self.scale).softmax(dim=-1)
        return self.proj((attn @ kv[1]).transpose(1, 2).reshape(B, N, C))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, sr_ratio):
        super().__init__()
        self.norm1, self.norm2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.attn = Attention(dim, num_heads, qkv_bias, sr_ratio)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, grid):
        x = x + self.attn(self.norm1(x), grid)
        return x + self.mlp(self.norm2(x), grid)


class OverlapPatchEmbed(nn.Module):
    def __init__(self, patch_size, stride, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=patch_size // 2)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x):
        x = self.proj(x)
        grid = tuple(x.shape[2:])
        return self.norm(x.flatten(2).transpose(1, 2)), grid


class MixVisionTransformer(nn.Module):
    def __init__(self, in_chans=1, embed_dim=48, num_heads=(1, 2, 4, 8), mlp_ratios=(4, 4, 4, 4), qkv_bias=False,
                 depths=(3, 4, 6, 3), sr_ratios=(8, 4, 2, 1)):
        super().__init__()
        dims = [embed_dim * 2 ** i for i in range(len(depths))]
        self.patch_embed1 = OverlapPatchEmbed(7, 4, in_chans, dims[0])
        for i in (1, 2, 3):
            self.add_module(f"patch_embed{i + 1}", OverlapPatchEmbed(3, 2, dims[i - 1], dims[i]))
        for i in range(4):
            self.add_module(f"block{i + 1}", nn.ModuleList([Block(dims[i], num_heads[i], mlp_ratios[i], qkv_bias, sr_ratios[i])
                                                            for _ in range(depths[i])]))
            self.add_module(f"norm{i + 1}", nn.LayerNorm(dims[i]))

    def forward(self, inp):
        x = inp[0] if isinstance(inp, (tuple, list)) else inp
        B = x.shape[0]
        outs = []
        for i in range(4):
            x, grid = getattr(self, f"patch_embed{i + 1}")(x)
            if i == 0:
                outs.append(x)
            for blk in getattr(self, f"block{i + 1}"):
                x = blk(x, grid)
            x = getattr(self, f"norm{i + 1}")(x)
            x = x.reshape(B, *grid, -1).permute(0, 4, 1, 2, 3).contiguous()
            outs.append(x)
        return outs


class _MLP(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.proj = nn.Linear(i, o)

    def forward(self, x):
        return self.proj(x.flatten(2).transpose(1, 2))


class _BasicConv3d(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.conv, self.bn = nn.Conv3d(i, o, kernel_size=1, bias=True), nn.BatchNorm3d(o, eps=0.001)

    def forward(self, x):
        return F.gelu(self.bn(self.conv(x)))


class SegFormerHeadOfficial(nn.Module):
    def __init__(self, encoder, in_channels, num_classes, dropout_ratio=0.1, embedding_dim=512):
        super().__init__()
        self.encoder = encoder
        c1, c2, c3, c4 = in_channels
        self.linear_c4, self.linear_c3 = _MLP(c4, embedding_dim), _MLP(c3, embedding_dim)
        self.linear_c2, self.linear_c1 = _MLP(c2, embedding_dim), _MLP(c1, embedding_dim)
        self.linear_fuse = _BasicConv3d(embedding_dim * 4, embedding_dim)
        self.dropout = nn.Dropout3d(dropout_ratio)
        self.linear_pred = nn.Conv3d(embedding_dim, num_classes, kernel_size=1)

    def forward(self, inputs):
        vol = inputs[0] if isinstance(inputs, (tuple, list)) else inputs
        _, c1, c2, c3, c4 = self.encoder(inputs)
        n = c1.shape[0]
        vol_of = lambda m, c: m(c).permute(0, 2, 1).reshape(n, -1, *c.shape[2:])
        up = lambda t: F.interpolate(t, size=c1.shape[2:], mode="trilinear", align_corners=False)
        cat = torch.cat([up(vol_of(self.linear_c4, c4)), up(vol_of(self.linear_c3, c3)), up(vol_of(self.linear_c2, c2)),
                         vol_of(self.linear_c1, c1)], dim=1)
        x = self.linear_pred(self.dropout(self.linear_fuse(cat)))
        return F.interpolate(x, size=vol.shape[2:], mode="trilinear", align_corners=False)


class SegFormerHead(nn.Module):
    """/root/reference/models/segmentors/segformer_head.py:40-121 (the 'SwinSegFormer' head): coarse-to-fine fusion of FIVE
    feature maps, the fused map upsampled to the input size before Dropout3d and the prediction conv"""

    def __init__(self, encoder, in_channels, num_classes, dropout_ratio=0.1, embedding_dim=512):
        super().__init__()
        self.encoder = encoder
        c0, c1, c2, c3, c4 = in_channels
        self.linear_c4, self.linear_c3 = _MLP(c4, embedding_dim), _MLP(c3, embedding_dim)
        self.linear_c2, self.linear_c1 = _MLP(c2, embedding_dim), _MLP(c1, embedding_dim)
        self.linear_c0 = _MLP(c0, embedding_dim)
        for i in range(4):
            self.add_module(f"linear_fuse_{i}", _BasicConv3d(embedding_dim * 2, embedding_dim))
        self.dropout = nn.Dropout3d(dropout_ratio)
        self.linear_pred = nn.Conv3d(embedding_dim, num_classes, kernel_size=1)

    def forward(self, inputs):
        vol = inputs[0] if isinstance(inputs, (tuple, list)) else inputs
        c = self.encoder(inputs)
        n = c[0].shape[0]
        vol_of = lambda m, t: m(t).permute(0, 2, 1).reshape(n, -1, *t.shape[2:])
        up = lambda t, size: F.interpolate(t, size=size, mode="trilinear", align_corners=False)
        x = vol_of(self.linear_c4, c[4])
        for i in (3, 2, 1, 0):
            x = getattr(self, f"linear_fuse_{i}")(torch.cat([up(x, c[i].shape[2:]), vol_of(getattr(self, f"linear_c{i}"), c[i])], 1))
        return self.linear_pred(self.dropout(up(x, vol.shape[2:])))
