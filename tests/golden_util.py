"""Deterministic parameter/input fills shared by ``oracle/gen_golden.py`` (which runs the
reference in the build container) and the tests (which run the oracle / HIP path).

Weights are a pure function of (state-dict key, shape), so fixtures only need to store
outputs -- the same fill is applied to the reference module when the golden vector is
generated and to the oracle / product module when it is checked.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rng(tag: str) -> np.random.Generator:
    return np.random.default_rng(zlib.crc32(tag.encode()) & 0xFFFFFFFF)


def det_tensor(tag: str, shape, scale: float = 1.0, shift: float = 0.0) -> torch.Tensor:
    a = _rng(tag).standard_normal(size=tuple(shape)).astype(np.float32) * scale + shift
    return torch.from_numpy(a)


def det_fill_(module: torch.nn.Module, salt: str = "") -> None:
    """In-place deterministic fill of every float parameter (buffers untouched).
    1-D 'weight' of norms ~ 1 + 0.1 n, biases ~ 0.1 n, matrices ~ n / sqrt(fan_in),
    bias tables ~ 0.5 n."""
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            tag = salt + name
            if name.endswith("relative_position_bias_table"):
                v = det_tensor(tag, p.shape, 0.5)
            elif p.dim() == 1 and name.endswith("weight"):
                v = det_tensor(tag, p.shape, 0.1, 1.0)
            elif p.dim() == 1:
                v = det_tensor(tag, p.shape, 0.1)
            else:
                fan_in = int(np.prod(p.shape[1:]))
                v = det_tensor(tag, p.shape, 1.0 / np.sqrt(fan_in))
            p.copy_(v)


def sw_predictor(model_in):
    """weight-free deterministic predictor for the sliding-window goldens: uses the window, the per-window relative
    centres (incl. the reference's unsqueeze quirk at sw_batch_size == 1) and the affine; 2 output classes"""
    win, centers, aff = model_in
    win = win.float()
    c = centers.reshape(-1, 3)[:win.shape[0]].to(win.device)
    c0 = win[:, 0] * 0.5 + c[:, 0].view(-1, 1, 1, 1)
    c1 = win[:, 0].abs() + (2.0 * c[:, 1] + 3.0 * c[:, 2]).view(-1, 1, 1, 1) + aff.to(win.device).sum() * 0.125
    return torch.stack([c0, c1], dim=1)


SW_CASES = [  # tag, volume, roi, sw_batch, overlap, mode, cval
    ("pad", (1, 1, 20, 30, 28), (24, 24, 24), 2, 0.5, "gaussian", -1.5),       # volume smaller than the roi in one dim
    ("noncubic", (1, 1, 40, 28, 52), (24, 16, 32), 4, 0.25, "constant", 0.0),
    ("sb1", (1, 1, 36, 36, 36), (24, 24, 24), 1, 0.5, "gaussian", 0.0),        # centers.unsqueeze(0) quirk
    ("batch2", (2, 1, 30, 26, 34), (16, 16, 16), 3, 0.5, "gaussian", 0.25),    # two volumes, short last batch
]


class ToyTokenEncoder(torch.nn.Module):
    """stand-in for the ViT in front of the reference's UNETRC decoder (/root/reference/models/segmentors/unetr.py:195-207
    reads `input_dim`, `embed_dim`, `vol_size`, `patch_size` and calls it on the raw volume): a patch embedding and four
    Linear taps, each [B, L, E] -- plain torch, shared by the golden generator, the oracle tests and the GPU tests"""

    def __init__(self, in_chans=1, embed_dim=48, vol_size=(32, 32, 32), patch_size=(16, 16, 16)):
        super().__init__()
        self.input_dim, self.embed_dim = in_chans, embed_dim
        self.vol_size, self.patch_size = list(vol_size), list(patch_size)
        self.embed = torch.nn.Conv3d(in_chans, embed_dim, kernel_size=tuple(patch_size), stride=tuple(patch_size))
        self.taps = torch.nn.ModuleList([torch.nn.Linear(embed_dim, embed_dim) for _ in range(4)])

    def forward(self, x):
        t = self.embed(x).flatten(2).transpose(1, 2)
        return [torch.tanh(f(t)) for f in self.taps]


UNETRC_PROBES = ["decoder0.0.block.0.block.weight", "decoder0.0.block.1.weight", "decoder0.0.block.1.bias",
                 "decoder0.1.block.0.block.weight", "decoder3.2.block.1.block.weight", "decoder3.2.block.2.weight",
                 "decoder6.0.block.0.block.weight", "decoder9.block.0.block.bias", "decoder12_upsampler.block.weight",
                 "decoder9_upsampler.0.block.0.block.weight", "decoder9_upsampler.3.block.weight",
                 "decoder6_upsampler.1.block.1.bias", "decoder3_upsampler.2.block.bias",
                 "decoder0_header.0.block.0.block.weight", "decoder0_header.2.block.weight", "decoder0_header.2.block.bias",
                 "encoder.embed.weight", "encoder.taps.3.weight"]


def probe(t, n=4096):
    """the first n elements of a (large) gradient: what the UNETRC fixtures keep of it"""
    return t.detach().reshape(-1)[:n].clone()


# the SegFormer3D fixture (tests/golden/segformer3d_ref.npz): MixVisionTransformer + SegFormerHeadOfficial, qkv_bias on
SEGFORMER_CFG = dict(vol=(64, 64, 64), embed_dim=32, depths=[2, 1, 1, 1], num_heads=[1, 2, 4, 8], classes=3, embedding_dim=64)

# the SwinSegFormer fixture (tests/golden/swin_segformer_ref.npz): SwinTransformerNNFormer + SegFormerHead
# (/root/reference/models/model_builder.py:173-189), five feature maps 16^3 ... 1^3 at 32^3 / patch 2
SWIN_SEGFORMER_CFG = dict(vol=(32, 32, 32), embed_dim=16, depths=[2, 1, 1, 1], num_heads=[1, 2, 4, 8], window_size=[4, 4, 4, 2],
                          classes=3, embedding_dim=32,
                          probes=["encoder.patch_embed.proj.weight", "encoder.layers.0.blocks.1.attn.qkv.weight",
                                  "encoder.layers.3.blocks.0.mlp.fc2.weight", "encoder.norm3.weight", "linear_c4.proj.weight",
                                  "linear_c0.proj.weight", "linear_c2.proj.weight", "linear_fuse_3.conv.weight",
                                  "linear_fuse_3.bn.weight", "linear_fuse_0.conv.weight", "linear_fuse_0.bn.bias",
                                  "linear_fuse_2.bn.weight", "linear_pred.weight", "linear_pred.bias"])
