"""Swin-UNETR on the HIP kernels: ``SwinTransformerNNFormer`` encoder + ``SwinUNETRCustom`` decoder.

Encoder: mirror of ``/root/reference/models/backbones/swin_nnformer.py`` (``SwinTransformerNNFormer`` :478-659,
``BasicLayer`` :315-405, ``SwinTransformerBlock`` :199-289, ``WindowAttention`` :67-196, ``PatchMerging`` :292-312) and
``PatchEmbed3D`` (``models/blocks/patch_embeddings.py:86-133``), default code path only.  Decoder: mirror of
``/root/reference/models/segmentors/swin_unetr.py:20-147`` with MONAI's ``UnetrBasicBlock / UnetrUpBlock /
UnetOutBlock``.  Parameter names equal the reference's state-dict keys.

Everything stays channels-last: a token tensor ``[B, L, C]`` IS the volume ``[B, S, H, W, C]``, so the reference's
flatten/transpose/permute/contiguous copies, pad, roll and window partition/reverse do not exist here -- the
window-attention kernel does them by addressing.  The encoder is a chain of single-kernel autograd ops
(``ops.py``); the decoder (93 % of the FLOPs) is one autograd node over ``layers.py`` with concat buffers written in
place.
"""
from __future__ import annotations

from math import ceil
from typing import Sequence

import torch
import torch.nn as nn

from .. import hip, ops
from ..layers import Conv1, ResBlock, UpBlock
from .unet import LOGIT_LD


def _rel_index(ws: int) -> torch.Tensor:
    r = torch.arange(ws)
    c = torch.stack(torch.meshgrid(r, r, r, indexing="ij")).flatten(1)
    rel = c[:, :, None] - c[:, None, :] + (ws - 1)
    m = 2 * ws - 1
    return rel[0] * m * m + rel[1] * m + rel[2]


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _DepthMlp(nn.Module):
    """MLP of SwinDepth (/root/reference/models/backbones/swindepth.py:25-73): fc1 -> GELU -> 3 x (depthwise Conv3d k3
    -> BatchNorm3d(eps 1e-3) -> GELU) on the token volume -> fc2.  Tokens are channels-last volumes here, so the
    reference's permute / reshape pairs around the convolutions do not exist."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        for i in (1, 2, 3):
            self.add_module(f"dwc{i}", nn.Conv3d(hidden, hidden, kernel_size=3, padding=1, groups=hidden))
        for i in (1, 2, 3):
            self.add_module(f"bn{i}", nn.BatchNorm3d(hidden, eps=0.001))
        self.fc2 = nn.Linear(hidden, dim)
        self.sync_group = None      # set to a process group (or True) for SyncBatchNorm under data parallelism

    def run(self, y):
        y = ops.gelu(ops.linear(y, self.fc1.weight, self.fc1.bias))
        for i in (1, 2, 3):
            conv, bn = getattr(self, f"dwc{i}"), getattr(self, f"bn{i}")
            y = ops.gelu(ops.batch_norm(ops.dwconv3(y, conv.weight, conv.bias), bn, self.sync_group))
        return ops.linear(y, self.fc2.weight, self.fc2.bias)


def _p8(n: int) -> int:
    return (n + 7) // 8 * 8


class _PaddedState:
    """Mixin for modules whose compute-side parameters are ZERO-PADDED versions of the reference's (channel counts raised
    to multiples of 8 = one 16-byte bf16 chunk, so every kernel keeps its vector paths).  `_real` maps a local parameter /
    buffer name to the reference shape; the state dict shows (and accepts) the reference shapes.

    The padding entries stay exactly zero through training: a padded output channel has zero weights and bias, a zero
    BatchNorm scale and shift, so its activation is 0 and every gradient that reaches a padding entry is a sum of
    products with those zeros; AdamW maps (p, m, v) = (0, 0, 0) to 0."""

    def _init_padded(self):
        self._register_state_dict_hook(_PaddedState._slice_hook)
        self._register_load_state_dict_pre_hook(self._pad_hook)
        self.zero_padding_()

    def _cut(self, name, t):
        return t[tuple(slice(0, n) for n in self._real[name])]

    @torch.no_grad()
    def zero_padding_(self):
        own = dict(self.named_parameters())
        own.update(dict(self.named_buffers()))
        for name in self._real:
            t = own[name]
            keep = self._cut(name, t).clone()
            t.zero_()
            self._cut(name, t).copy_(keep)

    @staticmethod
    def _slice_hook(module, state, prefix, local_metadata):
        for name in module._real:
            state[prefix + name] = module._cut(name, state[prefix + name]).clone()

    def _pad_hook(self, state, prefix, local_metadata, strict, missing, unexpected, errors):
        own = dict(self.named_parameters())
        own.update(dict(self.named_buffers()))
        for name, real in self._real.items():
            v = state.get(prefix + name)
            if v is not None and tuple(v.shape) == tuple(real) and tuple(v.shape) != tuple(own[name].shape):
                full = torch.zeros(own[name].shape, dtype=v.dtype, device=v.device)
                self._cut(name, full).copy_(v)
                state[prefix + name] = full


class _BasicConv3d(nn.Module, _PaddedState):
    """Conv3d(bias) -> BatchNorm3d(eps 1e-3) -> GELU of the Inception head (/root/reference/models/backbones/swinception.py:45-56),
    channel counts padded to multiples of 8 (see _PaddedState)"""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.k = k
        self.conv = nn.Conv3d(_p8(cin), _p8(cout), kernel_size=k, padding=k // 2, bias=True)
        self.bn = nn.BatchNorm3d(_p8(cout), eps=0.001)
        # the reference's layer draws its default initialisation from the REAL fan-in
        ref = nn.Conv3d(cin, cout, kernel_size=k, padding=k // 2, bias=True)
        with torch.no_grad():
            self.conv.weight.zero_(); self.conv.bias.zero_()
            self.conv.weight[:cout, :cin].copy_(ref.weight); self.conv.bias[:cout].copy_(ref.bias)
        self._real = {"conv.weight": (cout, cin, k, k, k), "conv.bias": (cout,), "bn.weight": (cout,), "bn.bias": (cout,),
                      "bn.running_mean": (cout,), "bn.running_var": (cout,)}
        self._init_padded()

    def run(self, x, group):
        c = self.conv
        y = ops.linear(x, c.weight, c.bias) if self.k == 1 else ops.Conv3Fn.apply(x, c.weight, c.bias, 1)
        return ops.gelu(ops.batch_norm(y, self.bn, group))


class _InceptionBranch(nn.Module):
    NAMES = (("branch1x1",), ("branch3x3_1", "branch3x3_2"), ("branch3x3dbl_1", "branch3x3dbl_2", "branch3x3dbl_3"),
             ("branch3x3trpl_1", "branch3x3trpl_2", "branch3x3trpl_3", "branch3x3trpl_4"), ("branch_pool_2",))

    def __init__(self, kind, cin, cout, bottleneck_divisor=8):
        super().__init__()
        self.kind = kind
        names = self.NAMES[kind]
        bn_dim = cin // bottleneck_divisor
        dims = [cin] + [bn_dim] * (len(names) - 1) + [cout]
        for i, n in enumerate(names):
            self.add_module(n, _BasicConv3d(dims[i], dims[i + 1], 1 if (i == 0 or len(names) == 1) else 3))

    def run(self, x, group):
        if self.kind == 4:
            x = ops.avg_pool3(x)
        for n in self.NAMES[self.kind]:
            x = getattr(self, n).run(x, group)
        return x


class _InceptionFc(nn.Linear, _PaddedState):
    """Linear over the concatenated branches; the compute-side weight has one zero-padded slot of p8(branch) columns per
    branch: reference [dim, 5 * branch] <-> padded [dim, 5 * p8(branch)]"""

    def __init__(self, branch, nb, dim):
        nn.Linear.__init__(self, nb * _p8(branch), dim)
        self.branch, self.nb = branch, nb
        ref = nn.Linear(nb * branch, dim)
        with torch.no_grad():
            self.weight.zero_()
            self._slots(self.weight).copy_(ref.weight.view(dim, nb, branch))
            self.bias.copy_(ref.bias)
        self._register_state_dict_hook(_InceptionFc._fc_slice)
        self._register_load_state_dict_pre_hook(self._fc_pad)

    def _slots(self, w):
        return w.view(w.shape[0], self.nb, -1)[:, :, :self.branch]

    @staticmethod
    def _fc_slice(module, state, prefix, local_metadata):
        w = state[prefix + "weight"]
        state[prefix + "weight"] = module._slots(w).reshape(w.shape[0], -1).clone()

    def _fc_pad(self, state, prefix, local_metadata, strict, missing, unexpected, errors):
        v = state.get(prefix + "weight")
        if v is not None and v.shape[1] == self.nb * self.branch and v.shape[1] != self.weight.shape[1]:
            full = torch.zeros(self.weight.shape, dtype=v.dtype, device=v.device)
            self._slots(full).copy_(v.view(v.shape[0], self.nb, self.branch))
            state[prefix + "weight"] = full


class _InceptionMlp(nn.Module):
    """`InceptionHead` of SwInception (/root/reference/models/backbones/swinception.py:127-174): the MLP of a Swin block
    replaced by five convolutional branches on the token volume -- 1x1x1; 1x1x1 -> 3x3x3; -> 3x3x3 twice; -> three times;
    3x3x3 average pool -> 1x1x1, each conv followed by BatchNorm3d + GELU, int(hidden / 5) output channels per branch and
    dim // 8 bottleneck channels -- concatenated and projected back by one Linear.  Tokens are channels-last volumes, so
    the reference's permute / reshape pairs do not exist; the odd channel counts (38 / 6 at dim 48, 153 / 24 at 192 ...)
    run zero-padded to multiples of 8 (`_PaddedState`), the concat is written slot by slot by the copy form of the
    interpolation kernel."""

    def __init__(self, dim, hidden, res):
        super().__init__()
        self.res = tuple(res)
        self.branch = int(hidden * 0.2)
        self.branches = nn.ModuleList([_InceptionBranch(b, dim, self.branch) for b in range(5)])
        self.fc = _InceptionFc(self.branch, 5, dim)
        self.sync_group = None

    def run(self, y):
        outs = [b.run(y, self.sync_group) for b in self.branches]
        cat = ops.upsample_concat(tuple(y.shape[1:4]), outs)
        return ops.linear(cat, self.fc.weight, self.fc.bias)


class _WindowAttention(nn.Module):
    def __init__(self, dim, ws, heads, qkv_bias):
        super().__init__()
        self.ws, self.heads = ws, heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 3, heads))
        self.register_buffer("relative_position_index", _rel_index(ws))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)


class _Block(nn.Module):
    def __init__(self, dim, res, heads, ws, shift, mlp_ratio, qkv_bias, drop_path, mlp="plain"):
        super().__init__()
        self.res = tuple(res)
        if min(self.res) <= ws:          # swin_nnformer.py:213-216
            shift, ws = 0, min(self.res)
        self.ws, self.shift, self.heads, self.drop_path = ws, shift, heads, float(drop_path)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _WindowAttention(dim, ws, heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        if mlp == "depth":
            self.mlp = _DepthMlp(dim, int(dim * mlp_ratio))
        elif mlp == "inception":
            self.mlp = _InceptionMlp(dim, int(dim * mlp_ratio), self.res)
        else:
            self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def _dp_scale(self, x):
        """per-sample stochastic-depth factor mask[b] / keep (models/layers/drop_path.py:15-45), or None; the multiply
        happens inside the residual-add kernel.  `self.dp_mask` (fp32 [B] of 0 / 1) overrides the draw (tests)."""
        if self.drop_path == 0.0 or not self.training:
            return None
        keep = 1.0 - self.drop_path
        mask = getattr(self, "dp_mask", None)
        if mask is None:
            mask = torch.empty(x.shape[0], device=x.device, dtype=torch.float32).bernoulli_(keep)
        return mask.to(device=x.device, dtype=torch.float32) / keep

    def forward(self, x):
        a = self.attn
        x, xn = ops.layer_norm_res(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)   # x: the residual, through the node
        qkv = ops.linear(xn, a.qkv.weight, a.qkv.bias)
        y = ops.WindowAttnFn.apply(qkv, a.qkv.bias, a.relative_position_bias_table, self.heads, self.ws, self.shift)
        dp = self._dp_scale(x)
        if dp is None:     # no stochastic depth in this step: the residual adds ride on the Linear kernels' epilogues
            x = ops.linear_add(y, a.proj.weight, a.proj.bias, x)
        else:
            x = ops.add(x, ops.linear(y, a.proj.weight, a.proj.bias), dp)
        x, y = ops.layer_norm_res(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        dp = self._dp_scale(x)
        if isinstance(self.mlp, (_DepthMlp, _InceptionMlp)):
            return ops.add(x, self.mlp.run(y), dp)
        if dp is None:
            return ops.mlp(y, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, res=x)
        return ops.add(x, ops.mlp(y, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias), dp)


class _PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Conv3d(dim, dim * 2, kernel_size=3, stride=2, padding=1)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        g = ops.layer_norm(ops.gelu(x), self.norm.weight, self.norm.bias, self.norm.eps)
        return ops.Conv3Fn.apply(g, self.reduction.weight, self.reduction.bias, 2)


class _BasicLayer(nn.Module):
    def __init__(self, dim, res, depth, heads, ws, mlp_ratio, qkv_bias, drop_path, mlp="plain"):
        super().__init__()
        self.blocks = nn.ModuleList([_Block(dim, res, heads, ws, 0 if i % 2 == 0 else ws // 2, mlp_ratio, qkv_bias,
                                            drop_path[i], mlp) for i in range(depth)])
        self.downsample = _PatchMerging(dim)

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return self.downsample(x)


class _PatchEmbed3D(nn.Module):
    def __init__(self, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_size = tuple(patch_size)
        if len(set(self.patch_size)) != 1:
            raise NotImplementedError("anisotropic patch sizes are not implemented")
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x_cl):
        y = ops.PatchConvFn.apply(x_cl, self.proj.weight, self.proj.bias, self.patch_size[0])
        return ops.layer_norm(y, self.norm.weight, self.norm.bias, self.norm.eps)


class SwinTransformerNNFormer(nn.Module):
    """returns the 5 feature volumes channels-last: [C@R, 2C@R/2, 4C@R/4, 8C@R/8, 16C@R/16] (R = vol / patch)."""

    def __init__(self, pretrain_img_size=(96, 96, 96), patch_size=(2, 2, 2), in_chans=1, embed_dim=48,
                 depths: Sequence[int] = (2, 2, 2, 2), num_heads: Sequence[int] = (3, 6, 12, 24),
                 window_size: Sequence[int] = (6, 6, 6, 3), mlp_ratio=4.0, qkv_bias=True, drop_path_rate=0.2,
                 compute_dtype=torch.bfloat16, mlp="plain"):
        super().__init__()
        self.num_layers, self.embed_dim, self.compute_dtype = len(depths), embed_dim, compute_dtype
        self.patch_embed = _PatchEmbed3D(patch_size, in_chans, embed_dim)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            res = tuple(pretrain_img_size[d] // patch_size[d] // 2 ** i for d in range(3))
            self.layers.append(_BasicLayer(embed_dim * 2 ** i, res, depths[i], num_heads[i], window_size[i], mlp_ratio,
                                           qkv_bias, dpr[sum(depths[:i]):sum(depths[:i + 1])], mlp))
        self.num_features = [embed_dim * 2 ** (i + 1) for i in range(self.num_layers)]
        for i in range(self.num_layers):
            self.add_module(f"norm{i}", nn.LayerNorm(self.num_features[i]))

    def forward(self, inp):
        vol = inp[0] if isinstance(inp, (tuple, list)) else inp
        if not vol.is_cuda:
            raise RuntimeError("SwinTransformerNNFormer runs on the GPU only (no CPU fallback)")
        B, Cin, D, H, W = vol.shape
        p = self.patch_embed.patch_size[0]
        if D % p or H % p or W % p:
            raise ValueError("volume must be a multiple of the patch size")
        x_cl = torch.empty(B, D, H, W, Cin, dtype=self.compute_dtype, device=vol.device)
        hip.to_channels_last(vol if vol.dtype in (torch.float32, torch.bfloat16) else vol.float(), x_cl)
        x = self.patch_embed(x_cl)
        feats = [x]
        for i, layer in enumerate(self.layers):
            x = layer(x)
            n = getattr(self, f"norm{i}")
            feats.append(ops.layer_norm(x, n.weight, n.bias, n.eps))   # norm of the DOWNSAMPLED tensor (:653-658)
        return feats, x_cl


class SwinDepth(SwinTransformerNNFormer):
    """/root/reference/models/backbones/swindepth.py:400-691 with its default-off extras off (learned class vectors,
    affine / crop position terms, global token): the reference's Swin encoder whose MLP carries three depthwise
    Conv3d + BatchNorm3d + GELU stages (`_DepthMlp`).  `sync_batchnorm(group)` = what run_training.py:83 does under DDP."""

    def __init__(self, *a, **k):
        k["mlp"] = "depth"
        super().__init__(*a, **k)

    def sync_batchnorm(self, group=True):
        for m in self.modules():
            if isinstance(m, (_DepthMlp, _InceptionMlp)):
                m.sync_group = group
        return self


class SwInception(SwinDepth):
    """/root/reference/models/backbones/swinception.py:609-791 with its default-off extras off (learned class vectors,
    affine / crop position terms, global token): the reference's Swin encoder whose MLP is the Inception head
    (`_InceptionMlp`); everything else is SwinTransformerNNFormer.  `sync_batchnorm(group)` as for SwinDepth."""

    def __init__(self, *a, **k):
        k["mlp"] = "inception"
        SwinTransformerNNFormer.__init__(self, *a, **k)


# ------------------------------------------------------------------------------------------------------------
# decoder
# ------------------------------------------------------------------------------------------------------------
class _ConvOnly(nn.Sequential):
    def __init__(self, cin, cout, k, transposed=False, bias=False):
        super().__init__()
        if transposed:
            self.add_module("conv", nn.ConvTranspose3d(cin, cout, kernel_size=k, stride=k, bias=bias))
        else:
            self.add_module("conv", nn.Conv3d(cin, cout, kernel_size=k, padding=(k - 1) // 2, bias=bias))


class _UnetResBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = _ConvOnly(cin, cout, 3)
        self.conv2 = _ConvOnly(cout, cout, 3)
        if cin != cout:
            self.conv3 = _ConvOnly(cin, cout, 1)

    def op(self):
        c3 = self.conv3.conv.weight if hasattr(self, "conv3") else None
        return ResBlock(self.conv1.conv.weight, self.conv2.conv.weight, c3)


class _UnetrBasicBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.layer = _UnetResBlock(cin, cout)


class _UnetrUpBlock(nn.Module):
    def __init__(self, cin, cout, up_k):
        super().__init__()
        if up_k != 2:
            raise NotImplementedError("only 2x transposed-conv upsampling (patch_size 2) is implemented")
        self.transp_conv = _ConvOnly(cin, cout, up_k, transposed=True)
        self.conv_block = _UnetResBlock(cout + cout, cout)

    def op(self):
        b = self.conv_block
        return UpBlock(self.transp_conv.conv.weight, b.conv1.conv.weight, b.conv2.conv.weight, b.conv3.conv.weight)


class _UnetOutBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _ConvOnly(cin, cout, 1, bias=True)


class SwinUNETRCustom(nn.Module):
    def __init__(self, encoder, in_channels, out_channels, img_size=(96, 96, 96), hidden_size=48, patch_size=(2, 2, 2),
                 compute_dtype=torch.bfloat16):
        super().__init__()
        self.encoder, self.out_channels, self.compute_dtype = encoder, out_channels, compute_dtype
        hs = hidden_size
        ps = patch_size[0] if isinstance(patch_size, (tuple, list)) else patch_size
        self.unet_encoders = nn.ModuleList([_UnetrBasicBlock(in_channels, hs), _UnetrBasicBlock(hs, hs)])
        self.unet_decoders = nn.ModuleList([_UnetrUpBlock(hs, hs, ps)])
        for i in range(encoder.num_layers):
            self.unet_encoders.append(_UnetrBasicBlock(hs * 2 ** (i + 1), hs * 2 ** (i + 1)))
            self.unet_decoders.append(_UnetrUpBlock(hs * 2 ** (i + 1), hs * 2 ** i, 2))
        self.out = _UnetOutBlock(hs, out_channels)
        self._build_ops()

    def _build_ops(self):
        self._enc_ops = [m.layer.op() for m in self.unet_encoders]
        self._dec_ops = [m.op() for m in self.unet_decoders]
        self._out_op = Conv1(self.out.conv.conv.weight, self.out.conv.conv.bias)

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._build_ops()
        return r

    # ---- two-phase backward (data-parallel overlap), same protocol as models/unet.py ----
    # With `defer_backward_tail(True)` the autograd backward stops after the conv decoder (80 % of the 77 M parameters of
    # Swin-UNETR-48: a suffix of the flat gradient buffer) and `backward_tail()` continues into the Swin encoder; the
    # caller starts the all-reduce of the finished gradients in between (parallel.GradSync), which is what DDP's
    # bucketed overlap does for the reference (/root/reference/run_training.py:82-85).
    def defer_backward_tail(self, on: bool = True):
        self._defer_tail = bool(on)
        self._pending_tail = None
        return self

    def tail_parameters(self):
        """parameters whose gradients `backward_tail()` produces"""
        return list(self.encoder.parameters())

    def backward_tail(self):
        pend, self._pending_tail = getattr(self, "_pending_tail", None), None
        if pend is not None:
            feats, grads = pend
            keep = [(f, g) for f, g in zip(feats, grads) if g is not None and f.requires_grad]
            if keep:
                torch.autograd.backward([f for f, _ in keep], [g for _, g in keep])

    def forward(self, x_in):
        if not isinstance(x_in, (tuple, list)):
            x_in = (x_in, None, None)
        feats, x_cl = self.encoder(x_in)
        dec_params = [p for m in (self.unet_encoders, self.unet_decoders, self.out) for p in m.parameters()]
        if getattr(self, "_defer_tail", False) and torch.is_grad_enabled():
            # two-phase backward: the decoder sees detached copies of the feature maps, so the first backward stops there
            # (autograd would otherwise run the encoder's nodes with zero-filled gradients); backward_tail() feeds the
            # gradients the decoder produced into the encoder's graph
            self._tail_feats = feats
            feats = [f.detach().requires_grad_(f.requires_grad) for f in feats]
        return _DecoderFn.apply(self, x_cl, len(feats), *feats, *dec_params)


class _DecoderFn(torch.autograd.Function):
    """x = dec[-1](enc[-1](z[-1]), enc[-2](z[-2])); ... ; x = dec[0](x, enc[0](x_in)); out(x)
    (/root/reference/models/segmentors/swin_unetr.py:138-147)"""

    @staticmethod
    def forward(ctx, net: SwinUNETRCustom, x_cl, nf, *rest):
        feats = [f.contiguous() for f in rest[:nf]]
        E, Dd = net._enc_ops, net._dec_ops
        L = len(Dd)                      # number of up blocks = num_layers + 1
        saved = {"enc": [None] * (L + 1), "dec": [None] * L}
        # top of the pyramid
        top, saved["enc"][L] = E[L].fwd(feats[L - 1])
        x = top
        for k in range(L - 1, -1, -1):   # decoder k consumes skip enc[k](source k)
            src = feats[k - 1] if k >= 1 else x_cl
            cat = Dd[k].alloc_cat(x)
            cout = Dd[k].cout
            _, saved["enc"][k] = E[k].fwd(src, out=cat[..., cout:])
            x, saved["dec"][k] = Dd[k].fwd(x, cat)
        N, D, H, W, _ = x.shape
        logits_cl = torch.empty(N, D, H, W, LOGIT_LD, dtype=x.dtype, device=x.device)
        net._out_op.fwd(x, logits_cl[..., :net.out_channels])
        if any(ctx.needs_input_grad):
            ctx.net, ctx.saved, ctx.last, ctx.nf, ctx.n_in = net, saved, x, nf, 3 + len(rest)
            ctx.feat_needs = [ctx.needs_input_grad[3 + i] for i in range(nf)]
        ctx.set_materialize_grads(False)
        # [B, C, D, H, W] view of the channels-last logits rows (see models/unet.py): no NCDHW round trip to the loss
        return logits_cl[..., :net.out_channels].permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, dlogits):
        net, saved, nf = ctx.net, ctx.saved, ctx.nf
        if dlogits is None:
            return (None,) * ctx.n_in
        E, Dd = net._enc_ops, net._dec_ops
        L = len(Dd)
        N, C, D, H, W = dlogits.shape
        from ..losses import channels_last_grad
        dl = channels_last_grad(dlogits, LOGIT_LD, net.compute_dtype)
        if dl is None:
            dl = torch.zeros(N, D, H, W, LOGIT_LD, dtype=net.compute_dtype, device=dlogits.device)
            hip.to_channels_last(dlogits.contiguous(), dl[..., :C])
        g = net._out_op.bwd(ctx.last, dl, True, dy_channels=LOGIT_LD)
        dfeats = [None] * nf
        for k in range(0, L):
            g, dskip = Dd[k].bwd(saved["dec"][k], g)
            need = k >= 1 and ctx.feat_needs[k - 1]
            d = E[k].bwd(saved["enc"][k], dskip, need_dx=need)
            if k >= 1:
                dfeats[k - 1] = d
        dfeats[L - 1] = E[L].bwd(saved["enc"][L], g, need_dx=ctx.feat_needs[L - 1])
        ctx.saved = None
        tail_feats = getattr(net, "_tail_feats", None)
        if tail_feats is not None and getattr(net, "_defer_tail", False):
            # two-phase backward: the encoder's part of the graph runs in net.backward_tail()
            net._pending_tail, net._tail_feats = (tail_feats, dfeats), None
            dfeats = [None] * nf
        return (None, None, None, *dfeats) + (None,) * (ctx.n_in - 3 - nf)
