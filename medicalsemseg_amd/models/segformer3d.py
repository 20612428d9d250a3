"""SegFormer3D: ``MixVisionTransformer`` encoder + ``SegFormerHeadOfficial``, the
``cfg.model == 'SegFormer3D'`` branch of ``/root/reference/models/model_builder.py:190-205``.

Reference: ``/root/reference/models/backbones/segformer_backbone.py`` (Mlp with depthwise conv :13-47 / :346-357,
spatial-reduction Attention :51-117, Block :120-160, OverlapPatchEmbed :163-197, MixVisionTransformer :200-343) and
``/root/reference/models/segmentors/segformer_head_official.py:38-90``.

MI355X design: tokens stay channels-last volumes ``[B, d0, d1, d2, C]`` end to end, so every ``flatten / transpose /
reshape / permute`` pair of the reference disappears; each op is one HIP kernel behind ``ops.py``:
overlapping patch embeddings = gather conv (k7 s4 p3) / conv k3 s2, attention = ``msseg_kv_attention`` (the reduced key
set -- 27 keys at 96^3 -- lives in LDS, no score tensor), spatial reduction = gather conv k = s, MLP = Linear ->
``msseg_dwconv3d_k3`` -> GELU -> Linear, head = four Linears, ``msseg_interp_trilinear`` writing straight into the
2048-channel concat, Linear + BatchNorm + GELU, channel dropout, 1x1x1 prediction conv and the final x4 trilinear
upsampling on 8-channel logits rows.  State-dict keys equal the reference's.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import hip, ops
from ..layers import Conv1

LOGIT_LD = 8


class _DWConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv3d(dim, dim, 3, 1, 1, bias=True, groups=dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.dwconv = _DWConv(hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        x = ops.linear(x, self.fc1.weight, self.fc1.bias)
        x = ops.gelu(ops.dwconv3(x, self.dwconv.dwconv.weight, self.dwconv.dwconv.bias))
        return ops.linear(x, self.fc2.weight, self.fc2.bias)


class _Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias, sr_ratio):
        super().__init__()
        if dim % num_heads or (dim // num_heads) not in (16, 32, 48, 64):
            raise ValueError(f"head_dim {dim}/{num_heads} must be one of 16, 32, 48, 64")
        self.num_heads, self.sr_ratio = num_heads, sr_ratio
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        if sr_ratio > 1:
            self.sr = nn.Conv3d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        B, d0, d1, d2, C = x.shape
        q = ops.linear(x, self.q.weight, self.q.bias)
        if self.sr_ratio > 1:
            if d0 % self.sr_ratio or d1 % self.sr_ratio or d2 % self.sr_ratio:
                raise ValueError("token grid must be a multiple of the spatial-reduction ratio")
            r = ops.PatchConvFn.apply(x, self.sr.weight, self.sr.bias, self.sr_ratio)
            r = ops.layer_norm(r, self.norm.weight, self.norm.bias, self.norm.eps)
        else:
            r = x
        kv = ops.linear(r, self.kv.weight, self.kv.bias)
        o = ops.kv_attention(q.reshape(B, -1, C), kv.reshape(B, -1, 2 * C), self.num_heads).reshape(x.shape)
        return ops.linear(o, self.proj.weight, self.proj.bias)


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, drop_path, sr_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _Attention(dim, num_heads, qkv_bias, sr_ratio)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.drop_path = float(drop_path)

    def _dp(self, x):
        if self.drop_path == 0.0 or not self.training:
            return None
        keep = 1.0 - self.drop_path
        return torch.empty(x.shape[0], device=x.device, dtype=torch.float32).bernoulli_(keep) / keep

    def forward(self, x):
        y = self.attn(ops.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps))
        x = ops.add(x, y, self._dp(x))
        y = self.mlp(ops.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps))
        return ops.add(x, y, self._dp(x))


class _OverlapPatchEmbed(nn.Module):
    def __init__(self, patch_size, stride, in_chans, embed_dim):
        super().__init__()
        self.k, self.s = patch_size, stride
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=patch_size // 2)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x):
        if self.k == 3 and self.s == 2 and self.proj.in_channels % 8 == 0:
            y = ops.Conv3Fn.apply(x, self.proj.weight, self.proj.bias, 2)
        else:
            y = ops.PatchConvFn.apply(x, self.proj.weight, self.proj.bias, self.k, self.s, self.k // 2)
        return ops.layer_norm(y, self.norm.weight, self.norm.bias, self.norm.eps)


class MixVisionTransformer(nn.Module):
    """returns [tokens after patch_embed1, c1, c2, c3, c4] as channels-last volumes (the reference returns the first as
    [B, L, C] tokens and the rest NCDHW)"""

    def __init__(self, img_size=96, patch_size=16, in_chans=1, embed_dim=48, num_heads=(1, 2, 4, 8), mlp_ratios=(4, 4, 4, 4),
                 qkv_bias=False, drop_path_rate=0.0, depths=(3, 4, 6, 3), sr_ratios=(8, 4, 2, 1), compute_dtype=torch.bfloat16):
        super().__init__()
        self.depths, self.compute_dtype = tuple(depths), compute_dtype
        dims = [embed_dim * 2 ** i for i in range(len(depths))]
        self.embed_dims = dims
        self.patch_embed1 = _OverlapPatchEmbed(7, 4, in_chans, dims[0])
        self.patch_embed2 = _OverlapPatchEmbed(3, 2, dims[0], dims[1])
        self.patch_embed3 = _OverlapPatchEmbed(3, 2, dims[1], dims[2])
        self.patch_embed4 = _OverlapPatchEmbed(3, 2, dims[2], dims[3])
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        cur = 0
        for i in range(4):
            blocks = nn.ModuleList([_Block(dims[i], num_heads[i], mlp_ratios[i], qkv_bias, dpr[cur + j], sr_ratios[i])
                                    for j in range(depths[i])])
            self.add_module(f"block{i + 1}", blocks)
            self.add_module(f"norm{i + 1}", nn.LayerNorm(dims[i]))
            cur += depths[i]

    def forward(self, inp):
        vol = inp[0] if isinstance(inp, (tuple, list)) else inp
        if not vol.is_cuda:
            raise RuntimeError("MixVisionTransformer runs on the GPU only (no CPU fallback)")
        B, Cin, D, H, W = vol.shape
        x = torch.empty(B, D, H, W, Cin, dtype=self.compute_dtype, device=vol.device)
        hip.to_channels_last(vol if vol.dtype in (torch.float32, torch.bfloat16) else vol.float(), x)
        outs = []
        for i in range(4):
            x = getattr(self, f"patch_embed{i + 1}")(x)
            if i == 0:
                outs.append(x)
            for blk in getattr(self, f"block{i + 1}"):
                x = blk(x)
            n = getattr(self, f"norm{i + 1}")
            x = ops.layer_norm(x, n.weight, n.bias, n.eps)
            outs.append(x)
        return outs


class _MLP(nn.Module):
    def __init__(self, input_dim, embed_dim):
        super().__init__()
        self.proj = nn.Linear(input_dim, embed_dim)


class _BasicConv3d(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel_size=1, bias=True)
        self.bn = nn.BatchNorm3d(cout, eps=0.001)


class SegFormerHeadOfficial(nn.Module):
    """``model((vol, crop_loc, affine)) -> logits [B, classes, D, H, W]`` (a view of channels-last rows, as the other
    models return)"""

    graph_safe = False

    def __init__(self, encoder, in_channels, num_classes, dropout_ratio=0.1, embedding_dim=512, compute_dtype=torch.bfloat16):
        super().__init__()
        self.encoder, self.num_classes, self.compute_dtype = encoder, num_classes, compute_dtype
        c1, c2, c3, c4 = in_channels
        self.linear_c4 = _MLP(c4, embedding_dim)
        self.linear_c3 = _MLP(c3, embedding_dim)
        self.linear_c2 = _MLP(c2, embedding_dim)
        self.linear_c1 = _MLP(c1, embedding_dim)
        self.linear_fuse = _BasicConv3d(embedding_dim * 4, embedding_dim)
        self.dropout_ratio = float(dropout_ratio)
        self.linear_pred = nn.Conv3d(embedding_dim, num_classes, kernel_size=1)
        self.sync_group = None          # SyncBatchNorm group under data parallelism (run_training.py:83)
        self.dropout_mask = None        # fp32 [B, embedding_dim] of 0 / 1: overrides the Bernoulli draw (tests)

    def forward(self, inputs):
        vol = inputs[0] if isinstance(inputs, (tuple, list)) else inputs
        size = tuple(vol.shape[2:])
        _, c1, c2, c3, c4 = self.encoder(inputs)
        lin = lambda m, c: ops.linear(c, m.proj.weight, m.proj.bias)
        cat = ops.upsample_concat(c1.shape[1:4], [lin(self.linear_c4, c4), lin(self.linear_c3, c3), lin(self.linear_c2, c2),
                                                  lin(self.linear_c1, c1)])
        f = self.linear_fuse
        x = ops.gelu(ops.batch_norm(ops.linear(cat, f.conv.weight, f.conv.bias), f.bn, self.sync_group))
        x = ops.dropout3d(x, self.dropout_ratio, self.training, self.dropout_mask)
        op = getattr(self, "_pred_op", None)
        if op is None or op.w is not self.linear_pred.weight:
            op = self._pred_op = Conv1(self.linear_pred.weight, self.linear_pred.bias)   # keeps its packed images across steps
        return _PredUpsampleFn.apply(x, self.linear_pred.weight, self.linear_pred.bias, size, self.num_classes, op)


class SegFormerHead(nn.Module):
    """The 'SwinSegFormer' head (``/root/reference/models/segmentors/segformer_head.py:40-121``, wired at
    ``models/model_builder.py:173-189`` behind the SwinTransformerNNFormer encoder): the five feature maps are fused
    coarse to fine -- Linear per map, trilinear upsampling to the next finer map, cat([coarse, fine]) -> Conv3d 1x1x1 +
    BatchNorm3d(eps 1e-3) + GELU -- then Dropout3d and the prediction conv.

    The reference upsamples the fused 512-channel map to the input size BEFORE the dropout and the prediction conv
    (:107-116).  Channel dropout, a 1x1x1 conv and trilinear interpolation commute (the interpolation weights of a voxel
    sum to 1, so the bias passes through; the dropout scales whole channels), so here the dropout and the prediction
    conv run on the c0-resolution map and only the `classes` logits rows are upsampled (``_PredUpsampleFn``): 1/8 of the
    conv FLOPs at patch 2 and no [B, 512, D, H, W] tensor.  Same state-dict keys as the reference."""

    graph_safe = False

    def __init__(self, encoder, in_channels, num_classes, dropout_ratio=0.1, embedding_dim=512, compute_dtype=torch.bfloat16):
        super().__init__()
        if len(in_channels) != 5:
            raise ValueError("SegFormerHead fuses five feature maps (a four-stage encoder + its patch embedding)")
        self.encoder, self.num_classes, self.compute_dtype = encoder, num_classes, compute_dtype
        c0, c1, c2, c3, c4 = in_channels
        self.linear_c4 = _MLP(c4, embedding_dim)
        self.linear_c3 = _MLP(c3, embedding_dim)
        self.linear_c2 = _MLP(c2, embedding_dim)
        self.linear_c1 = _MLP(c1, embedding_dim)
        self.linear_c0 = _MLP(c0, embedding_dim)
        for i in range(4):
            self.add_module(f"linear_fuse_{i}", _BasicConv3d(embedding_dim * 2, embedding_dim))
        self.dropout_ratio = float(dropout_ratio)
        self.linear_pred = nn.Conv3d(embedding_dim, num_classes, kernel_size=1)
        self.sync_group = None          # SyncBatchNorm group under data parallelism (run_training.py:83)
        self.dropout_mask = None        # fp32 [B, embedding_dim] of 0 / 1: overrides the Bernoulli draw (tests)

    def forward(self, inputs):
        vol = inputs[0] if isinstance(inputs, (tuple, list)) else inputs
        size = tuple(vol.shape[2:])
        feats, _ = self.encoder(inputs)
        if len(feats) != 5:
            raise ValueError(f"SegFormerHead needs five feature maps, the encoder returned {len(feats)}")
        lin = lambda m, c: ops.linear(c, m.proj.weight, m.proj.bias)
        x = lin(self.linear_c4, feats[4])
        for i in (3, 2, 1, 0):
            f = getattr(self, f"linear_fuse_{i}")
            cat = ops.upsample_concat(feats[i].shape[1:4], [x, lin(getattr(self, f"linear_c{i}"), feats[i])])
            x = ops.gelu(ops.batch_norm(ops.linear(cat, f.conv.weight, f.conv.bias), f.bn, self.sync_group))
        x = ops.dropout3d(x, self.dropout_ratio, self.training, self.dropout_mask)
        op = getattr(self, "_pred_op", None)
        if op is None or op.w is not self.linear_pred.weight:
            op = self._pred_op = Conv1(self.linear_pred.weight, self.linear_pred.bias)
        return _PredUpsampleFn.apply(x, self.linear_pred.weight, self.linear_pred.bias, size, self.num_classes, op)


class _PredUpsampleFn(torch.autograd.Function):
    """linear_pred (1x1x1 conv to the classes) + the final trilinear upsampling to the input size, on 16-byte logits
    rows [.., 8] (segformer_head_official.py:86-90)"""

    @staticmethod
    def forward(ctx, x, weight, bias, size, ncls, op):
        x = x if x.is_contiguous() else x.contiguous()
        B, d0, d1, d2, _ = x.shape
        low = torch.zeros(B, d0, d1, d2, LOGIT_LD, dtype=x.dtype, device=x.device)
        op.fwd(x, low[..., :ncls])
        logits = torch.empty((B,) + tuple(size) + (LOGIT_LD,), dtype=x.dtype, device=x.device)
        hip.interp_trilinear(low, logits)
        ctx.op, ctx.ncls, ctx.low_shape = op, ncls, low.shape
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)
        return logits[..., :ncls].permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, dlogits):
        if dlogits is None:
            return None, None, None, None, None, None
        x, = ctx.saved_tensors
        T = x.dtype
        B, C, D, H, W = dlogits.shape
        from ..losses import channels_last_grad
        dl = channels_last_grad(dlogits, LOGIT_LD, T)
        if dl is None:
            dl = torch.zeros(B, D, H, W, LOGIT_LD, dtype=T, device=dlogits.device)
            hip.to_channels_last(dlogits.contiguous(), dl[..., :C])
        dlow = torch.empty(ctx.low_shape, dtype=T, device=x.device)
        hip.interp_trilinear_bwd(dl, dlow)
        dx = ctx.op.bwd(x, dlow, ctx.needs_input_grad[0], dy_channels=LOGIT_LD)
        return dx, None, None, None, None, None
