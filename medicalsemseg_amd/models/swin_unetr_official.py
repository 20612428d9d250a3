"""The MONAI / "official" Swin-UNETR on the HIP kernels (BASELINE configs[3] names it literally: "Swin-UNETR 48-feat").

Mirror of ``/root/reference/models/segmentors/swin_unetr_official.py`` (a vendored MONAI ``SwinUNETR``): window 7 with
``get_window_size`` clamping (:371-388), relative-position index built for the full window and sliced ``[:n, :n]`` when the
window is clamped (:477-480), shift mask per stage (:726-763), Linear patch merging over 8 strided sub-grids with the
reference's duplicated sub-grids x2/x5 and x3/x6 (:699-708), un-affine ``proj_out`` layer norm (:955-968), decoder wiring
of ``SwinUNETR.forward`` (:282-295: encoder10 on the deepest feature, decoder5 takes the RAW stage-3 feature as its skip).
Parameter / buffer names equal the reference's state-dict keys (``swinViT.layers1.0.blocks.0.attn.qkv.weight`` ...).

Same execution model as ``models/swin_unetr.py``: channels-last token volumes, the encoder a chain of single-kernel
autograd ops, the conv decoder ONE autograd node over ``layers.py``.  Window padding is explicit here (pad -> attention on
a whole number of windows -> crop), so that the padded tokens' ``qkv.bias`` gradient is exact; the clamped stage passes
``bias_ws = 7`` to the attention kernel (``msseg_window_attention_fwd2``).  Only cubic volumes / windows are implemented.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn

from .. import hip, ops
from ..layers import Conv1
from .swin_unetr import _UnetOutBlock, _UnetrBasicBlock, _UnetrUpBlock
from .unet import LOGIT_LD


def _rel_index(ws: int) -> torch.Tensor:
    r = torch.arange(ws)
    c = torch.stack(torch.meshgrid(r, r, r, indexing="ij")).flatten(1)
    rel = c[:, :, None] - c[:, None, :] + (ws - 1)
    m = 2 * ws - 1
    return rel[0] * m * m + rel[1] * m + rel[2]


class _WindowAttention(nn.Module):
    def __init__(self, dim, heads, ws, qkv_bias=True):
        super().__init__()
        self.heads, self.ws = heads, ws
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 3, heads))
        self.register_buffer("relative_position_index", _rel_index(ws))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)


class _MLPBlock(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.linear1 = nn.Linear(dim, hidden)
        self.linear2 = nn.Linear(hidden, dim)


class _Block(nn.Module):
    def __init__(self, dim, heads, ws, shift, mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.window_size, self.shift_size, self.heads = ws, shift, heads
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _WindowAttention(dim, heads, ws, qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _MLPBlock(dim, int(dim * mlp_ratio))

    def forward(self, x):
        B, d, h, w, C = x.shape
        if not (d == h == w):
            raise NotImplementedError("only cubic token grids are implemented")
        ws, shift = self.window_size, self.shift_size
        if d <= ws:                                   # get_window_size: clamp the window, no shift
            ws, shift = d, 0
        a = self.attn
        x, xn = ops.layer_norm_res(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)   # x: the residual, through the node
        pad = (ws - d % ws) % ws
        if pad:                                       # zero tokens AFTER the norm (their qkv is the bias), cropped below
            xn = ops.box_resize(xn, (d + pad, h + pad, w + pad))
        qkv = ops.linear(xn, a.qkv.weight, a.qkv.bias)
        y = ops.WindowAttnFn.apply(qkv, a.qkv.bias, a.relative_position_bias_table, self.heads, ws, shift, a.ws)
        if pad:
            y = ops.box_resize(y, (d, h, w))
        x = ops.linear_add(y, a.proj.weight, a.proj.bias, x)   # per-token: commutes with the crop; residual add in the epilogue
        x, y = ops.layer_norm_res(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return ops.mlp(y, self.mlp.linear1.weight, self.mlp.linear1.bias, self.mlp.linear2.weight, self.mlp.linear2.bias, res=x)


class _PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(8 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(8 * dim)

    def forward(self, x):
        B, d, h, w, C = x.shape
        # sub-grids in the reference's order, including its duplicates (x5 == x2, x6 == x3); an odd grid is zero-padded by the
        # gather itself (one kernel for F.pad + eight strided slices + torch.cat, one for their backward)
        sub = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 0), (0, 0, 1), (1, 1, 1)]
        x = ops.merge_gather(x, sub)
        x = ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return ops.linear(x, self.reduction.weight, None)


class _BasicLayer(nn.Module):
    def __init__(self, dim, depth, heads, ws):
        super().__init__()
        self.blocks = nn.ModuleList([_Block(dim, heads, ws, 0 if i % 2 == 0 else ws // 2) for i in range(depth)])
        self.downsample = _PatchMerging(dim)

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return self.downsample(x)


class _PatchEmbed(nn.Module):
    def __init__(self, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=2, stride=2)


class _SwinViT(nn.Module):
    def __init__(self, in_chans, embed_dim, ws, depths, heads):
        super().__init__()
        self.patch_embed = _PatchEmbed(in_chans, embed_dim)
        for i in range(4):
            setattr(self, f"layers{i + 1}", nn.ModuleList([_BasicLayer(embed_dim * 2 ** i, depths[i], heads[i], ws)]))

    def forward(self, x_cl, normalize=True):
        pe = self.patch_embed.proj
        x = ops.PatchConvFn.apply(x_cl, pe.weight, pe.bias, 2)
        outs = [ops.layer_norm(x, None, None, 1e-5) if normalize else x]
        for i in range(4):
            x = getattr(self, f"layers{i + 1}")[0](x)
            outs.append(ops.layer_norm(x, None, None, 1e-5) if normalize else x)
        return outs


class SwinUNETR(nn.Module):
    """``model((vol[B,C,D,H,W], crop_loc, affine)) -> logits[B,out,D,H,W]`` -- or a bare volume, as MONAI's inferer feeds it."""

    def __init__(self, img_size, in_channels: int, out_channels: int, depths: Sequence[int] = (2, 2, 2, 2),
                 num_heads: Sequence[int] = (3, 6, 12, 24), feature_size: int = 24, normalize: bool = True,
                 window_size: int = 7, compute_dtype=torch.bfloat16):
        super().__init__()
        img = (img_size,) * 3 if isinstance(img_size, int) else tuple(img_size)
        if any(m % 32 for m in img):
            raise ValueError("input image size (img_size) should be divisible by stage-wise image resolution.")
        if feature_size % 12 != 0:
            raise ValueError("feature_size should be divisible by 12.")
        if len(depths) != 4 or len(num_heads) != 4:
            raise ValueError("four stages are expected")
        fs = feature_size
        self.normalize, self.out_channels, self.compute_dtype = normalize, out_channels, compute_dtype
        self.swinViT = _SwinViT(in_channels, fs, window_size, depths, num_heads)
        self.encoder1 = _UnetrBasicBlock(in_channels, fs)
        self.encoder2 = _UnetrBasicBlock(fs, fs)
        self.encoder3 = _UnetrBasicBlock(2 * fs, 2 * fs)
        self.encoder4 = _UnetrBasicBlock(4 * fs, 4 * fs)
        self.encoder10 = _UnetrBasicBlock(16 * fs, 16 * fs)
        self.decoder5 = _UnetrUpBlock(16 * fs, 8 * fs, 2)
        self.decoder4 = _UnetrUpBlock(8 * fs, 4 * fs, 2)
        self.decoder3 = _UnetrUpBlock(4 * fs, 2 * fs, 2)
        self.decoder2 = _UnetrUpBlock(2 * fs, fs, 2)
        self.decoder1 = _UnetrUpBlock(fs, fs, 2)
        self.out = _UnetOutBlock(fs, out_channels)
        self._build_ops()

    def _build_ops(self):
        self._top = self.encoder10.layer.op()
        # (up block, skip encoder or None for a raw feature skip), deepest first
        self._levels = [(self.decoder5.op(), None), (self.decoder4.op(), self.encoder4.layer.op()),
                        (self.decoder3.op(), self.encoder3.layer.op()), (self.decoder2.op(), self.encoder2.layer.op()),
                        (self.decoder1.op(), self.encoder1.layer.op())]
        self._out_op = Conv1(self.out.conv.conv.weight, self.out.conv.conv.bias)

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._build_ops()
        return r

    def load_from(self, weights):
        """copy a self-supervised MONAI Swin-ViT checkpoint into the encoder (swin_unetr_official.py:232-280 and the
        per-block mapper :617-649): keys ``module.patch_embed.proj.*``, ``module.layers{1..4}.0.blocks.{n}.*``,
        ``module.layers{1..4}.0.downsample.{reduction,norm}.*`` of ``weights["state_dict"]``"""
        sd = weights["state_dict"]
        block_names = ["norm1.weight", "norm1.bias", "attn.relative_position_bias_table", "attn.relative_position_index",
                       "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "norm2.weight",
                       "norm2.bias", "mlp.linear1.weight", "mlp.linear1.bias", "mlp.linear2.weight", "mlp.linear2.bias"]
        src_names = dict(zip(block_names, block_names))
        src_names.update({"mlp.linear1.weight": "mlp.fc1.weight", "mlp.linear1.bias": "mlp.fc1.bias",
                          "mlp.linear2.weight": "mlp.fc2.weight", "mlp.linear2.bias": "mlp.fc2.bias"})
        own = dict(self.swinViT.named_parameters())
        own.update(dict(self.swinViT.named_buffers()))
        with torch.no_grad():
            own["patch_embed.proj.weight"].copy_(sd["module.patch_embed.proj.weight"])
            own["patch_embed.proj.bias"].copy_(sd["module.patch_embed.proj.bias"])
            for li in range(1, 5):
                layer = getattr(self.swinViT, f"layers{li}")[0]
                for bname, _ in layer.blocks.named_children():
                    for n in block_names:
                        own[f"layers{li}.0.blocks.{bname}.{n}"].copy_(sd[f"module.layers{li}.0.blocks.{bname}.{src_names[n]}"])
                for n in ("reduction.weight", "norm.weight", "norm.bias"):
                    own[f"layers{li}.0.downsample.{n}"].copy_(sd[f"module.layers{li}.0.downsample.{n}"])
        from .. import layers as _layers
        _layers.bump_weights_epoch()      # packed weight images are rebuilt on the next forward

    def forward(self, x_in):
        vol = x_in[0] if isinstance(x_in, (tuple, list)) else x_in
        if not vol.is_cuda:
            raise RuntimeError("SwinUNETR runs on the GPU only (no CPU fallback)")
        B, Cin, D, H, W = vol.shape
        if any(int(v) % 32 for v in (D, H, W)):
            raise ValueError("volume must be divisible by 32")
        x_cl = torch.empty(B, D, H, W, Cin, dtype=self.compute_dtype, device=vol.device)
        hip.to_channels_last(vol if vol.dtype in (torch.float32, torch.bfloat16) else vol.float(), x_cl)
        hs = self.swinViT(x_cl, self.normalize)
        dec_params = [p for n, m in self.named_children() if n != "swinViT" for p in m.parameters()]
        return _DecoderFn.apply(self, x_cl, *hs, *dec_params)


class _DecoderFn(torch.autograd.Function):
    """dec4 = encoder10(hs4); dec3 = decoder5(dec4, hs3); dec2 = decoder4(dec3, encoder4(hs2)); ...;
    out = decoder1(dec0, encoder1(x_in)); logits = out(out)   (swin_unetr_official.py:282-295)"""

    @staticmethod
    def forward(ctx, net: SwinUNETR, x_cl, hs0, hs1, hs2, hs3, hs4, *params):
        srcs = [hs3, hs2, hs1, hs0, x_cl]
        x, s_top = net._top.fwd(hs4.contiguous())
        saved = []
        for (up, enc), src in zip(net._levels, srcs):
            cat = up.alloc_cat(x)
            cout = up.cout
            if enc is not None:
                _, s_e = enc.fwd(src.contiguous(), out=cat[..., cout:])
            else:
                cat[..., cout:].copy_(src)        # raw feature skip
                s_e = None
            x, s_d = up.fwd(x, cat)
            saved.append((s_e, s_d))
        N, D, H, W, _ = x.shape
        logits_cl = torch.empty(N, D, H, W, LOGIT_LD, dtype=x.dtype, device=x.device)
        net._out_op.fwd(x, logits_cl[..., :net.out_channels])
        if any(ctx.needs_input_grad):
            ctx.net, ctx.saved, ctx.s_top, ctx.last, ctx.n_in = net, saved, s_top, x, 7 + len(params)
            ctx.hs_needs = list(ctx.needs_input_grad[2:7])
        ctx.set_materialize_grads(False)
        return logits_cl[..., :net.out_channels].permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, dlogits):
        net, saved = ctx.net, ctx.saved
        if dlogits is None:
            return (None,) * ctx.n_in
        from ..losses import channels_last_grad
        N, C, D, H, W = dlogits.shape
        dl = channels_last_grad(dlogits, LOGIT_LD, net.compute_dtype)
        if dl is None:
            dl = torch.zeros(N, D, H, W, LOGIT_LD, dtype=net.compute_dtype, device=dlogits.device)
            hip.to_channels_last(dlogits.contiguous(), dl[..., :C])
        g = net._out_op.bwd(ctx.last, dl, True, dy_channels=LOGIT_LD)
        d_src = [None] * 5                       # gradients of [hs3, hs2, hs1, hs0, x_cl]
        needs = [ctx.hs_needs[3], ctx.hs_needs[2], ctx.hs_needs[1], ctx.hs_needs[0], False]
        for k in range(4, -1, -1):
            up, enc = net._levels[k]
            s_e, s_d = saved[k]
            g, dskip = up.bwd(s_d, g)
            if enc is not None:
                d_src[k] = enc.bwd(s_e, dskip, need_dx=needs[k])
            else:
                d_src[k] = dskip.contiguous() if needs[k] else None
        d_hs4 = net._top.bwd(ctx.s_top, g, need_dx=ctx.hs_needs[4])
        ctx.saved = None
        return (None, None, d_src[3], d_src[2], d_src[1], d_src[0], d_hs4) + (None,) * (ctx.n_in - 7)
