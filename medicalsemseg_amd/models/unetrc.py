"""UNETR conv decoder (``UNETRC``) of the reference, ``/root/reference/models/segmentors/unetr.py:9-52,195-289``
(SURVEY.md 8(a) row A13): a U-shaped fusion of four token feature maps of a ViT-style encoder with the raw input, built
from ``Conv3d k3 -> BatchNorm3d -> ReLU`` units and ``ConvTranspose3d k2 s2`` upsamplers.

MI355X design: the whole decoder is ONE autograd node over channels-last tensors in the compute dtype.  A token map
``[B, L, E]`` *is* the channels-last volume ``[B, d, h, w, E]``, so the reference's transpose + view disappears; every
``torch.cat`` is replaced by its two producers writing into the halves of one buffer; BatchNorm runs on the InstanceNorm
kernels over the merged batch (``layers.BatchNormAct``), with the statistics taken from the convolution's epilogue.
The encoder is any ``nn.Module`` with the attributes the reference reads (``embed_dim``, ``vol_size``, ``patch_size``)
that returns four ``[B, L, E]`` tensors; gradients flow back into it through the node's inputs.
State-dict keys equal the reference's (``decoder0.0.block.0.block.weight`` ...).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import hip
from ..layers import BatchNormAct, Conv1, Conv3, ConvBNAct, Deconv2

LOGIT_LD = 8


class _Keyed(nn.Module):
    """holds a module under the attribute ``block`` -- the key layout of the reference's Single*/Conv3D/Deconv3D blocks"""

    def __init__(self, inner):
        super().__init__()
        self.block = inner


def _up2(i, o):
    return _Keyed(nn.ConvTranspose3d(i, o, kernel_size=2, stride=2))


def _conv(i, o, k=3):
    return _Keyed(nn.Conv3d(i, o, kernel_size=k, stride=1, padding=(k - 1) // 2))


def _cbr(i, o):          # Conv3DBlock
    return _Keyed(nn.Sequential(_conv(i, o), nn.BatchNorm3d(o), nn.ReLU(True)))


def _dbr(i, o):          # Deconv3DBlock
    return _Keyed(nn.Sequential(_up2(i, o), _conv(o, o), nn.BatchNorm3d(o), nn.ReLU(True)))


# ---- op records: each has fwd(x, out=None) -> (y, saved) and bwd(saved, dy, need_dx) -> dx -----------------------
class _Up:
    def __init__(self, m: _Keyed):
        self.op = Deconv2(m.block.weight, m.block.bias)

    def fwd(self, x, out=None):
        return self.op.fwd(x, out), x

    def bwd(self, x, dy, need_dx=True):
        return self.op.bwd(x, dy, need_dx)


class _Cbr:
    def __init__(self, conv: _Keyed, bn: nn.BatchNorm3d, group=None):
        self.op = ConvBNAct(Conv3(conv.block.weight, conv.block.bias), BatchNormAct(bn, 0.0, group))

    def fwd(self, x, out=None):
        return self.op.fwd(x, out)

    def bwd(self, saved, dy, need_dx=True):
        return self.op.bwd(saved, dy, need_dx)


def _ops_of(m, group=None):
    """op records of a Conv3DBlock / Deconv3DBlock / Single* block or a Sequential of them, in execution order"""
    if isinstance(m, nn.Sequential) and not isinstance(m, _Keyed):
        return [o for sub in m for o in _ops_of(sub, group)]
    inner = m.block
    if isinstance(inner, nn.ConvTranspose3d):
        return [_Up(m)]
    if isinstance(inner, nn.Conv3d):
        return [m]                               # the 1x1x1 output conv: handled by the caller
    mods = list(inner)
    if isinstance(mods[0].block, nn.ConvTranspose3d):
        return [_Up(mods[0]), _Cbr(mods[1], mods[2], group)]
    return [_Cbr(mods[0], mods[1], group)]


class UNETRC(nn.Module):
    graph_safe = False   # the encoder in front is arbitrary torch code

    def __init__(self, encoder, in_chans=1, output_dim=3, compute_dtype=torch.bfloat16):
        super().__init__()
        self.encoder = encoder
        self.embed_dim = E = encoder.embed_dim
        self.in_chans, self.output_dim, self.compute_dtype = in_chans, output_dim, compute_dtype
        self.patch_dim = [int(v // p) for v, p in zip(encoder.vol_size, encoder.patch_size)]
        self.decoder0 = nn.Sequential(_cbr(in_chans, 32), _cbr(32, 64))
        self.decoder3 = nn.Sequential(_dbr(E, 512), _dbr(512, 256), _dbr(256, 128))
        self.decoder6 = nn.Sequential(_dbr(E, 512), _dbr(512, 256))
        self.decoder9 = _dbr(E, 512)
        self.decoder12_upsampler = _up2(E, 512)
        self.decoder9_upsampler = nn.Sequential(_cbr(1024, 512), _cbr(512, 512), _cbr(512, 512), _up2(512, 256))
        self.decoder6_upsampler = nn.Sequential(_cbr(512, 256), _cbr(256, 256), _up2(256, 128))
        self.decoder3_upsampler = nn.Sequential(_cbr(256, 128), _cbr(128, 128), _up2(128, 64))
        self.decoder0_header = nn.Sequential(_cbr(128, 64), _cbr(64, 64), _conv(64, output_dim, 1))
        self.sync_group = None      # SyncBatchNorm group under data parallelism (parallel.convert_sync_batchnorm)
        self._build_ops()

    def _build_ops(self):
        head = list(self.decoder0_header)
        self._head = Conv1(head[-1].block.weight, head[-1].block.bias)
        # (first-half branch fed by a token map / the input, second-half chain fed by the level below, channels of a half)
        g = getattr(self, "sync_group", None)
        self._branch = {"z12": _ops_of(self.decoder12_upsampler, g), "z9": _ops_of(self.decoder9, g), "z6": _ops_of(self.decoder6, g),
                        "z3": _ops_of(self.decoder3, g), "x": _ops_of(self.decoder0, g)}
        self._trunk = {9: _ops_of(self.decoder9_upsampler, g), 6: _ops_of(self.decoder6_upsampler, g),
                       3: _ops_of(self.decoder3_upsampler, g), 0: _ops_of(nn.Sequential(*head[:-1]), g)}

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._build_ops()
        return r

    def _dec_params(self):
        mods = (self.decoder0, self.decoder3, self.decoder6, self.decoder9, self.decoder12_upsampler,
                self.decoder9_upsampler, self.decoder6_upsampler, self.decoder3_upsampler, self.decoder0_header)
        return [p for m in mods for p in m.parameters()]

    def forward(self, x_in):
        vol = x_in[0] if isinstance(x_in, (tuple, list)) else x_in
        if not vol.is_cuda:
            raise RuntimeError("medicalsemseg_amd.UNETRC runs on the GPU only (no CPU fallback)")
        z3, z6, z9, z12 = self.encoder(vol)
        return _UNETRCFn.apply(self, vol, z3, z6, z9, z12, *self._dec_params())


def _run(ops, x, out=None):
    """forward through a chain; the last op writes into `out` (a half of a concat buffer) when given"""
    saved = []
    for i, op in enumerate(ops):
        x, s = op.fwd(x, out if i == len(ops) - 1 else None)
        saved.append(s)
    return x, saved


def _run_bwd(ops, saved, g, need_dx=True):
    for i in range(len(ops) - 1, -1, -1):
        g = ops[i].bwd(saved[i], g, need_dx or i > 0)
    return g


class _UNETRCFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net: UNETRC, vol, z3, z6, z9, z12, *params):
        T = net.compute_dtype
        B, Cin, D, H, W = vol.shape
        dev = vol.device
        d, h, w = net.patch_dim
        E = net.embed_dim
        x = torch.empty(B, D, H, W, Cin, dtype=T, device=dev)
        hip.to_channels_last(vol if vol.dtype in (torch.float32, torch.bfloat16) else vol.float(), x)
        tok = lambda z: z.detach().to(T).reshape(B, d, h, w, E).contiguous()   # [B, L, E] is already channels-last
        z = {"z3": tok(z3), "z6": tok(z6), "z9": tok(z9), "z12": tok(z12)}
        S = {}
        cat = lambda lvl, c: torch.empty(B, d * lvl, h * lvl, w * lvl, 2 * c, dtype=T, device=dev)
        c9 = cat(2, 512)
        _, S["z9"] = _run(net._branch["z9"], z["z9"], c9[..., :512])
        _, S["z12"] = _run(net._branch["z12"], z["z12"], c9[..., 512:])
        c6 = cat(4, 256)
        _, S["z6"] = _run(net._branch["z6"], z["z6"], c6[..., :256])
        _, S[9] = _run(net._trunk[9], c9, c6[..., 256:])
        c3 = cat(8, 128)
        _, S["z3"] = _run(net._branch["z3"], z["z3"], c3[..., :128])
        _, S[6] = _run(net._trunk[6], c6, c3[..., 128:])
        c0 = cat(16, 64)
        _, S["x"] = _run(net._branch["x"], x, c0[..., :64])
        _, S[3] = _run(net._trunk[3], c3, c0[..., 64:])
        y, S[0] = _run(net._trunk[0], c0)
        logits_cl = torch.empty(B, D, H, W, LOGIT_LD, dtype=T, device=dev)
        net._head.fwd(y, logits_cl[..., :net.output_dim])
        if any(ctx.needs_input_grad):
            ctx.net, ctx.S, ctx.last, ctx.n_in = net, S, y, 6 + len(params)
            ctx.zdt = (z3.dtype, z6.dtype, z9.dtype, z12.dtype)
        ctx.set_materialize_grads(False)
        return logits_cl[..., :net.output_dim].permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, dlogits):
        net, S = ctx.net, ctx.S
        if dlogits is None:
            return (None,) * ctx.n_in
        T = net.compute_dtype
        B, C, D, H, W = dlogits.shape
        from ..losses import channels_last_grad
        dl = channels_last_grad(dlogits, LOGIT_LD, T)
        if dl is None:
            dl = torch.zeros(B, D, H, W, LOGIT_LD, dtype=T, device=dlogits.device)
            hip.to_channels_last(dlogits.contiguous(), dl[..., :C])
        need = ctx.needs_input_grad
        g = net._head.bwd(ctx.last, dl, True, dy_channels=LOGIT_LD)
        dc0 = _run_bwd(net._trunk[0], S[0], g)
        _run_bwd(net._branch["x"], S["x"], dc0[..., :64], need_dx=False)
        dc3 = _run_bwd(net._trunk[3], S[3], dc0[..., 64:])
        dz3 = _run_bwd(net._branch["z3"], S["z3"], dc3[..., :128], need_dx=need[2])
        dc6 = _run_bwd(net._trunk[6], S[6], dc3[..., 128:])
        dz6 = _run_bwd(net._branch["z6"], S["z6"], dc6[..., :256], need_dx=need[3])
        dc9 = _run_bwd(net._trunk[9], S[9], dc6[..., 256:])
        dz9 = _run_bwd(net._branch["z9"], S["z9"], dc9[..., :512], need_dx=need[4])
        dz12 = _run_bwd(net._branch["z12"], S["z12"], dc9[..., 512:], need_dx=need[5])
        ctx.S = None
        back = lambda gz, dt, on: gz.reshape(B, -1, net.embed_dim).to(dt) if on and gz is not None else None
        return (None, None, back(dz3, ctx.zdt[0], need[2]), back(dz6, ctx.zdt[1], need[3]), back(dz9, ctx.zdt[2], need[4]),
                back(dz12, ctx.zdt[3], need[5])) + (None,) * (ctx.n_in - 6)
