"""``torch.autograd.Function`` wrappers of single HIP ops on channels-last tensors ``[B, S, H, W, C]``.

Used by the Swin encoder (``models/swin_unetr.py``), whose block algebra is a chain of small ops; the conv-heavy
networks (UNet, the UNETR decoder) use whole-network functions over ``layers.py`` instead.  Every forward/backward
here is one or two kernel launches through the C ABI; there is no torch arithmetic on activations.
"""
from __future__ import annotations

import torch

from . import hip, layers

def _gbuf(p):
    """Parameter gradients are written (or accumulated) in place into `p.grad` by the kernels and the autograd functions
    return None for them: no AccumulateGrad nodes, no extra add kernels, and the step stays hipGraph-capturable
    (as the whole-network functions of layers.py do)."""
    return layers._grad_buf(p)


def _packed(weight, dtype, kind, builder):
    """packed-weight images live on the parameter object: built once, refreshed by the batched repack of
    layers.PACK_REGISTRY after each optimiser step (instead of one pack launch per forward / backward call)"""
    c = getattr(weight, "_msseg_packed", None)
    if c is None:
        c = layers.PackedCache()
        weight._msseg_packed = c
    return c.get(weight, dtype, kind, builder)


def _c(x):
    return x if x.is_contiguous() else x.contiguous()


def _linear_param_grads(x, dy, weight, bias, cin, cout, want_w, want_b):
    """weight.grad (+)= dy^T x, bias.grad (+)= dy.sum(tokens): one pass over the tokens where the fused kernel takes the shape"""
    want_b = want_b and bias is not None
    if want_w and hip.linear_wgrad_ok(x, cin, cout):
        gw, aw = _gbuf(weight)
        gb, ab = _gbuf(bias) if want_b else (None, False)
        hip.linear_wgrad(x, dy, gw.view(cout, cin), gb, cin, cout, aw, ab)
        return
    if want_w:
        g, acc = _gbuf(weight)
        hip.conv3d_k1_wgrad(x, dy, g.view(cout, cin), cin, cout, acc)
    if want_b:
        g, acc = _gbuf(bias)
        hip.channel_sum(dy, g, acc)


class LinearFn(torch.autograd.Function):
    """y = x @ W^T + b on the last dim (nn.Linear): the 1x1x1 igemm on tokens."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _c(x)
        cout, cin = weight.shape[:2]           # nn.Linear weight, or a 1x1x1 conv weight [cout, cin, 1, 1, 1]
        T = x.dtype
        wp = _packed(weight, T, "f", lambda: hip.pack_conv_k1(weight.detach().reshape(cout, cin), T))
        y = torch.empty(x.shape[:-1] + (cout,), dtype=T, device=x.device)
        hip.conv3d_k1(x, wp, bias, y, cin, cout)
        ctx.save_for_backward(x)
        ctx.weight, ctx.bias = weight, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        dy = _c(dy)
        cout, cin = weight.shape[:2]
        T = x.dtype
        dx = None
        if ctx.needs_input_grad[0]:
            wpd = _packed(weight, T, "d", lambda: hip.pack_conv_k1(weight.detach().reshape(cout, cin), T, dgrad=True))
            dx = torch.empty_like(x)
            hip.conv3d_k1(dy, wpd, None, dx, cout, cin)
        _linear_param_grads(x, dy, weight, bias, cin, cout, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, None, None


class LinearAddFn(torch.autograd.Function):
    """res + Linear(x): the attention projection of a Swin block with the residual add in the Linear kernel's epilogue
    (hip.linear_add_ok shapes; bit-identical to linear followed by add)"""

    @staticmethod
    def forward(ctx, x, weight, bias, res):
        x, res = _c(x), _c(res)
        cout, cin = weight.shape[:2]
        T = x.dtype
        wp = _packed(weight, T, "f", lambda: hip.pack_conv_k1(weight.detach().reshape(cout, cin), T))
        y = torch.empty(x.shape[:-1] + (cout,), dtype=T, device=x.device)
        hip.linear_add(x, wp, bias, res, y, cin, cout)
        ctx.save_for_backward(x)
        ctx.weight, ctx.bias = weight, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        dy = _c(dy)
        cout, cin = weight.shape[:2]
        T = x.dtype
        dx = None
        if ctx.needs_input_grad[0]:
            wpd = _packed(weight, T, "d", lambda: hip.pack_conv_k1(weight.detach().reshape(cout, cin), T, dgrad=True))
            dx = torch.empty_like(x)
            hip.conv3d_k1(dy, wpd, None, dx, cout, cin)
        _linear_param_grads(x, dy, weight, bias, cin, cout, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, None, None, dy


def linear_add(x, weight, bias, res):
    """res + linear(x, weight, bias): one launch where the fused kernel takes the shape, linear + add otherwise"""
    cout, cin = weight.shape[:2]
    if x.is_cuda and res.shape[:-1] == x.shape[:-1] and res.shape[-1] == cout and hip.linear_add_ok(x, res, cin, cout):
        return LinearAddFn.apply(x, weight, bias, res)
    return add(res, linear(x, weight, bias))


class MlpFn(torch.autograd.Function):
    """fc2(gelu(fc1(x))) -- the MLP of a Swin block (/root/reference/models/backbones/swin_nnformer.py:24-42) -- as ONE
    autograd node, so that the GELU rides on the Linear kernels' epilogues: forward fc1 writes the pre-activation and the
    activation from one launch, backward fc2's input gradient is multiplied by gelu'(pre) as it is stored.  Two passes over
    the 4C-wide hidden tensor less in each direction; values bit-identical to linear -> gelu -> linear.  Shapes the fused
    kernel does not take (hip.linear_gelu_ok) run the three kernels."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res=None):
        """res (optional, shaped like the output): the result is res + fc2(...), the add in fc2's epilogue where the kernel takes
        the shape (hip.linear_add_ok), a separate add pass otherwise"""
        x = _c(x)
        chid, cin = w1.shape[:2]
        cout = w2.shape[0]
        T = x.dtype
        wp1 = _packed(w1, T, "f", lambda: hip.pack_conv_k1(w1.detach().reshape(chid, cin), T))
        wp2 = _packed(w2, T, "f", lambda: hip.pack_conv_k1(w2.detach().reshape(cout, chid), T))
        pre = torch.empty(x.shape[:-1] + (chid,), dtype=T, device=x.device)
        act = torch.empty_like(pre)
        ctx.fused = hip.linear_gelu_ok(x, cin, chid)
        if ctx.fused:
            hip.linear_gelu_fwd(x, wp1, b1, pre, act, cin, chid)
        else:
            hip.conv3d_k1(x, wp1, b1, pre, cin, chid)
            hip.gelu_fwd(pre, act)
        y = torch.empty(x.shape[:-1] + (cout,), dtype=T, device=x.device)
        ctx.has_res = res is not None
        if res is not None:
            res = _c(res)
            if hip.linear_add_ok(act, res, chid, cout):
                hip.linear_add(act, wp2, b2, res, y, chid, cout)
            else:
                hip.conv3d_k1(act, wp2, b2, y, chid, cout)
                lin, y = y, torch.empty_like(y)
                epc = 16 // y.element_size()
                if (y.numel() // y.shape[0]) % epc == 0 and res.data_ptr() % 16 == 0:
                    hip.axpy_rows(res, lin, None, y)
                else:
                    hip.add(res, lin, y)
        else:
            hip.conv3d_k1(act, wp2, b2, y, chid, cout)
        ctx.save_for_backward(x, pre, act)
        ctx.params = (w1, b1, w2, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pre, act = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        dy = _c(dy)
        chid, cin = w1.shape[:2]
        cout = w2.shape[0]
        T = x.dtype
        wpd2 = _packed(w2, T, "d", lambda: hip.pack_conv_k1(w2.detach().reshape(cout, chid), T, dgrad=True))
        dpre = torch.empty_like(pre)
        if ctx.fused:
            hip.linear_gelu_bwd(dy, wpd2, pre, dpre, cout, chid)
        else:
            dact = torch.empty_like(pre)
            hip.conv3d_k1(dy, wpd2, None, dact, cout, chid)
            hip.gelu_bwd(pre, dact, dpre)
        _linear_param_grads(act, dy, w2, b2, chid, cout, ctx.needs_input_grad[3], ctx.needs_input_grad[4])
        dx = None
        if ctx.needs_input_grad[0]:
            wpd1 = _packed(w1, T, "d", lambda: hip.pack_conv_k1(w1.detach().reshape(chid, cin), T, dgrad=True))
            dx = torch.empty_like(x)
            hip.conv3d_k1(dpre, wpd1, None, dx, chid, cin)
        _linear_param_grads(x, dpre, w1, b1, cin, chid, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        return dx, None, None, None, None, (dy if ctx.has_res else None)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = _c(x)
        y = torch.empty_like(x)
        mean, rstd = hip.layernorm_fwd(x, gamma, beta, y, eps)
        ctx.save_for_backward(x, mean, rstd)
        ctx.gamma, ctx.beta = gamma, beta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.gamma, ctx.beta
        dy = _c(dy)
        dx = torch.empty_like(x)
        dg = db = None
        acc = False
        if ctx.needs_input_grad[1]:
            dg, acc = _gbuf(gamma)
            db, _ = _gbuf(beta)
        hip.layernorm_bwd(x, gamma, mean, rstd, dy, dx, dg, db, acc)
        return dx, None, None, None


class LayerNormResFn(torch.autograd.Function):
    """(x, LayerNorm(x)) as ONE autograd node: a Swin block uses x twice -- normalised into the attention / MLP branch and as
    the residual (swin_nnformer.py:243-262) -- and autograd sums the two gradients of a tensor with two consumers with a
    separate add kernel.  Here the node hands x through as its first output, so x has one consumer, and the backward adds the
    residual's gradient inside the LayerNorm backward kernel (msseg_layernorm_bwd_add): same rounding, one pass less."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = _c(x)
        y = torch.empty_like(x)
        mean, rstd = hip.layernorm_fwd(x, gamma, beta, y, eps)
        ctx.save_for_backward(x, mean, rstd)
        ctx.gamma, ctx.beta = gamma, beta
        ctx.set_materialize_grads(False)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dres, dy):
        x, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.gamma, ctx.beta
        if dy is None:
            return dres, None, None, None
        dy = _c(dy)
        dx = torch.empty_like(x)
        dg = db = None
        acc = False
        if ctx.needs_input_grad[1]:
            dg, acc = _gbuf(gamma)
            db, _ = _gbuf(beta)
        if dres is not None and hip.layernorm_bwd_add_ok(x):
            hip.layernorm_bwd(x, gamma, mean, rstd, dy, dx, dg, db, acc, add=_c(dres))
        else:
            hip.layernorm_bwd(x, gamma, mean, rstd, dy, dx, dg, db, acc)
            if dres is not None:
                dx = dx + dres
        return dx, None, None, None


def layer_norm_res(x, gamma, beta, eps=1e-5):
    """(x, LayerNorm(x)); take the residual from the first result (see LayerNormResFn)"""
    return LayerNormResFn.apply(x, gamma, beta, eps)


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return hip.gelu_fwd(x, torch.empty_like(x))

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        return hip.gelu_bwd(x, _c(dy), torch.empty_like(x))


class AddFn(torch.autograd.Function):
    """a + scale[n] * b: the residual add of a Swin block; `scale` (fp32 [B] or None) is the stochastic-depth factor
    mask[n] / keep of /root/reference/models/layers/drop_path.py:15-45, folded into the add (one kernel, no product tensor)"""

    @staticmethod
    def forward(ctx, a, b, scale=None):
        a, b = _c(a), _c(b)
        ctx.scale = scale
        epc = 16 // a.element_size()
        if (a.numel() // a.shape[0]) % epc == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0:
            return hip.axpy_rows(a, b, scale, torch.empty_like(a))
        if scale is not None:
            raise ValueError("stochastic depth needs 16-byte aligned dense activations")
        return hip.add(a, b, torch.empty_like(a))

    @staticmethod
    def backward(ctx, dy):
        if ctx.scale is None:
            return dy, dy, None
        dy = _c(dy)
        return dy, hip.axpy_rows(None, dy, ctx.scale, torch.empty_like(dy)), None


class BoxCopyFn(torch.autograd.Function):
    """zero padding at the high end of the spatial axes of a token volume [B, D, H, W, C] (F.pad(x, (0, 0, 0, pw, 0, ph, 0, pd))) or
    the crop x[:, :d, :h, :w] back, as one pass each way (csrc/layout.hip; the two are each other's adjoint)"""

    @staticmethod
    def forward(ctx, x, size):
        x = _c(x)
        ctx.src_size = tuple(x.shape[1:4])
        return hip.box_copy(x, torch.empty((x.shape[0],) + tuple(size) + (x.shape[4],), dtype=x.dtype, device=x.device))

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty((dy.shape[0],) + ctx.src_size + (dy.shape[4],), dtype=dy.dtype, device=dy.device)
        return hip.box_copy(dy, dx), None


def _chunked(x) -> bool:
    """channel count a multiple of the 16-byte chunk and a 16-byte aligned base: what the layout kernels need (GPU tensors only:
    there is no CPU path)"""
    if not x.is_cuda:
        raise RuntimeError("medicalsemseg_amd ops run on the GPU only (no CPU fallback)")
    return x.shape[-1] % (16 // x.element_size()) == 0 and x.data_ptr() % 16 == 0


def box_resize(x, size):
    """x zero-padded (at the high end) or cropped to the spatial size `size`"""
    size = tuple(int(v) for v in size)
    if not _chunked(x):      # odd channel counts: the torch ops this replaces
        d, h, w = x.shape[1:4]
        if all(s >= v for s, v in zip(size, (d, h, w))):
            return torch.nn.functional.pad(x, (0, 0, 0, size[2] - w, 0, size[1] - h, 0, size[0] - d))
        return x[:, :size[0], :size[1], :size[2], :].contiguous()
    return BoxCopyFn.apply(x, size)


class MergeGatherFn(torch.autograd.Function):
    """torch.cat([x[:, a::2, b::2, c::2, :] for (a, b, c) in offsets], -1) of a token volume, odd sizes zero-padded first: the
    sub-grid gather of MONAI's PatchMerging (/root/reference/models/segmentors/swin_unetr_official.py:699-708) in one pass,
    its adjoint in one pass (duplicated sub-grids are summed in slot order)"""

    @staticmethod
    def forward(ctx, x, offsets):
        x = _c(x)
        B, D, H, W, C = x.shape
        ctx.subs, ctx.shape = hip.merge_subs(offsets), tuple(x.shape)
        out = torch.empty(B, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2, 8 * C, dtype=x.dtype, device=x.device)
        return hip.merge_gather(x, out, ctx.subs)

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        return hip.merge_gather_bwd(dy, torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device), ctx.subs), None


def merge_gather(x, offsets):
    if not _chunked(x):      # odd channel counts: the torch ops this replaces
        d, h, w = x.shape[1:4]
        xp = torch.nn.functional.pad(x, (0, 0, 0, w % 2, 0, h % 2, 0, d % 2))
        return torch.cat([xp[:, a::2, b::2, c::2, :] for a, b, c in offsets], -1)
    return MergeGatherFn.apply(x, tuple(tuple(o) for o in offsets))


class WindowAttnFn(torch.autograd.Function):
    """shifted-window attention core on a token volume: qkv [B,S,H,W,3C] -> [B,S,H,W,C]"""

    @staticmethod
    def forward(ctx, qkv, qkv_bias, table, heads, ws, shift, bias_ws=None):
        qkv = _c(qkv)
        B, S, H, W, C3 = qkv.shape
        out = torch.empty(B, S, H, W, C3 // 3, dtype=qkv.dtype, device=qkv.device)
        qb = qkv_bias.detach() if qkv_bias is not None else None
        tab = table.detach().contiguous()
        lse = hip.window_attention_fwd(qkv, qb, tab, out, heads, ws, shift, bias_ws)
        ctx.save_for_backward(qkv, qb, tab, out, lse)
        ctx.cfg = (heads, ws, shift, bias_ws)
        ctx.table = table
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, qb, tab, out, lse = ctx.saved_tensors
        heads, ws, shift, bias_ws = ctx.cfg
        dqkv = torch.empty_like(qkv)
        dtable = None
        if ctx.needs_input_grad[2]:
            table = ctx.table
            if table.is_contiguous():
                dtable, acc = _gbuf(table)       # the kernel adds into the buffer
                if not acc:
                    dtable.zero_()
            else:
                dtable = torch.zeros_like(tab)
        hip.window_attention_bwd(qkv, qb, tab, out, lse, _c(dout), dqkv, dtable, heads, ws, shift, bias_ws)
        # gradient w.r.t. qkv_bias through PADDED tokens (only when the grid is not a window multiple) is dropped
        return dqkv, None, (None if ctx.table.is_contiguous() else dtable), None, None, None, None


class Conv3Fn(torch.autograd.Function):
    """Conv3d k3 p1, stride 1 or 2 (+bias) on [B,D,H,W,C]"""

    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        x = _c(x)
        cout, cin = weight.shape[0], weight.shape[1]
        T = x.dtype
        B, D, H, W, _ = x.shape
        if stride == 1:
            vol = (B, D, H, W)
            wp = _packed(weight, T, ("f", vol), lambda: hip.pack_conv_k3(weight.detach().contiguous(), T, vol=vol))
            y = torch.empty(B, D, H, W, cout, dtype=T, device=x.device)
            hip.conv3d_k3(x, wp, bias, y, cin, cout)
        else:
            wp = _packed(weight, T, "f2", lambda: hip.pack_conv_k3(weight.detach().contiguous(), T))
            y = torch.empty(B, (D - 1) // 2 + 1, (H - 1) // 2 + 1, (W - 1) // 2 + 1, cout, dtype=T, device=x.device)
            hip.conv3d_k3s2(x, wp, bias, y, cin, cout)
        ctx.save_for_backward(x)
        ctx.weight, ctx.bias = weight, bias
        ctx.stride = stride
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        dy = _c(dy)
        cout, cin = weight.shape[0], weight.shape[1]
        T = x.dtype
        if bias is not None and ctx.needs_input_grad[2]:
            g, acc = _gbuf(bias)
            hip.channel_sum(dy, g, acc)
        if ctx.stride == 2:   # stride-1 problems on the zero-stuffed gradient
            dyz = torch.empty(x.shape[:-1] + (cout,), dtype=T, device=x.device)
            hip.zero_stuff2(dy, dyz)
            dy = dyz
        dx = None
        if ctx.needs_input_grad[0]:
            dvol = tuple(dy.shape[:4])
            wpd = _packed(weight, T, ("d", dvol),
                          lambda: hip.pack_conv_k3(weight.detach().contiguous(), T, dgrad=True, vol=dvol))
            dx = torch.empty_like(x)
            hip.conv3d_k3(dy, wpd, None, dx, cout, cin)
        if ctx.needs_input_grad[1]:
            g, acc = _gbuf(weight)
            hip.conv3d_k3_wgrad(x, dy, g, cin, cout, acc)
        return dx, None, None, None


def _patchify(x, k):
    """[B, D, H, W, C] -> [B, D/k, H/k, W/k, C k^3] with the patch flattened in the weight's own (ci, kd, kh, kw) order"""
    B, D, H, W, C = x.shape
    return x.view(B, D // k, k, H // k, k, W // k, k, C).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(B, D // k, H // k, W // k, C * k ** 3)


def _unpatchify(xs, k, C):
    B, d, h, w, _ = xs.shape
    return xs.view(B, d, h, w, C, k, k, k).permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(B, d * k, h * k, w * k, C)


class PatchConvFn(torch.autograd.Function):
    """Conv3d with kernel k, stride s (default k), padding p outside the k3 kernels' reach.

    * few input channels (PatchEmbed3D.proj k = s = 2, SegFormer's OverlapPatchEmbed k7 s4 p3 on the raw volume): the
      im2col-style gather kernel; no input gradient (the input is the data).
    * k = s, p = 0, channel count a multiple of the 16-byte chunk (the spatial-reduction conv of SegFormer's attention,
      segformer_backbone.py:76-78): non-overlapping patches are a layout change, so the conv is ONE flat GEMM on the
      weight's own [Cout, Cin k^3] view after a space-to-depth copy -- forward, input gradient and weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, k, s=None, p=0):
        x = _c(x)
        s = k if s is None else s
        cout, cin = weight.shape[0], weight.shape[1]
        T = x.dtype
        B, D, H, W, _ = x.shape
        ctx.weight, ctx.bias = weight, bias
        ctx.k, ctx.s, ctx.p = k, s, p
        ctx.gemm = (s == k and p == 0 and cin % (16 // x.element_size()) == 0 and D % k == 0 and H % k == 0 and W % k == 0)
        if ctx.gemm:
            K = cin * k ** 3
            xs = _patchify(x, k)
            wp = _packed(weight, T, "pf", lambda: hip.pack_conv_k1(weight.detach().reshape(cout, K), T))
            y = torch.empty(xs.shape[:-1] + (cout,), dtype=T, device=x.device)
            hip.conv3d_k1(xs, wp, bias, y, K, cout)
            ctx.save_for_backward(xs)
            return y
        wp = _packed(weight, T, "g", lambda: hip.pack_conv_gather(weight.detach().contiguous(), T))
        od, oh, ow = ((v + 2 * p - k) // s + 1 for v in (D, H, W))
        y = torch.empty(B, od, oh, ow, cout, dtype=T, device=x.device)
        hip.conv3d_gather(x, wp, bias, y, cin, cout, k, s, p)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        dy = _c(dy)
        cout, cin = weight.shape[0], weight.shape[1]
        k = ctx.k
        if ctx.gemm:
            K = cin * k ** 3
            T = x.dtype
            _linear_param_grads(x, dy, weight, bias, K, cout, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
            dx = None
            if ctx.needs_input_grad[0]:
                wpd = _packed(weight, T, "pd", lambda: hip.pack_conv_k1(weight.detach().reshape(cout, K), T, dgrad=True))
                dxs = torch.empty_like(x)
                hip.conv3d_k1(dy, wpd, None, dxs, cout, K)
                dx = _unpatchify(dxs, k, cin)
            return dx, None, None, None, None, None
        if bias is not None and ctx.needs_input_grad[2]:
            g, acc = _gbuf(bias)
            hip.channel_sum(dy, g, acc)
        if ctx.needs_input_grad[1]:
            g, acc = _gbuf(weight)
            hip.conv3d_gather_wgrad(x, dy, g, cin, cout, k, ctx.s, ctx.p, acc)
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("input gradient of a gather conv (few channels / overlapping patches) is not on this path")
        return None, None, None, None, None, None


class DwConv3Fn(torch.autograd.Function):
    """depthwise Conv3d k3 p1 (groups = channels) on a token volume [B, S, H, W, C]; weight [C, 1, 3, 3, 3] (+bias):
    nn.Conv3d(C, C, 3, padding=1, groups=C) of /root/reference/models/backbones/swindepth.py:36-41"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _c(x)
        C = x.shape[-1]
        taps = weight.detach().reshape(C, 27).t().to(x.dtype).contiguous()  # tap-major [27, C] in the compute dtype
        ctx.save_for_backward(x, taps)
        ctx.weight, ctx.bias = weight, bias
        return hip.dwconv3d_k3(x, taps, bias.detach().float() if bias is not None else None, torch.empty_like(x))

    @staticmethod
    def backward(ctx, dy):
        x, taps = ctx.saved_tensors
        dy = _c(dy)
        w, b = ctx.weight, ctx.bias
        want_w, want_b = w.requires_grad, b is not None and b.requires_grad
        if want_w or want_b:
            gw, aw = _gbuf(w) if want_w else (None, False)
            gb, ab = _gbuf(b) if want_b else (None, False)
            hip.dwconv3d_k3_wgrad(x, dy, gw, gb, aw, ab)
        dx = hip.dwconv3d_k3(dy, taps, None, torch.empty_like(dy), flip=True) if ctx.needs_input_grad[0] else None
        return dx, None, None


class AvgPool3Fn(torch.autograd.Function):
    """nn.AvgPool3d(kernel_size=3, stride=1, padding=1) (count_include_pad: every window divides by 27) on a token volume
    (/root/reference/models/backbones/swinception.py:113-116): the depthwise k3 kernel with unit taps and the fp32 sum scaled
    by 1/27 (`msseg_avgpool3d_k3`; a bf16 tap of 1/27 would be 0.0371094: +0.195 % on every output).  The operator is
    symmetric, so the input gradient is the same launch on dy."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        return hip.avgpool3d_k3(x, torch.empty_like(x))

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        return hip.avgpool3d_k3(dy, torch.empty_like(dy))


class BatchNormFn(torch.autograd.Function):
    """nn.BatchNorm3d on a channels-last volume (layers.BatchNormAct: training statistics + running-statistics update,
    eval statistics, optional cross-rank synchronisation); `weight` / `bias` are passed so autograd sees the parameters"""

    @staticmethod
    def forward(ctx, x, bn, weight, bias, group):
        x = _c(x)
        op = layers.BatchNormAct(bn, 1.0, group)
        y, stats = op.fwd(x)
        ctx.op = op
        ctx.save_for_backward(x, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats = ctx.saved_tensors
        return ctx.op.bwd(x, stats, _c(dy)), None, None, None, None


class InterpFn(torch.autograd.Function):
    """F.interpolate(mode='trilinear', align_corners=False) on a channels-last volume"""

    @staticmethod
    def forward(ctx, x, size):
        x = _c(x)
        ctx.in_shape = x.shape
        y = torch.empty((x.shape[0],) + tuple(size) + (x.shape[-1],), dtype=x.dtype, device=x.device)
        return hip.interp_trilinear(x, y)

    @staticmethod
    def backward(ctx, dy):
        dx = torch.empty(ctx.in_shape, dtype=dy.dtype, device=dy.device)
        return hip.interp_trilinear_bwd(_c(dy), dx), None


class UpsampleConcatFn(torch.autograd.Function):
    """cat([interpolate(t, size) for t in tensors], channel) with every piece written straight into its channel slice of
    the result (SegFormer head, /root/reference/models/segmentors/segformer_head_official.py:72-84); a tensor that
    already has the target size is copied by the same kernel (all interpolation weights are 0 / 1)."""

    @staticmethod
    def forward(ctx, size, *tensors):
        ts = [_c(t) for t in tensors]
        B, T, dev = ts[0].shape[0], ts[0].dtype, ts[0].device
        chans = [t.shape[-1] for t in ts]
        cat = torch.empty((B,) + tuple(size) + (sum(chans),), dtype=T, device=dev)
        off = 0
        for t, c in zip(ts, chans):
            hip.interp_trilinear(t, cat[..., off:off + c])
            off += c
        ctx.shapes, ctx.chans = [t.shape for t in ts], chans
        return cat

    @staticmethod
    def backward(ctx, dcat):
        dcat = _c(dcat)
        grads, off = [], 0
        for shp, c, need in zip(ctx.shapes, ctx.chans, ctx.needs_input_grad[1:]):
            g = None
            if need:
                g = torch.empty(shp, dtype=dcat.dtype, device=dcat.device)
                hip.interp_trilinear_bwd(dcat[..., off:off + c], g)
            grads.append(g)
            off += c
        return (None, *grads)


class KvAttnFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd)) v with the keys / values of a spatially reduced token set (SegFormer's attention,
    /root/reference/models/backbones/segformer_backbone.py:96-117): q [B, N, C], kv [B, M, 2C] -> [B, N, C]"""

    @staticmethod
    def forward(ctx, q, kv, heads):
        q, kv = _c(q), _c(kv)
        scale = float(q.shape[-1] // heads) ** -0.5
        o, lse = hip.kv_attention_fwd(q, kv, heads, scale)
        ctx.save_for_backward(q, kv, o, lse)
        ctx.heads, ctx.scale = heads, scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, o, lse = ctx.saved_tensors
        dq, dkv = hip.kv_attention_bwd(q, kv, o, lse, _c(do), ctx.heads, ctx.scale)
        return dq, dkv, None


class ScaleChannelsFn(torch.autograd.Function):
    """x * scale[n, c]: nn.Dropout3d with the keep mask / (1 - p) given as `scale` (fp32 [B, C])"""

    @staticmethod
    def forward(ctx, x, scale):
        x = _c(x)
        ctx.save_for_backward(scale)
        return hip.scale_channels(x, scale, torch.empty_like(x))

    @staticmethod
    def backward(ctx, dy):
        scale, = ctx.saved_tensors
        dy = _c(dy)
        return hip.scale_channels(dy, scale, torch.empty_like(dy)), None


def interp_trilinear(x, size):
    return InterpFn.apply(x, tuple(size))


def upsample_concat(size, tensors):
    return UpsampleConcatFn.apply(tuple(size), *tensors)


def kv_attention(q, kv, heads):
    return KvAttnFn.apply(q, kv, heads)


def dropout3d(x, p, training, mask=None):
    """channel dropout; `mask` (fp32 [B, C] of 0 / 1) overrides the Bernoulli draw"""
    if not training or p == 0.0:
        return x
    if mask is None:
        mask = torch.empty(x.shape[0], x.shape[-1], device=x.device, dtype=torch.float32).bernoulli_(1.0 - p)
    return ScaleChannelsFn.apply(x, (mask.to(x.device, torch.float32) / (1.0 - p)).contiguous())


def dwconv3(x, weight, bias=None):
    return DwConv3Fn.apply(x, weight, bias)


def avg_pool3(x):
    return AvgPool3Fn.apply(x)


def batch_norm(x, bn, group=None):
    return BatchNormFn.apply(x, bn, bn.weight, bn.bias, group)


def linear(x, weight, bias=None):
    return LinearFn.apply(x, weight, bias)


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(x, gamma, beta, eps)


def gelu(x):
    return GeluFn.apply(x)


def mlp(x, w1, b1, w2, b2, res=None):
    """fc2(gelu(fc1(x))) [+ res] with the GELU (and the residual add) fused into the Linear kernels where the shape allows (MlpFn)"""
    return MlpFn.apply(x, w1, b1, w2, b2, res)


def add(a, b, scale=None):
    return AddFn.apply(a, b, scale)
