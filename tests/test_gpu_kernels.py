"""GPU parity tests: every HIP kernel (through the C ABI / ctypes) against stock torch-CPU fp32 ops.

Tolerances: fp32 path rtol 1e-4 on outputs (north_star's logits tolerance); bf16 path is compared with the
same fp32 reference evaluated on bf16-rounded inputs/weights, to ~1 bf16 ulp of the output scale.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def cl(x_ncdhw, dtype, dev):
    """NCDHW cpu -> channels-last gpu"""
    return x_ncdhw.permute(0, 2, 3, 4, 1).contiguous().to(dev, dtype)


def ncdhw(x_cl):
    return x_cl.float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def rnd(dtype, *ts):
    """round tensors through the compute dtype (what the kernel will actually see)"""
    out = [t.to(dtype).float() for t in ts]
    return out if len(out) > 1 else out[0]


def check(got, ref, dtype, what, scale=None):
    got, ref = got.float().cpu(), ref.float().cpu()
    s = float(ref.abs().max()) if scale is None else scale
    s = max(s, 1e-6)
    tol = 2e-5 if dtype == torch.float32 else 6e-3
    err = float((got - ref).abs().max()) / s
    assert err < tol, f"{what}: max rel-to-scale error {err:.3e} (tol {tol})"


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,sp", [(32, 32, (16, 16, 16)), (64, 32, (32, 32, 32)), (32, 64, (12, 12, 12)),
                                         (48, 48, (12, 12, 24)), (16, 16, (8, 8, 8)), (128, 256, (6, 6, 6)),
                                         (96, 48, (6, 10, 18)), (8, 24, (5, 7, 9)),
                                         # small grids, two channel blocks per stage: odd block counts (3, 5), a partial last block
                                         (96, 64, (6, 10, 18)), (160, 32, (8, 8, 8)), (72, 32, (12, 12, 12)),
                                         # grids large enough for the ping-pong kernel (bf16, 32 channels per stage)
                                         (32, 32, (32, 48, 64)), (32, 64, (30, 29, 70)), (64, 32, (34, 31, 50)),
                                         # 48-channel layers (Swin-UNETR decoder): partial 32-blocks in the ping-pong wgrad
                                         (48, 48, (32, 32, 64)), (96, 48, (30, 33, 36))])
def test_conv3d_k3_fwd_dgrad_wgrad(dtype, cin, cout, sp):
    from medicalsemseg_amd import hip
    from medicalsemseg_amd.layers import Conv3
    dev = _dev()
    N = 2
    x = gen(N, cin, *sp, seed=1)
    w = gen(cout, cin, 3, 3, 3, seed=2, scale=(cin * 27) ** -0.5)
    b = gen(cout, seed=3)
    dy = gen(N, cout, *sp, seed=4)
    xr, wr, dyr = rnd(dtype, x, w, dy)
    xr.requires_grad_(True); wr.requires_grad_(True)
    yref = F.conv3d(xr, wr, b, padding=1)
    yref.backward(dyr)
    wp = torch.nn.Parameter(w.to(dev)); bp = torch.nn.Parameter(b.to(dev))
    op = Conv3(wp, bp)
    xg = cl(x, dtype, dev)
    y = op.fwd(xg)
    check(ncdhw(y), yref.detach(), dtype, "conv3d_k3 fwd")
    if dtype == torch.bfloat16 and sp in ((32, 32, 64), (30, 33, 36)):
        # the 48-channel ping-pong kernel takes the 48-input-channel launches of these two layers: forward of 48 -> 48,
        # input gradient of both (48 gradient channels in)
        assert hip.lib().msseg_conv3d_k3_kernel(N, *sp, 48, cin, hip.BF16) == 4
    dyg = cl(dy, dtype, dev)
    dx = op.bwd(xg, dyg, True)
    check(ncdhw(dx), xr.grad, dtype, "conv3d_k3 dgrad")
    check(wp.grad, wr.grad, dtype, "conv3d_k3 wgrad")
    check(bp.grad, dyr.sum((0, 2, 3, 4)), dtype, "conv3d_k3 bias grad")
    # accumulate path
    op.bwd(xg, dyg, False)
    check(wp.grad, 2 * wr.grad, dtype, "conv3d_k3 wgrad accumulate")


@pytest.mark.parametrize("sp,N", [((32, 32, 64), 2), ((30, 33, 36), 3)])
def test_conv3d_k3_96_channels_as_two_48_channel_launches(sp, N):
    """Conv3.fwd(want_stats) of a 96 -> 48 layer over a concat buffer (Swin-UNETR decoder, swin_unetr.py:73-128 of the
    reference) = two launches of the 48-channel kernel on the buffer's channel halves, the second accumulating onto the
    first's stored result and emitting the InstanceNorm statistics"""
    from medicalsemseg_amd import hip
    from medicalsemseg_amd.layers import Conv3
    dev, dtype = _dev(), torch.bfloat16
    x = gen(N, 96, *sp, seed=1)
    w = gen(48, 96, 3, 3, 3, seed=2, scale=(96 * 27) ** -0.5)
    b = gen(48, seed=3)
    xr, wr = rnd(dtype, x, w)
    yref = F.conv3d(xr, wr, b, padding=1)
    op = Conv3(torch.nn.Parameter(w.to(dev)), torch.nn.Parameter(b.to(dev)))
    xg = cl(x, dtype, dev)
    assert op.halves_ok((N, *sp), dtype) == 48
    y, stats = op.fwd(xg, want_stats=True)
    check(ncdhw(y), yref, dtype, "conv 96 -> 48 as two halves")
    yf = y.float().reshape(N, -1, 48)
    ref = torch.stack([yf.sum(1), (yf * yf).sum(1)], dim=-1)
    assert float((stats - ref).abs().max()) / float(ref.abs().max()) < 1e-5
    y2, stats2 = op.fwd(xg, want_stats=True)
    assert torch.equal(y2, y) and torch.equal(stats2, stats)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3d_k3_channel_slices(dtype):
    """input and output as channel slices of wider (concat) buffers"""
    from medicalsemseg_amd import hip
    dev = _dev()
    x = gen(1, 32, 8, 8, 16, seed=5)
    w = gen(16, 32, 3, 3, 3, seed=6, scale=0.05)
    xr, wr = rnd(dtype, x, w)
    yref = F.conv3d(xr, wr, None, padding=1)
    big_in = torch.zeros(1, 8, 8, 16, 64, dtype=dtype, device=dev)
    big_in[..., 32:] = cl(x, dtype, dev)
    big_out = torch.full((1, 8, 8, 16, 48), 7.0, dtype=dtype, device=dev)
    wp = hip.pack_conv_k3(w.to(dev), dtype, vol=(1, 8, 8, 16))
    hip.conv3d_k3(big_in[..., 32:], wp, None, big_out[..., 16:32], 32, 16)
    check(ncdhw(big_out[..., 16:32]), yref, dtype, "conv slice out")
    assert float((big_out[..., :16].float() - 7).abs().max()) == 0 and float((big_out[..., 32:].float() - 7).abs().max()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,k,s,p,sp", [(1, 32, 3, 1, 1, (16, 16, 16)), (1, 48, 2, 2, 0, (12, 12, 12)),
                                                (4, 16, 3, 1, 1, (8, 8, 8)), (2, 48, 2, 2, 0, (10, 6, 14)),
                                                # 1x1x1 conv of a one-channel volume (UnetResBlock conv3 of Swin-UNETR's encoder1):
                                                # bf16 runs on the stem kernels (centre tap only), ragged tiles
                                                (1, 48, 1, 1, 0, (9, 14, 35)), (1, 32, 1, 1, 0, (16, 16, 16))])
def test_conv3d_gather(dtype, cin, cout, k, s, p, sp):
    from medicalsemseg_amd import hip
    dev = _dev()
    N = 2
    x = gen(N, cin, *sp, seed=1)
    w = gen(cout, cin, k, k, k, seed=2, scale=(cin * k ** 3) ** -0.5)
    b = gen(cout, seed=3)
    xr, wr = rnd(dtype, x, w)
    wr.requires_grad_(True)
    yref = F.conv3d(xr, wr, b, stride=s, padding=p)
    dy = gen(*yref.shape, seed=4)
    dyr = rnd(dtype, dy)
    yref.backward(dyr)
    xg = cl(x, dtype, dev)
    wp = hip.pack_conv_gather(w.to(dev), dtype)
    y = torch.empty(N, *yref.shape[2:], cout, dtype=dtype, device=dev)
    hip.conv3d_gather(xg, wp, b.to(dev), y, cin, cout, k, s, p)
    check(ncdhw(y), yref.detach(), dtype, "gather fwd")
    dw = torch.empty(cout, cin, k, k, k, device=dev)
    hip.conv3d_gather_wgrad(xg, cl(dy, dtype, dev), dw, cin, cout, k, s, p)
    check(dw, wr.grad, dtype, "gather wgrad")
    if cin == 1 and k == 1 and dtype == torch.bfloat16:
        # the same conv through the stem entry point with fused InstanceNorm statistics (layers.Conv1.fwd(want_stats))
        y2 = torch.empty_like(y)
        stats = torch.full((N, cout, 2), float("nan"), device=dev)
        hip.conv3d_stem(xg, wp, b.to(dev), y2, cout, stats, k=1)
        assert torch.equal(y2, y)
        yf = y.float().reshape(N, -1, cout)
        ref = torch.stack([yf.sum(1), (yf * yf).sum(1)], dim=-1)
        assert float((stats - ref).abs().max()) / float(ref.abs().max()) < 1e-5


@pytest.mark.parametrize("cout,sp,N", [(32, (16, 16, 16), 2), (64, (10, 13, 21), 3), (32, (33, 20, 48), 2),
                                       # 48-wide cout blocks: Swin-UNETR's 1 -> 48 first conv
                                       (48, (16, 16, 16), 2), (48, (9, 14, 35), 3), (96, (8, 8, 16), 1)])
def test_conv3d_stem(cout, sp, N):
    """one-input-channel stem kernels (bf16): forward + fused InstanceNorm statistics + weight gradient, on grids that
    are not tile multiples"""
    from medicalsemseg_amd import hip
    dev = _dev()
    dtype = torch.bfloat16
    x = gen(N, 1, *sp, seed=1)
    w = gen(cout, 1, 3, 3, 3, seed=2, scale=27 ** -0.5)
    b = gen(cout, seed=3)
    xr, wr = rnd(dtype, x, w)
    wr.requires_grad_(True)
    yref = F.conv3d(xr, wr, b, padding=1)
    dy = gen(*yref.shape, seed=4)
    dyr = rnd(dtype, dy)
    yref.backward(dyr)
    xg = cl(x, dtype, dev)
    wp = hip.pack_conv_gather(w.to(dev), dtype)
    y = torch.empty(N, *sp, cout, dtype=dtype, device=dev)
    stats = torch.full((N, cout, 2), float("nan"), device=dev)
    hip.conv3d_stem(xg, wp, b.to(dev), y, cout, stats)
    check(ncdhw(y), yref.detach(), dtype, "stem fwd")
    yf = y.float().reshape(N, -1, cout)
    ref = torch.stack([yf.sum(1), (yf * yf).sum(1)], dim=-1)
    assert float((stats - ref).abs().max()) / float(ref.abs().max()) < 1e-5
    s2 = torch.empty_like(stats)
    y2 = torch.empty_like(y)
    hip.conv3d_stem(xg, wp, b.to(dev), y2, cout, s2)
    assert torch.equal(y2, y) and torch.equal(s2, stats)
    # the gather entry points route to the same kernels
    y3 = torch.empty_like(y)
    hip.conv3d_gather(xg, wp, b.to(dev), y3, 1, cout, 3, 1, 1)
    assert torch.equal(y3, y)
    dw = torch.empty(cout, 1, 3, 3, 3, device=dev)
    hip.conv3d_gather_wgrad(xg, cl(dy, dtype, dev), dw, 1, cout, 3, 1, 1)
    check(dw, wr.grad, dtype, "stem wgrad")
    dw2 = dw.clone()
    hip.conv3d_gather_wgrad(xg, cl(dy, dtype, dev), dw2, 1, cout, 3, 1, 1, True)
    check(dw2, 2 * wr.grad, dtype, "stem wgrad accumulate")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,sp", [(32, 32, (8, 8, 8)), (256, 128, (3, 3, 3)), (96, 48, (6, 6, 6)), (48, 48, (4, 6, 10)),
                                         # register-resident-weight kernels (bf16): ragged W segments, both channel counts
                                         (64, 32, (6, 7, 9)), (32, 32, (5, 6, 20)), (64, 32, (3, 4, 33)),
                                         # sliced-output kernels (bf16, csrc/deconv_k2s2_gen.hip): Swin-UNETR's and the deep
                                         # BasicUNet shapes, ragged W segments, one- and two-child slices
                                         (48, 48, (3, 5, 19)), (96, 48, (2, 3, 17)), (128, 64, (3, 4, 7)), (192, 96, (3, 5, 7)),
                                         (384, 192, (3, 3, 3)), (768, 384, (2, 3, 3)), (256, 128, (2, 3, 18))])
def test_deconv_k2s2(dtype, cin, cout, sp):
    from medicalsemseg_amd.layers import Deconv2
    dev = _dev()
    N = 2
    x = gen(N, cin, *sp, seed=1)
    w = gen(cin, cout, 2, 2, 2, seed=2, scale=cin ** -0.5)
    b = gen(cout, seed=3)
    xr, wr = rnd(dtype, x, w)
    xr.requires_grad_(True); wr.requires_grad_(True)
    yref = F.conv_transpose3d(xr, wr, b, stride=2)
    dy = gen(*yref.shape, seed=4)
    dyr = rnd(dtype, dy)
    yref.backward(dyr)
    wp = torch.nn.Parameter(w.to(dev)); bp = torch.nn.Parameter(b.to(dev))
    op = Deconv2(wp, bp)
    xg = cl(x, dtype, dev)
    y = op.fwd(xg)
    check(ncdhw(y), yref.detach(), dtype, "deconv fwd")
    dx = op.bwd(xg, cl(dy, dtype, dev), True)
    check(ncdhw(dx), xr.grad, dtype, "deconv bwd data")
    check(wp.grad, wr.grad, dtype, "deconv wgrad")
    check(bp.grad, dyr.sum((0, 2, 3, 4)), dtype, "deconv bias grad")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout", [(32, 3), (96, 48), (48, 14), (16, 2)])
def test_conv3d_k1(dtype, cin, cout):
    from medicalsemseg_amd.layers import Conv1
    dev = _dev()
    sp = (6, 10, 12)
    x = gen(2, cin, *sp, seed=1)
    w = gen(cout, cin, 1, 1, 1, seed=2, scale=cin ** -0.5)
    b = gen(cout, seed=3)
    xr, wr = rnd(dtype, x, w)
    xr.requires_grad_(True); wr.requires_grad_(True)
    yref = F.conv3d(xr, wr, b)
    dy = gen(*yref.shape, seed=4)
    dyr = rnd(dtype, dy)
    yref.backward(dyr)
    wp = torch.nn.Parameter(w.to(dev)); bp = torch.nn.Parameter(b.to(dev))
    op = Conv1(wp, bp)
    xg = cl(x, dtype, dev)
    ld = ((cout + 7) // 8) * 8
    ybuf = torch.zeros(2, *sp, ld, dtype=dtype, device=dev)
    op.fwd(xg, ybuf[..., :cout])
    check(ncdhw(ybuf[..., :cout]), yref.detach(), dtype, "k1 fwd")
    dybuf = torch.zeros(2, *sp, ld, dtype=dtype, device=dev)
    dybuf[..., :cout] = cl(dy, dtype, dev)
    dx = op.bwd(xg, dybuf, True, dy_channels=ld)
    check(ncdhw(dx), xr.grad, dtype, "k1 dgrad")
    check(wp.grad, wr.grad, dtype, "k1 wgrad")
    check(bp.grad, dyr.sum((0, 2, 3, 4)), dtype, "k1 bias grad")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,affine,res,slope", [(32, True, False, 0.1), (48, False, True, 0.01), (48, False, False, 1.0),
                                                 (20, True, True, 0.01)])
def test_instnorm_act(dtype, C, affine, res, slope):
    from medicalsemseg_amd.layers import InstNormAct
    dev = _dev()
    sp = (8, 8, 10)
    x = gen(2, C, *sp, seed=1) * 2 + 0.5
    r = gen(2, C, *sp, seed=2)
    ga = torch.nn.Parameter((gen(C, seed=3) * 0.2 + 1).to(dev)) if affine else None
    be = torch.nn.Parameter((gen(C, seed=4) * 0.2).to(dev)) if affine else None
    xr, rr = rnd(dtype, x, r)
    xr.requires_grad_(True); rr.requires_grad_(True)
    gar = ga.detach().cpu().clone().requires_grad_(True) if affine else None
    ber = be.detach().cpu().clone().requires_grad_(True) if affine else None
    z = F.instance_norm(xr, weight=gar, bias=ber, eps=1e-5)
    if res:
        z = z + rr
    yref = F.leaky_relu(z, slope) if slope != 1.0 else z
    dy = gen(*yref.shape, seed=5)
    dyr = rnd(dtype, dy)
    yref.backward(dyr)
    op = InstNormAct(ga, be, slope)
    xg = cl(x, dtype, dev)
    rg = cl(r, dtype, dev) if res else None
    a, stats = op.fwd(xg, residual=rg)
    check(ncdhw(a), yref.detach(), dtype, "instnorm fwd")
    # the backward mask uses the stored (rounded) output: use the reference's own output sign in bf16
    out = op.bwd(xg, stats, a, cl(dy, dtype, dev), want_dres=res)
    dx = out[0] if res else out
    check(ncdhw(dx), xr.grad, dtype, "instnorm dx", scale=float(xr.grad.abs().max()))
    if res:
        check(ncdhw(out[1]), rr.grad, dtype, "instnorm dres")
    if affine:
        check(ga.grad, gar.grad, dtype, "dgamma", scale=float(gar.grad.abs().max()) * (1 if dtype == torch.float32 else 4))
        check(be.grad, ber.grad, dtype, "dbeta", scale=float(ber.grad.abs().max()) * (1 if dtype == torch.float32 else 4))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [32, 12])
def test_maxpool2(dtype, C):
    from medicalsemseg_amd import hip
    dev = _dev()
    x = gen(2, C, 8, 12, 16, seed=1)
    xr = rnd(dtype, x).requires_grad_(True)
    yref = F.max_pool3d(xr, 2)
    dy = gen(*yref.shape, seed=2)
    dyr = rnd(dtype, dy)
    yref.backward(dyr)
    xg = cl(x, dtype, dev)
    y = torch.empty(2, 4, 6, 8, C, dtype=dtype, device=dev)
    hip.maxpool2_fwd(xg, y)
    check(ncdhw(y), yref.detach(), dtype, "maxpool fwd")
    dx = torch.empty_like(xg)
    hip.maxpool2_bwd(xg, cl(dy, dtype, dev), dx)
    check(ncdhw(dx), xr.grad, dtype, "maxpool bwd")
    base = cl(gen(2, C, 8, 12, 16, seed=3), dtype, dev)
    acc = base.clone()
    hip.maxpool2_bwd(xg, cl(dy, dtype, dev), acc, accumulate=True)
    check(ncdhw(acc), ncdhw(base) + xr.grad, dtype, "maxpool bwd accumulate")


@pytest.mark.parametrize("C,lab_dtype", [(3, torch.float32), (2, torch.int64), (14, torch.uint8)])
def test_dice_ce(C, lab_dtype):
    from medicalsemseg_amd.losses import DiceCELoss, dice_from_counts
    from oracle.losses import dice_ce_loss, dice_metric
    dev = _dev()
    sp = (12, 10, 14)
    logits = gen(2, C, *sp, seed=1) * 2
    g = torch.Generator().manual_seed(2)
    labels = torch.randint(0, C, (2, 1, *sp), generator=g)
    labels[1][labels[1] == C - 1] = 0  # a class absent from sample 1 -> NaN dice there
    lr = logits.clone().requires_grad_(True)
    ref = dice_ce_loss(lr, labels.float(), 1e-5, 1e-5)
    (ref * 3.0).backward()
    crit = DiceCELoss(smooth_nr=1e-5, smooth_dr=1e-5)
    lg = logits.to(dev).requires_grad_(True)
    loss = crit(lg, labels.to(dev).to(lab_dtype))
    (loss * 3.0).backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    np.testing.assert_allclose(lg.grad.cpu().numpy(), lr.grad.numpy(), rtol=1e-4, atol=1e-9)
    score, nn_ = dice_from_counts(crit.last["hard"])
    sref, nref = dice_metric(logits, labels)
    np.testing.assert_allclose(score.cpu().numpy(), sref.numpy(), rtol=1e-6, equal_nan=True)
    assert torch.equal(nn_.cpu(), nref)


def test_sw_gather_blend():
    from medicalsemseg_amd import hip
    dev = _dev()
    vol = gen(2, 20, 24, 28, seed=1)
    out = torch.zeros(3, 20, 24, 28, device=dev)
    cnt = torch.zeros(20, 24, 28, device=dev)
    imp = gen(8, 8, 12, seed=2).abs() + 0.1
    oref, cref = torch.zeros(3, 20, 24, 28), torch.zeros(20, 24, 28)
    for i, st in enumerate([(0, 0, 0), (4, 8, 6), (12, 16, 16), (4, 8, 6)]):
        w = gen(3, 8, 8, 12, seed=10 + i)
        hip.sw_blend(w.to(dev), imp.to(dev), out, cnt, st)
        sl = (slice(None), slice(st[0], st[0] + 8), slice(st[1], st[1] + 8), slice(st[2], st[2] + 12))
        oref[sl] += imp * w
        cref[sl[1:]] += imp
    # same fp32 operation order as the reference loop -> bit-exact
    assert torch.equal(cnt.cpu(), cref), f"cnt max diff {float((cnt.cpu() - cref).abs().max())}"
    assert torch.equal(out.cpu(), oref), f"out max diff {float((out.cpu() - oref).abs().max())}"
    win = torch.empty(2, 8, 8, 12, device=dev)
    hip.sw_gather(vol.to(dev), win, (14, 20, 20), cval=-3.0)
    ref = torch.full((2, 8, 8, 12), -3.0)
    ref[:, :6, :4, :8] = vol[:, 14:20, 20:24, 20:28]
    assert torch.equal(win.cpu(), ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_unet_small_fwd_bwd(dtype):
    """whole UNetSmall forward + DiceCE + backward vs the oracle BasicUNet (same state dict)"""
    from medicalsemseg_amd.losses import DiceCELoss
    from medicalsemseg_amd.models.unet import UNET_FEATURES, UNet
    from oracle.blocks import BasicUNet
    from oracle.losses import dice_ce_loss
    dev = _dev()
    torch.manual_seed(0)
    ref = BasicUNet(1, 2, UNET_FEATURES["UNetSmall"])
    net = UNet(1, 2, UNET_FEATURES["UNetSmall"], compute_dtype=dtype)
    missing = net.load_state_dict(ref.state_dict(), strict=True)
    net = net.to(dev)
    x = gen(2, 1, 32, 32, 32, seed=13)
    g = torch.Generator().manual_seed(14)
    y = torch.randint(0, 2, (2, 1, 32, 32, 32), generator=g).float()
    out_ref = ref(x)
    loss_ref = dice_ce_loss(out_ref, y)
    loss_ref.backward()
    crit = DiceCELoss()
    out = net((x.to(dev), None, None))
    loss = crit(out, y.to(dev))
    loss.backward()
    if dtype == torch.float32:
        np.testing.assert_allclose(out.detach().cpu().numpy(), out_ref.detach().numpy(), rtol=1e-4, atol=1e-4)
        assert abs(float(loss) - float(loss_ref)) < 1e-4
        gtol = 5e-3
    else:
        err = float((out.detach().cpu() - out_ref.detach()).abs().max()) / float(out_ref.abs().max())
        print(f"[bf16] UNet-small logits drift {err:.3e}, |loss diff| {abs(float(loss) - float(loss_ref)):.2e}")
        assert err < 2.5e-2, f"bf16 logits drift {err}"        # measured 1.08e-2
        assert abs(float(loss) - float(loss_ref)) < 1e-3      # measured 5.5e-5
        gtol = 0.3   # bf16 storage of activations AND gradients through 18 conv layers; fp32 path is the parity gate
    pr = dict(ref.named_parameters())
    num = den = 0.0
    worst_p = 0.0
    for name, p in net.named_parameters():
        assert p.grad is not None, name
        if name.endswith("conv.bias") and "final" not in name:
            continue  # conv bias before InstanceNorm: true gradient is exactly 0, both sides hold rounding noise
        gr = pr[name].grad
        d2, r2 = float(((p.grad.cpu() - gr) ** 2).sum()), float((gr ** 2).sum())
        num, den = num + d2, den + r2
        if dtype == torch.float32:
            err = float((p.grad.cpu() - gr).abs().max()) / (float(gr.abs().max()) + 1e-8)
            assert err < gtol, f"{name}: grad rel err {err:.3e}"
        else:
            worst_p = max(worst_p, (d2 / (r2 + 1e-20)) ** 0.5)
            assert (d2 / (r2 + 1e-20)) ** 0.5 < 0.6, f"{name}: bf16 grad rel L2 err {(d2 / r2) ** 0.5:.3e}"   # measured worst 0.26 ... 0.38
    tot = (num / den) ** 0.5
    print(f"[{dtype}] whole-net grad rel L2 err {tot:.3e}, worst single parameter {worst_p:.3e}")
    assert tot < (1e-3 if dtype == torch.float32 else 8e-2)      # bf16 measured 4.2e-2 (8^3 ... 2^3 grids: few values per statistic)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,sp,N", [(32, 32, (32, 32, 32), 2), (32, 64, (12, 12, 12), 3), (16, 48, (6, 6, 6), 1),
                                           (32, 32, (32, 48, 64), 2), (32, 64, (30, 29, 38), 3),
                                           # 48 input channels: the 16-wide-cout-block ping-pong kernel (bf16)
                                           (48, 48, (32, 32, 64), 2), (48, 96, (30, 33, 36), 3), (48, 16, (40, 40, 40), 4)])
def test_conv3d_k3_fused_stats(dtype, cin, cout, sp, N):
    """InstanceNorm statistics from the conv epilogue == statistics of the stored output (separate pass + torch)"""
    from medicalsemseg_amd import hip
    dev = _dev()
    x = gen(N, cin, *sp, seed=1)
    w = gen(cout, cin, 3, 3, 3, seed=2, scale=(cin * 27) ** -0.5)
    b = gen(cout, seed=3)
    xg = cl(x, dtype, dev)
    wp = hip.pack_conv_k3(w.to(dev), dtype, vol=(N, *sp))
    y = torch.empty(N, *sp, cout, dtype=dtype, device=dev)
    stats = torch.full((N, cout, 2), float("nan"), device=dev)
    for _ in range(3):   # the scratch counter must come back to zero after every launch
        hip.conv3d_k3(xg, wp, b.to(dev), y, cin, cout, stats)
    yf = y.float().reshape(N, -1, cout)
    ref = torch.stack([yf.sum(1), (yf * yf).sum(1)], dim=-1)
    sep = hip.channel_stats(y)
    scale = float(ref.abs().max())
    assert float((stats - ref).abs().max()) / scale < 1e-5
    assert float((sep - ref).abs().max()) / scale < 1e-5
    # deterministic: identical bits on a re-run
    s2 = torch.empty_like(stats)
    hip.conv3d_k3(xg, wp, b.to(dev), y, cin, cout, s2)
    assert torch.equal(s2, stats)


def test_batched_weight_packing_equals_single_packing():
    """msseg_pack_weights_batch (one launch per network and step, layers.PACK_REGISTRY) == per-image msseg_pack_weights"""
    from medicalsemseg_amd import hip, layers
    dev = _dev()
    ws = [torch.nn.Parameter(gen(*shape, seed=i).to(dev)) for i, shape in
          enumerate([(32, 32, 3, 3, 3), (64, 32, 3, 3, 3), (48, 96, 3, 3, 3), (3, 32, 1, 1, 1), (32, 1, 3, 3, 3),
                     (48, 48, 3, 3, 3),       # the four-part image of the 48-channel kernel (bf16)
                     (64, 32, 3, 3, 3), (24, 40, 1, 1, 1), (64, 32, 2, 2, 2), (32, 24, 2, 2, 2), (32, 4, 3, 3, 3)])]
    for dtype in DTYPES:
        caches = [layers.PackedCache() for _ in ws]
        builders = [lambda w=w: hip.pack_conv_k3(w.detach(), dtype, vol=(2, 32, 32, 32)) for w in ws[:3]]
        builders += [lambda w=ws[3]: hip.pack_conv_k1(w.detach().reshape(3, 32), dtype), lambda w=ws[4]: hip.pack_conv_gather(w.detach(), dtype)]
        builders += [lambda w=ws[5]: hip.pack_conv_k3(w.detach(), dtype, vol=(2, 32, 32, 64))]
        # input-gradient (flipped / transposed) images, ConvTranspose images (K = 8 * Cout in runs of K0 = Cout), a gather image
        # with K0 = 4 (several carries inside one 16-byte chunk)
        builders += [lambda w=ws[6]: hip.pack_conv_k3(w.detach(), dtype, dgrad=True, vol=(2, 32, 32, 32)),
                     lambda w=ws[7]: hip.pack_conv_k1(w.detach().reshape(24, 40), dtype, dgrad=True),
                     lambda w=ws[8]: hip.pack_deconv(w.detach(), dtype), lambda w=ws[9]: hip.pack_deconv(w.detach(), dtype, bwd=True),
                     lambda w=ws[10]: hip.pack_conv_gather(w.detach(), dtype)]
        first = [c.get(w, dtype, "f", b).clone() for c, w, b in zip(caches, ws, builders)]
        with torch.no_grad():
            for w in ws:
                w.mul_(-0.5)          # in-place edit: every image is stale now
        got = [c.get(w, dtype, "f", b) for c, w, b in zip(caches, ws, builders)]   # ONE batched refresh on the first get
        for g, f, b in zip(got, first, builders):
            ref = b()
            assert torch.equal(g, ref) and not torch.equal(g, f)


def test_conv3d_k3_kernel_choice():
    from medicalsemseg_amd import hip
    _dev()
    L = hip.lib()
    assert L.msseg_conv3d_k3_kernel(2, 96, 96, 96, 32, 32, hip.BF16) == 3      # ping-pong kernel
    assert L.msseg_conv3d_k3_kernel(2, 96, 96, 96, 32, 32, hip.F32) == 0       # fp32: generic big tile
    assert L.msseg_conv3d_k3_kernel(2, 96, 96, 96, 64, 32, hip.BF16) == 0      # two channel blocks per stage
    assert L.msseg_conv3d_k3_kernel(2, 12, 12, 12, 32, 32, hip.BF16) in (1, 2)  # small grid
    assert L.msseg_conv3d_k3_kernel(2, 96, 96, 96, 48, 48, hip.BF16) == 4      # 48 input channels: 16-wide-block ping-pong
    assert L.msseg_conv3d_k3_kernel(2, 48, 48, 48, 48, 96, hip.BF16) == 4
    assert L.msseg_conv3d_k3_kernel(2, 96, 96, 96, 48, 48, hip.F32) == 0
    assert L.msseg_conv3d_k3_kernel(8, 96, 96, 96, 48, 48, hip.BF16) == 0      # more samples than its statistics slots
    assert L.msseg_conv3d_k3_kernel(2, 12, 12, 12, 48, 48, hip.BF16) in (1, 2)


@pytest.mark.parametrize("tokens,cin,cout", [(432, 1536, 384), (432, 1152, 384), (54, 3072, 768), (433, 1536, 48), (17, 1152, 16), (4096, 1536, 96)])
def test_linear_few_tokens_deep_k(tokens, cin, cout):
    """conv3d_k1 / Linear on few tokens with K = 1152 ... 3072 (last Swin stage): the workgroup's four waves split K
    (linear_ksplit_kernel, csrc/linear_regw.hip); with and without bias, output as a channel slice of a wider buffer"""
    from medicalsemseg_amd import hip
    dev, dtype = _dev(), torch.bfloat16
    x = gen(tokens, cin, seed=1)
    w = gen(cout, cin, seed=2, scale=cin ** -0.5)
    b = gen(cout, seed=3)
    xr, wr = rnd(dtype, x, w)
    ref = F.linear(xr, wr, b)
    wp = hip.pack_conv_k1(w.to(dev), dtype)
    xg = x.to(dev, dtype)
    y = torch.empty(tokens, cout, device=dev, dtype=dtype)
    hip.conv3d_k1(xg, wp, b.to(dev), y, cin, cout)
    check(y, ref, dtype, "few-token linear")
    big = torch.full((tokens, cout + 16), 7.0, device=dev, dtype=dtype)
    hip.conv3d_k1(xg, wp, None, big[:, 8:8 + cout], cin, cout)
    check(big[:, 8:8 + cout], F.linear(xr, wr), dtype, "few-token linear, no bias, slice")
    assert float((big[:, :8].float() - 7).abs().max()) == 0 and float((big[:, 8 + cout:].float() - 7).abs().max()) == 0


@pytest.mark.parametrize("C,tokens", [(48, (2, 9, 11, 13)), (96, (1, 6, 7, 8)), (192, (1, 3, 5, 7)), (384, (1, 3, 3, 3)), (40, (1, 4, 5, 6))])
def test_mlp_fused_gelu_equals_linear_gelu_linear(C, tokens):
    """ops.mlp (GELU in the epilogues of fc1 and of fc2's input gradient, csrc/linear_regw.hip) == ops.linear -> ops.gelu ->
    ops.linear, bit for bit: outputs, input gradient and all four parameter gradients (bf16; C = 40 is a shape the fused
    kernel does not take: the autograd node then runs the three kernels itself)"""
    from medicalsemseg_amd import hip, ops
    dev, dtype = _dev(), torch.bfloat16
    x0 = gen(*tokens, C, seed=1).to(dev).to(dtype)
    dy = gen(*tokens, C, seed=2).to(dev).to(dtype)
    mk = lambda *sh, seed, sc: torch.nn.Parameter((gen(*sh, seed=seed) * sc).to(dev))   # noqa: E731
    ps = [mk(4 * C, C, seed=3, sc=C ** -0.5), mk(4 * C, seed=4, sc=0.5), mk(C, 4 * C, seed=5, sc=(4 * C) ** -0.5), mk(C, seed=6, sc=0.5)]
    assert hip.linear_gelu_ok(x0, C, 4 * C) == (C != 40)
    res = []
    for fused in (True, False):
        for q in ps:
            q.grad = None
        x = x0.clone().requires_grad_(True)
        if fused:
            y = ops.mlp(x, *ps)
        else:
            y = ops.linear(ops.gelu(ops.linear(x, ps[0], ps[1])), ps[2], ps[3])
        y.backward(dy)
        res.append([y.detach().clone(), x.grad.clone()] + [q.grad.clone() for q in ps])
    for a, b, nm in zip(res[0], res[1], ("y", "dx", "dW1", "db1", "dW2", "db2")):
        assert torch.equal(a, b), f"{nm}: max diff {float((a.float() - b.float()).abs().max()):.3e}"
    # and against torch in fp32 on the bf16-rounded operands
    xr = x0.float().cpu().requires_grad_(True)
    w1, b1, w2, b2 = [q.detach().cpu() for q in ps]
    yr = F.linear(F.gelu(F.linear(xr, w1.to(dtype).float(), b1)), w2.to(dtype).float(), b2)
    yr.backward(dy.float().cpu())
    check(res[0][0], yr.detach(), dtype, "mlp y", scale=float(yr.detach().abs().max()))
    check(res[0][1], xr.grad, dtype, "mlp dx", scale=float(xr.grad.abs().max()))


@pytest.mark.parametrize("tokens,cin,cout", [(4099, 48, 144), (700, 48, 192), (1000, 192, 48), (333, 48, 48), (129, 96, 288), (257, 384, 96),
                                             (50, 144, 48), (1100, 384, 1536), (64, 96, 48), (3001, 96, 96)])
def test_linear_wgrad_one_pass_weight_and_bias(tokens, cin, cout):
    """msseg_linear_wgrad (csrc/linear_wgrad.hip: weight and bias gradient of nn.Linear from one pass over the tokens) vs
    torch on the bf16-rounded operands, write and accumulate forms, deterministic"""
    from medicalsemseg_amd import hip
    dev, dtype = _dev(), torch.bfloat16
    x = gen(tokens, cin, seed=1).to(dev).to(dtype)
    dy = gen(tokens, cout, seed=2).to(dev).to(dtype)
    assert hip.linear_wgrad_ok(x, cin, cout)
    dw = torch.full((cout, cin), float("nan"), device=dev)
    db = torch.full((cout,), float("nan"), device=dev)
    hip.linear_wgrad(x, dy, dw, db, cin, cout)
    ref_w = dy.double().t() @ x.double()
    ref_b = dy.double().sum(0)
    sw, sb = float(ref_w.abs().max()), float(ref_b.abs().max())
    assert float((dw.double() - ref_w).abs().max()) / sw < 2e-5
    assert float((db.double() - ref_b).abs().max()) / sb < 2e-5
    dw2, db2 = dw.clone(), db.clone()
    hip.linear_wgrad(x, dy, dw2, db2, cin, cout, True, True)
    assert float((dw2.double() - 2 * ref_w).abs().max()) / sw < 4e-5 and float((db2.double() - 2 * ref_b).abs().max()) / sb < 4e-5
    dw3 = torch.empty_like(dw)
    hip.linear_wgrad(x, dy, dw3, None, cin, cout)            # weight only
    assert torch.equal(dw3, dw)
    # channel slices of wider buffers (qkv gradient slices, concat halves)
    big_x = torch.zeros(tokens, cin + 16, device=dev, dtype=dtype); big_x[:, 8:8 + cin] = x
    big_dy = torch.zeros(tokens, cout + 8, device=dev, dtype=dtype); big_dy[:, :cout] = dy
    dw4, db4 = torch.empty_like(dw), torch.empty_like(db)
    hip.linear_wgrad(big_x[:, 8:8 + cin], big_dy[:, :cout], dw4, db4, cin, cout)
    assert torch.equal(dw4, dw) and torch.equal(db4, db)


def test_flat_adamw_matches_torch_adamw():
    """fused flat-buffer AdamW (+ weight-decay grouping, + folded gradient clipping) vs torch.optim.AdamW"""
    from medicalsemseg_amd.optim import FlatAdamW, add_weight_decay
    dev = _dev()
    torch.manual_seed(0)
    mk = lambda: torch.nn.Sequential(torch.nn.Conv3d(2, 4, 3), torch.nn.InstanceNorm3d(4, affine=True), torch.nn.Conv3d(4, 3, 1))
    a, b = mk().to(dev), mk().to(dev)
    b.load_state_dict(a.state_dict())
    oa = FlatAdamW(add_weight_decay(a, 0.05), lr=1e-2, betas=(0.9, 0.95), eps=1e-6)
    ob = torch.optim.AdamW(add_weight_decay(b, 0.05), lr=1e-2, betas=(0.9, 0.95), eps=1e-6)
    for it in range(5):
        gs = [torch.randn_like(p) * (3.0 if it == 2 else 0.1) for p in a.parameters()]
        for p, g in zip(a.parameters(), gs):
            p.grad.copy_(g)
        for p, g in zip(b.parameters(), gs):
            p.grad = g.clone()
        na = oa.clip_grad_norm_(1.0)
        nb = torch.nn.utils.clip_grad_norm_(b.parameters(), 1.0)
        assert abs(float(na) - float(nb)) < 1e-4 * float(nb)
        if it == 3:
            for g_ in oa.param_groups + ob.param_groups:
                g_["lr"] = 5e-3
        oa.step(); ob.step()
        oa.zero_grad(); ob.zero_grad()
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert float((p - q).abs().max()) < 2e-6, n


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,sp,N", [(32, 32, (32, 32, 32), 2), (64, 32, (12, 12, 12), 2), (16, 48, (6, 6, 6), 1),
                                           (32, 32, (32, 48, 64), 2), (64, 32, (30, 29, 38), 3),
                                           # 48 gradient channels in: the 48-channel ping-pong kernel (bf16)
                                           (48, 48, (32, 32, 64), 2), (96, 48, (30, 33, 36), 2)])
def test_conv_dgrad_fused_instnorm_backward_reductions(dtype, cin, cout, sp, N):
    """da = dgrad(dy) with the InstanceNorm-backward reductions of the receiving layer fused in the epilogue ==
    separate dgrad + msseg_instnorm_act_bwd_reduce"""
    from medicalsemseg_amd import hip
    from medicalsemseg_amd.layers import Conv3, InstNormAct
    dev = _dev()
    # layer L: yraw_L [cin ch] -> act_L ; layer L+1: conv cin -> cout ; dy = grad of conv output
    yraw = cl(gen(N, cin, *sp, seed=1) * 1.5 + 0.3, dtype, dev)
    w = torch.nn.Parameter(gen(cout, cin, 3, 3, 3, seed=2, scale=(cin * 27) ** -0.5).to(dev))
    dy = cl(gen(N, cout, *sp, seed=3), dtype, dev)
    ga = torch.nn.Parameter((gen(cin, seed=4) * 0.2 + 1).to(dev))
    be = torch.nn.Parameter((gen(cin, seed=5) * 0.2).to(dev))
    nrm = InstNormAct(ga, be, 0.1)
    act, stats = nrm.fwd(yraw)
    conv = Conv3(w, None)
    w.requires_grad_(False)
    # separate path
    da_ref = conv.bwd(act, dy, True)
    dyraw_ref = nrm.bwd(yraw, stats, act, da_ref)
    g_ref, b_ref = ga.grad.clone(), be.grad.clone()
    ga.grad = be.grad = None
    # fused path
    from medicalsemseg_amd.layers import APPLIED
    da, red = conv.bwd(act, dy, True, next_norm=(nrm, yraw, stats, act))
    assert red is not None
    dyraw = nrm.bwd(yraw, stats, act, da, red=red)
    if red is APPLIED:      # small-grid path: the finish kernel ran the receiving unit's whole backward, `da` IS its dy
        assert dyraw is da
    else:
        assert torch.equal(da, da_ref)
    tol = 2e-4 if dtype == torch.float32 else 2e-3
    sc = float(dyraw_ref.float().abs().max())
    assert float((dyraw.float() - dyraw_ref.float()).abs().max()) / sc < tol
    for a, b, nm in ((ga.grad, g_ref, "dgamma"), (be.grad, b_ref, "dbeta")):
        assert float((a - b).abs().max()) / (float(b.abs().max()) + 1e-6) < 5e-4, nm


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,C", [((2, 8, 12, 16), 32), ((1, 6, 4, 10), 64), ((1, 2, 2, 2), 8)])
def test_instnorm_act_pool_fwd_equals_two_kernels(dtype, shape, C):
    """InstanceNorm + LeakyReLU + MaxPool3d(2) in one pass == instnorm_act_fwd followed by maxpool2_fwd, bit for bit
    (activation written into the first half of a [skip | up] concat buffer, as the UNet encoder does)"""
    from medicalsemseg_amd import hip
    DEV = _dev()
    N, D, H, W = shape
    torch.manual_seed(11)
    y = torch.randn(N, D, H, W, C, device=DEV).to(dtype)
    gamma = torch.randn(C, device=DEV)
    beta = torch.randn(C, device=DEV)
    stats = hip.channel_stats(y)
    cat_a = torch.zeros(N, D, H, W, 2 * C, device=DEV, dtype=dtype)
    cat_b = torch.zeros_like(cat_a)
    pool_a = torch.empty(N, D // 2, H // 2, W // 2, C, device=DEV, dtype=dtype)
    pool_b = torch.empty_like(pool_a)
    assert hip.instnorm_pool_ok(y, cat_a[..., :C], pool_a)
    hip.instnorm_act_pool_fwd(y, stats, gamma, beta, cat_a[..., :C], pool_a, 0.1)
    hip.instnorm_act_fwd(y, stats, gamma, beta, cat_b[..., :C], 0.1)
    hip.maxpool2_fwd(cat_b[..., :C], pool_b)
    assert torch.equal(cat_a, cat_b) and torch.equal(pool_a, pool_b)
    ref = torch.nn.functional.max_pool3d(cat_b[..., :C].float().permute(0, 4, 1, 2, 3), 2).permute(0, 2, 3, 4, 1)
    assert torch.equal(pool_a.float(), ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,C", [((2, 8, 12, 16), 32), ((1, 6, 4, 10), 64), ((1, 2, 2, 2), 8), ((1, 16, 16, 16), 256)])
def test_instnorm_act_poolbwd_reduce_equals_two_kernels(dtype, shape, C):
    """skip + maxpool-backward and the InstanceNorm-backward sums in one pass == maxpool2_bwd(accumulate) followed by
    instnorm_act_bwd_reduce: the gradient tensor bit for bit (skip gradient read from the first half of a concat-shaped
    buffer, as in the UNet), the sums to fp32 summation-order rounding"""
    from medicalsemseg_amd import hip
    DEV = _dev()
    N, D, H, W = shape
    torch.manual_seed(12)
    y = torch.randn(N, D, H, W, C, device=DEV).to(dtype)
    gamma = torch.randn(C, device=DEV)
    beta = torch.randn(C, device=DEV) * 0.3
    stats = hip.channel_stats(y)
    a = torch.empty_like(y)
    pooled = torch.empty(N, D // 2, H // 2, W // 2, C, device=DEV, dtype=dtype)
    hip.instnorm_act_pool_fwd(y, stats, gamma, beta, a, pooled, 0.1)
    dcat = torch.randn(N, D, H, W, 2 * C, device=DEV).to(dtype)
    g = torch.randn_like(pooled)
    # reference: two kernels, in place in the concat-shaped gradient
    ref_cat = dcat.clone()
    hip.maxpool2_bwd(a, g, ref_cat[..., :C], accumulate=True)
    dg_ref, db_ref = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx_ref = torch.empty_like(y)
    red_ref = hip.instnorm_act_bwd(y, stats, gamma, None, ref_cat[..., :C], dx_ref, 0.1, 1e-5, None, dg_ref, db_ref, False, beta)
    # fused
    da = torch.empty_like(y)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    red = hip.instnorm_act_poolbwd_reduce(y, stats, gamma, beta, dcat[..., :C], g, da, 0.1, 1e-5, dg, db, False)
    assert torch.equal(da, ref_cat[..., :C].contiguous())
    scale = float(red_ref.abs().max())
    assert float((red - red_ref).abs().max()) < 1e-5 * scale + 1e-6
    assert float((dg - dg_ref).abs().max()) < 1e-5 * float(dg_ref.abs().max()) + 1e-6
    assert float((db - db_ref).abs().max()) < 1e-5 * float(db_ref.abs().max()) + 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_and_deconv_vs_reference_unetr_blocks(dtype):
    """HIP conv3d k3 / ConvTranspose3d k2 s2 (forward, input, weight and bias gradients) against outputs of the REFERENCE's
    own UNETR decoder blocks (tests/golden/unetr_blocks.npz, made by oracle/gen_golden.py from
    /root/reference/models/segmentors/unetr.py:9-52); fp32 gate 1e-4 of the output scale, bf16 reported to 2e-2."""
    import os
    from medicalsemseg_amd import layers
    from tests.test_oracle_golden import _unetr_params
    from tests.golden_util import det_tensor
    DEV = _dev()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "unetr_blocks.npz"))
    P = {k: torch.nn.Parameter(v.to(DEV)) for k, v in _unetr_params().items()}
    tol = 1e-4 if dtype == torch.float32 else 2e-2

    def rel(got, key):
        ref = torch.from_numpy(g[key])
        return float((got.detach().float().cpu() - ref).abs().max()) / float(ref.abs().max())

    x = cl(det_tensor("unetr_x", (2, 16, 12, 12, 12)), dtype, DEV)
    r = cl(det_tensor("unetr_r", (2, 32, 12, 12, 12)), dtype, DEV)
    conv = layers.Conv3(P["conv_w"], P["conv_b"])
    y = conv.fwd(x)
    dx = conv.bwd(x, r, need_dx=True)
    assert rel(ncdhw(y), "conv_y") < tol and rel(ncdhw(dx), "conv_dx") < tol
    assert rel(P["conv_w"].grad, "conv_dw") < tol and rel(P["conv_b"].grad, "conv_db") < tol
    x2 = cl(det_tensor("unetr_x2", (2, 32, 6, 6, 6)), dtype, DEV)
    r2 = cl(det_tensor("unetr_r2", (2, 16, 12, 12, 12)), dtype, DEV)
    dec = layers.Deconv2(P["deconv_w"], P["deconv_b"])
    y2 = dec.fwd(x2)
    dx2 = dec.bwd(x2, r2, need_dx=True)
    assert rel(ncdhw(y2), "deconv_y") < tol and rel(ncdhw(dx2), "deconv_dx") < tol
    assert rel(P["deconv_w"].grad, "deconv_dw") < tol and rel(P["deconv_b"].grad, "deconv_db") < tol
    # Deconv3DBlock in eval mode: deconv -> conv on the HIP kernels, BatchNorm (initial running statistics) + ReLU as torch ops
    t = layers.Conv3(P["blk_cw"], P["blk_cb"]).fwd(layers.Deconv2(P["blk_dw"], P["blk_db"]).fwd(x2)).float()
    t = torch.relu(t / np.sqrt(1.0 + 1e-5) * P["blk_bn_w"] + P["blk_bn_b"])
    assert rel(ncdhw(t), "block_y") < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout", [(32, 3), (48, 3), (8, 1), (64, 4), (32, 2)])
def test_conv3d_k1_head(dtype, cin, cout):
    """segmentation-head kernel (1x1x1 conv, <= 4 classes, fp32 weight as it is) vs F.conv3d on the rounded operands;
    output written into the first channels of 8-channel rows, as the UNet does with its logits buffer"""
    from medicalsemseg_amd import hip
    DEV = _dev()
    x = gen(2, cin, 6, 10, 12, seed=31)
    w = gen(cout, cin, 1, 1, 1, seed=32, scale=0.2)
    b = gen(cout, seed=33)
    xr, wr = rnd(dtype, x, w)
    ref = F.conv3d(xr, wr, b)
    xc = cl(x, dtype, DEV)
    buf = torch.full((2, 6, 10, 12, 8), 7.0, device=DEV, dtype=dtype)
    hip.conv3d_k1_head(xc, w.to(DEV).reshape(cout, cin), b.to(DEV), buf[..., :cout], cin, cout)
    check(ncdhw(buf[..., :cout]), ref, dtype, "conv3d_k1_head")
    assert bool((buf[..., cout:] == 7.0).all())          # the padding channels are not touched


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C,affine", [((2, 3, 3, 3), 3072, True), ((2, 3, 3, 3), 1536, True), ((1, 5, 6, 7), 384, True),
                                            ((2, 6, 7, 8), 48, False), ((1, 4, 5, 6), 40, True), ((1, 2, 2, 3), 4104, True),
                                            # many rows per lane: the parameter-gradient partial rows of the backward kernel
                                            ((2, 24, 24, 24), 48, True), ((1, 12, 12, 12), 96, True), ((1, 6, 6, 6), 192, True),
                                            ((1, 5, 6, 7), 768, True), ((1, 3, 3, 3), 1024, True)])
def test_layer_norm_wide_and_narrow_rows(dtype, rows, C, affine):
    """ops.layer_norm forward / backward vs torch on the rounded operands: the vector kernel up to 8 chunks per lane (the
    3072-wide LayerNorm of the MONAI variant's last patch merging: /root/reference/models/segmentors/swin_unetr_official.py:699-708),
    un-affine (proj_out, :955-968), and the scalar fall-backs (40 channels: no 16-byte chunks in bf16; 4104: beyond 8 x 64 chunks)"""
    from medicalsemseg_amd import ops
    dev = _dev()
    x = gen(*rows, C, seed=61)
    g, b = gen(C, seed=62) * 0.3 + 1.0, gen(C, seed=63) * 0.3
    dy = gen(*rows, C, seed=64)
    xr, dyr = rnd(dtype, x, dy)
    xr.requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yref = F.layer_norm(xr, (C,), gr if affine else None, br if affine else None, 1e-5)
    yref.backward(dyr)
    xg = x.detach().clone().to(dev, dtype).requires_grad_(True)     # (rnd() aliases x in fp32)
    gp, bp = torch.nn.Parameter(g.to(dev)), torch.nn.Parameter(b.to(dev))
    y = ops.layer_norm(xg, gp if affine else None, bp if affine else None, 1e-5)
    y.backward(dy.to(dev, dtype))
    check(y.detach(), yref.detach(), dtype, "layer_norm fwd")
    check(xg.grad, xr.grad, dtype, "layer_norm dx")
    if affine:
        check(gp.grad, gr.grad, dtype, "layer_norm dgamma")
        check(bp.grad, br.grad, dtype, "layer_norm dbeta")


@pytest.mark.parametrize("cin,cout,tokens", [
    (48, 144, (2, 24, 24, 32)), (48, 48, (1, 25, 27, 29)), (48, 192, (1, 25, 27, 29)), (192, 48, (2, 24, 24, 32)),
    (144, 48, (1, 25, 27, 29)), (40, 48, (1, 25, 27, 29)),
    # deeper stages: output channels sliced over grid.y, few tokens (one 16-token group per wave or less)
    (96, 288, (2, 12, 12, 13)), (384, 96, (2, 12, 12, 13)), (192, 576, (2, 6, 6, 7)), (768, 192, (2, 6, 6, 7)),
    (384, 1152, (1, 3, 3, 5)), (576, 192, (2, 6, 6, 7)), (96, 96, (1, 1, 1, 7))])
def test_linear_many_tokens_register_weight_kernel(cin, cout, tokens):
    """nn.Linear on >= 16 k tokens with the first Swin stage's widths (bf16): `linear_regw_kernel` (weights in registers,
    operand straight from global, LDS-transposed coalesced rows) for the forward and, through the transposed weight image,
    the input gradient; a token count that is no multiple of 16, a partial last 32-channel step (48, 144, 40 channels),
    output rows inside a wider buffer; vs F.linear on the rounded operands (fp32 accumulation)"""
    from medicalsemseg_amd import hip, ops
    dev, dtype = _dev(), torch.bfloat16
    x = gen(*tokens, cin, seed=51)
    w = gen(cout, cin, seed=52, scale=cin ** -0.5)
    b = gen(cout, seed=53)
    xr, wr = rnd(dtype, x, w)
    xr.requires_grad_(True); wr.requires_grad_(True)
    yref = F.linear(xr, wr, b)
    dy = gen(*yref.shape, seed=54)
    dyr = rnd(dtype, dy)
    yref.backward(dyr)
    xg = x.to(dev, dtype).requires_grad_(True)
    wp, bp = torch.nn.Parameter(w.to(dev)), torch.nn.Parameter(b.to(dev))
    y = ops.linear(xg, wp, bp)
    y.backward(dy.to(dev, dtype))
    check(y.detach(), yref.detach(), dtype, "linear fwd")
    check(xg.grad, xr.grad, dtype, "linear dgrad")
    check(wp.grad, wr.grad, dtype, "linear wgrad")
    # rows inside a wider buffer (ld > channels), no bias
    wide = torch.full(tokens + (cout + 16,), 3.0, dtype=dtype, device=dev)
    wpk = hip.pack_conv_k1(w.to(dev), dtype)
    hip.conv3d_k1(xg.detach(), wpk, None, wide[..., :cout], cin, cout)
    check(wide[..., :cout], F.linear(xr.detach(), wr.detach()), dtype, "linear fwd into a view")
    assert bool((wide[..., cout:] == 3.0).all())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout", [(32, 3), (48, 2), (64, 4)])
def test_conv3d_k1_head_dgrad_inbwd(dtype, cin, cout):
    """streaming head input-gradient + InstanceNorm-backward sums vs the torch fp32 formulas on the rounded operands"""
    from medicalsemseg_amd import hip
    DEV = _dev()
    N, sp = 2, (6, 10, 12)
    S = sp[0] * sp[1] * sp[2]
    yraw = gen(N, cin, *sp, seed=41)
    dl = gen(N, cout, *sp, seed=42)
    w = gen(cout, cin, seed=43, scale=0.2)
    gamma, beta = gen(cin, seed=44) * 0.3 + 1.0, gen(cin, seed=45) * 0.3
    yr, dlr, wr = rnd(dtype, yraw, dl, w)
    # reference (fp32, NCDHW): da = dl . w, rounded as stored; z from the stored raw output
    da = torch.einsum("nkdhw,kc->ncdhw", dlr, wr)
    da = rnd(dtype, da)
    mean = yr.mean(dim=(2, 3, 4), keepdim=True)
    var = (yr * yr).mean(dim=(2, 3, 4), keepdim=True) - mean * mean
    rstd = torch.rsqrt(var.clamp_min(0) + 1e-5)
    z = (yr - mean) * rstd * gamma.view(1, -1, 1, 1, 1) + beta.view(1, -1, 1, 1, 1)
    dz = torch.where(z > 0, da, da * 0.1)
    red_ref = torch.stack([dz.sum(dim=(2, 3, 4)), (dz * (yr - mean) * rstd).sum(dim=(2, 3, 4))], dim=-1)
    # HIP
    yc = cl(yraw, dtype, DEV)
    dlc = torch.zeros(N, *sp, 8, device=DEV, dtype=dtype)
    dlc[..., :cout] = cl(dl, dtype, DEV)
    stats = hip.channel_stats(yc)
    dx = torch.empty_like(yc)
    dg, db = torch.zeros(cin, device=DEV), torch.zeros(cin, device=DEV)
    red = hip.conv3d_k1_head_dgrad_inbwd(dlc, w.to(DEV), dx, cin, cout, yc, stats, gamma.to(DEV), beta.to(DEV), 0.1, 1e-5,
                                         dg, db, False)
    check(ncdhw(dx), da, dtype, "head dgrad")
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    sc = float(red_ref.abs().max())
    assert float((red.cpu() - red_ref).abs().max()) / sc < tol
    assert float((db.cpu() - red_ref[..., 0].sum(0)).abs().max()) / sc < tol
    assert float((dg.cpu() - red_ref[..., 1].sum(0)).abs().max()) / sc < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,affine", [(32, 3, True), (48, 2, False), (64, 4, True)])
def test_conv3d_k1_head_norm_fused_fwd_bwd(dtype, cin, cout, affine):
    """head fused with the InstanceNorm + LeakyReLU in front of it (normalise on load; backward recomputes the activation
    and also produces the head's weight gradient) == the unfused kernels on the same operands"""
    from medicalsemseg_amd import hip
    DEV = _dev()
    N, sp = 2, (6, 10, 12)
    yraw = gen(N, cin, *sp, seed=51)
    dl = gen(N, cout, *sp, seed=52)
    w = gen(cout, cin, seed=53, scale=0.2).to(DEV)
    b = gen(cout, seed=56).to(DEV)
    gamma = (gen(cin, seed=54) * 0.3 + 1.0).to(DEV) if affine else None
    beta = (gen(cin, seed=55) * 0.3).to(DEV) if affine else None
    yc = cl(yraw, dtype, DEV)
    stats = hip.channel_stats(yc)
    # unfused: normalise (stores the activation), head on the activation
    act = torch.empty_like(yc)
    hip.instnorm_act_fwd(yc, stats, gamma, beta, act, 0.1, 1e-5)
    ref = torch.zeros(N, *sp, 8, device=DEV, dtype=dtype)
    hip.conv3d_k1_head(act, w, b, ref[..., :cout], cin, cout)
    got = torch.zeros(N, *sp, 8, device=DEV, dtype=dtype)
    hip.conv3d_k1_head_norm(yc, stats, gamma, beta, 0.1, 1e-5, w, b, got[..., :cout], cin, cout)
    check(got[..., :cout], ref[..., :cout], dtype, "fused head forward", scale=float(ref.float().abs().max()))
    # backward: da / sums as the unfused kernel, dw against einsum on the stored activation
    dlc = torch.zeros(N, *sp, 8, device=DEV, dtype=dtype)
    dlc[..., :cout] = cl(dl, dtype, DEV)
    dx0, dx1 = torch.empty_like(yc), torch.empty_like(yc)
    dg0, db0 = (torch.zeros(cin, device=DEV), torch.zeros(cin, device=DEV)) if affine else (None, None)
    dg1, db1 = (torch.zeros(cin, device=DEV), torch.zeros(cin, device=DEV)) if affine else (None, None)
    red0 = hip.conv3d_k1_head_dgrad_inbwd(dlc, w, dx0, cin, cout, yc, stats, gamma, beta, 0.1, 1e-5, dg0, db0, False)
    dw = torch.full((cout, cin), 0.25, device=DEV)
    red1 = hip.conv3d_k1_head_bwd_fused(dlc, w, dx1, cin, cout, yc, stats, gamma, beta, 0.1, 1e-5, dw, True, dg1, db1, False)
    if dtype == torch.bfloat16:
        assert torch.equal(dx0, dx1)
    else:   # two instantiations of the same arithmetic: the compiler may contract the fp32 dot products differently
        assert torch.allclose(dx0, dx1, rtol=1e-5, atol=1e-6)
    assert torch.allclose(red0, red1, rtol=1e-4, atol=1e-5 * float(red0.abs().max()))
    if affine:
        assert torch.allclose(dg0, dg1, rtol=1e-5, atol=1e-5 * float(dg0.abs().max()))
    dw_ref = torch.einsum("ndhwk,ndhwc->kc", dlc[..., :cout].float(), act.float()) + 0.25
    tol = 2e-4 if dtype == torch.float32 else 2e-3
    assert float((dw - dw_ref).abs().max()) / float(dw_ref.abs().max()) < tol


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,sp,N", [(32, 32, (6, 7, 9), 2), (64, 32, (4, 5, 18), 3), (48, 48, (4, 4, 6), 2)])
def test_deconv_bwd_fused_sums_and_bias(dtype, cin, cout, sp, N):
    """transposed-conv input gradient with the receiving layer's InstanceNorm-backward sums and the bias gradient from the
    same pass == input gradient + separate reduction passes (bf16 32/64 -> 32: the register-resident-weight kernel)"""
    from medicalsemseg_amd import hip
    from medicalsemseg_amd.layers import Deconv2, InstNormAct
    dev = _dev()
    yraw = cl(gen(N, cin, *sp, seed=1) * 1.5 + 0.3, dtype, dev)
    w = torch.nn.Parameter(gen(cin, cout, 2, 2, 2, seed=2, scale=cin ** -0.5).to(dev))
    b = torch.nn.Parameter(gen(cout, seed=6).to(dev))
    fine = tuple(2 * v for v in sp)
    dy = cl(gen(N, cout, *fine, seed=3), dtype, dev)
    ga = torch.nn.Parameter((gen(cin, seed=4) * 0.2 + 1).to(dev))
    be = torch.nn.Parameter((gen(cin, seed=5) * 0.2).to(dev))
    nrm = InstNormAct(ga, be, 0.1)
    act, stats = nrm.fwd(yraw)
    op = Deconv2(w, b)
    w.requires_grad_(False)
    # reference: plain input gradient, separate InstanceNorm-backward reduce, separate channel sum
    wp = hip.pack_deconv(w.detach(), dtype, bwd=True)
    dx_ref = torch.empty_like(act)
    hip.deconv_k2s2_bwd_data(dy, wp, dx_ref, cin, cout)
    dyraw_ref = nrm.bwd(yraw, stats, act, dx_ref)
    g_ref, b_ref = ga.grad.clone(), be.grad.clone()
    ga.grad = be.grad = None
    db_ref = dy.float().sum(dim=(0, 1, 2, 3))
    # fused
    dx, red = op.bwd(act, dy, True, next_norm=(nrm, yraw, stats, act))
    assert red is not None
    dyraw = nrm.bwd(yraw, stats, act, dx, red=red)
    assert torch.equal(dx, dx_ref)
    tol = 2e-4 if dtype == torch.float32 else 2e-3
    sc = float(dyraw_ref.float().abs().max())
    assert float((dyraw.float() - dyraw_ref.float()).abs().max()) / sc < tol
    for a, bb, nm in ((ga.grad, g_ref, "dgamma"), (be.grad, b_ref, "dbeta"), (b.grad, db_ref, "dbias")):
        assert float((a - bb).abs().max()) / (float(bb.abs().max()) + 1e-6) < 5e-4, nm


def test_deconv_wgrad_large_grid_one_pass():
    """Swin-UNETR's 48 -> 48 transposed conv on a grid of > 100 k coarse voxels: the weight gradient takes the one-pass kernel
    with the child-row gather (csrc/linear_wgrad.hip, DCV) -- against the fp32 contraction of the same bf16 operands"""
    from medicalsemseg_amd import hip
    dev = _dev()
    cin = cout = 48
    N, sp = 2, (40, 41, 32)
    x = (torch.randn(N, *sp, cin, device=dev) * 0.5).to(torch.bfloat16)
    dy = (torch.randn(N, *(2 * v for v in sp), cout, device=dev) * 0.5).to(torch.bfloat16)
    dw = torch.full((cin, cout, 2, 2, 2), float("nan"), device=dev)
    hip.deconv_k2s2_wgrad(x, dy, dw, cin, cout)
    dyv = dy.float().view(N, sp[0], 2, sp[1], 2, sp[2], 2, cout)
    ref = torch.einsum("ndhwi,ndahbwco->ioabc", x.float(), dyv)
    assert float((dw - ref).abs().max()) / float(ref.abs().max()) < 2e-4
    dw2 = dw.clone()
    hip.deconv_k2s2_wgrad(x, dy, dw2, cin, cout, True)
    assert float((dw2 - 2 * ref).abs().max()) / float(ref.abs().max()) < 4e-4


@pytest.mark.parametrize("cin,cout,sp,N", [(256, 128, (3, 6, 6), 2), (128, 64, (6, 6, 12), 2), (128, 64, (4, 5, 7), 3)])
def test_deconv_bwd_small_unit(cin, cout, sp, N):
    """deep UpCat levels (bf16): the transposed conv's input gradient as one fp32 block (csrc/deconv_k2s2_gen.hip) + the small-grid
    finish kernel that runs the receiving conv + InstanceNorm + LeakyReLU unit's whole backward == input gradient, separate
    InstanceNorm backward, separate channel sum"""
    from medicalsemseg_amd import hip
    from medicalsemseg_amd.layers import APPLIED, Deconv2, InstNormAct
    dev = _dev()
    dtype = torch.bfloat16
    yraw = cl(gen(N, cin, *sp, seed=1) * 1.5 + 0.3, dtype, dev)
    w = torch.nn.Parameter(gen(cin, cout, 2, 2, 2, seed=2, scale=cin ** -0.5).to(dev))
    b = torch.nn.Parameter(gen(cout, seed=6).to(dev))
    fine = tuple(2 * v for v in sp)
    dy = cl(gen(N, cout, *fine, seed=3), dtype, dev)
    ga = torch.nn.Parameter((gen(cin, seed=4) * 0.2 + 1).to(dev))
    be = torch.nn.Parameter((gen(cin, seed=5) * 0.2).to(dev))
    nrm = InstNormAct(ga, be, 0.1)
    act, stats = nrm.fwd(yraw)
    op = Deconv2(w, b)
    w.requires_grad_(False)
    assert hip.deconv_k2s2_small_unit_ok(tuple(act.shape), cin, cout, dtype)
    wp = hip.pack_deconv(w.detach(), dtype, bwd=True)
    dx_ref = torch.empty_like(act)
    hip.deconv_k2s2_bwd_data(dy, wp, dx_ref, cin, cout)
    # the sliced-output kernel against plain torch (bf16-rounded operands, fp32 accumulation)
    xr = torch.zeros(N, cin, *sp, requires_grad=True)
    F.conv_transpose3d(xr, w.detach().cpu().to(dtype).float(), None, stride=2).backward(ncdhw(dy).cpu().float())
    assert float((ncdhw(dx_ref).cpu().float() - xr.grad).abs().max()) / float(xr.grad.abs().max()) < 1e-2
    dyraw_ref = nrm.bwd(yraw, stats, act, dx_ref)
    g_ref, b_ref = ga.grad.clone(), be.grad.clone()
    ga.grad = be.grad = None
    db_ref = dy.float().sum(dim=(0, 1, 2, 3))
    dyraw, red = op.bwd(act, dy, True, next_norm=(nrm, yraw, stats, act))
    assert red is APPLIED
    sc = float(dyraw_ref.float().abs().max())
    assert float((dyraw.float() - dyraw_ref.float()).abs().max()) / sc < 2e-2     # one bf16 rounding of da more / less
    rel = float((dyraw.float() - dyraw_ref.float()).norm() / dyraw_ref.float().norm())
    assert rel < 4e-3, rel
    for a_, bb, nm in ((ga.grad, g_ref, "dgamma"), (be.grad, b_ref, "dbeta"), (b.grad, db_ref, "dbias")):
        assert float((a_ - bb).abs().max()) / (float(bb.abs().max()) + 1e-6) < 2e-3, nm


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp,C", [((6, 6, 6), 24), ((5, 7, 9), 16), ((13, 13, 13), 48)])
def test_layout_kernels_pad_crop_merge_gather(dtype, sp, C):
    """csrc/layout.hip against the torch ops the vendored MONAI encoder uses (F.pad / crop of the window partition, the eight
    strided slices + concat of PatchMerging with its duplicated sub-grids, odd sizes) -- forward and adjoint, bit for bit (the
    adjoint of the duplicated sub-grids adds two values: one rounding, as torch's)"""
    from medicalsemseg_amd import ops
    dev = _dev()
    N = 2
    x = (torch.randn(N, *sp, C, generator=torch.Generator().manual_seed(1))).to(dev, dtype).requires_grad_(True)
    big = tuple(v + p for v, p in zip(sp, (1, 2, 3)))
    # pad, then crop back
    y = ops.box_resize(x, big)
    ref = F.pad(x.detach(), (0, 0, 0, 3, 0, 2, 0, 1))
    assert torch.equal(y, ref)
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(2)).to(dev, dtype)
    y.backward(g)
    assert torch.equal(x.grad, g[:, :sp[0], :sp[1], :sp[2]])
    x.grad = None
    z = ops.box_resize(y.detach().requires_grad_(True), sp)
    assert torch.equal(z, x.detach())
    # PatchMerging gather
    sub = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 0), (0, 0, 1), (1, 1, 1)]
    m = ops.merge_gather(x, sub)
    xr = x.detach().clone().requires_grad_(True)
    xp = F.pad(xr, (0, 0, 0, sp[2] % 2, 0, sp[1] % 2, 0, sp[0] % 2))
    mref = torch.cat([xp[:, a::2, b::2, c::2, :] for a, b, c in sub], -1)
    assert torch.equal(m, mref.detach())
    gm = torch.randn(m.shape, generator=torch.Generator().manual_seed(3)).to(dev, dtype)
    m.backward(gm)
    mref.backward(gm)
    if dtype == torch.float32:
        assert torch.equal(x.grad, xr.grad)
    else:   # torch adds the duplicated sub-grids' gradients in bf16 (two roundings where three terms meet: none here, two terms)
        assert float((x.grad.float() - xr.grad.float()).abs().max()) <= 2 ** -7 * float(xr.grad.float().abs().max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,rows", [(48, 1000), (384, 77), (3072, 9)])
def test_layer_norm_res_one_node_equals_two_consumers(dtype, C, rows):
    """ops.layer_norm_res (x handed through the LayerNorm node, the residual's gradient added inside the LayerNorm backward kernel)
    == x feeding ops.layer_norm and the residual separately (autograd's own add of the two gradients), bit for bit"""
    from medicalsemseg_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x0 = (torch.randn(2, rows, 1, 1, C, generator=g) * 2 + 0.5).to(dev, dtype)
    w = torch.nn.Parameter((torch.randn(C, generator=g) * 0.2 + 1).to(dev))
    b = torch.nn.Parameter((torch.randn(C, generator=g) * 0.2).to(dev))
    gy, gr = (torch.randn(x0.shape, generator=g).to(dev, dtype) for _ in range(2))
    res = []
    for fused in (True, False):
        x = x0.clone().requires_grad_(True)
        w.grad = b.grad = None
        if fused:
            xr, y = ops.layer_norm_res(x, w, b, 1e-5)
        else:
            xr, y = x, ops.layer_norm(x, w, b, 1e-5)
        torch.autograd.backward([y, xr * 1.0 if not fused else xr], [gy, gr])
        res.append((y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone()))
    for a_, b_ in zip(res[0], res[1]):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("cout,sp,N", [(32, (16, 20, 48), 2), (48, (9, 14, 35), 3), (32, (96, 96, 96), 2)])
def test_stem_unit_inference_two_launches(cout, sp, N):
    """inference form of the one-channel conv + InstanceNorm + LeakyReLU unit (statistics-only launch, then the conv again with the
    normalisation in its epilogue: no raw output) against the training form (conv with fused statistics, normalisation pass)"""
    from medicalsemseg_amd import layers
    dev, dtype = _dev(), torch.bfloat16
    x = cl(gen(N, 1, *sp, seed=1) * 2 + 0.3, dtype, dev)
    w = torch.nn.Parameter(gen(cout, 1, 3, 3, 3, seed=2, scale=27 ** -0.5).to(dev))
    b = torch.nn.Parameter(gen(cout, seed=3).to(dev))
    ga = torch.nn.Parameter((gen(cout, seed=4) * 0.2 + 1).to(dev))
    be = torch.nn.Parameter((gen(cout, seed=5) * 0.2).to(dev))
    unit = layers.ConvNormAct(layers.Conv3(w, b), layers.InstNormAct(ga, be, 0.1))
    a_ref, saved = unit.fwd(x)
    assert saved[1] is not None                       # the training form keeps the raw output
    layers.INFERENCE_FORWARD = True
    try:
        with torch.no_grad():
            a_inf, saved_inf = unit.fwd(x)
    finally:
        layers.INFERENCE_FORWARD = False
    assert saved_inf[1] is None                       # no raw output
    assert torch.equal(saved_inf[2], saved[2])        # the same statistics
    d = (a_inf.float() - a_ref.float()).abs()
    # the same arithmetic on the same bf16-rounded conv outputs; at most the last bit where the two compilations contract differently
    assert float(d.max()) <= 2 ** -7 * float(a_ref.float().abs().max())
    assert float((d > 0).float().mean()) < 1e-3


@pytest.mark.parametrize("cin,cout,tokens", [(48, 48, 1000), (192, 48, 777), (384, 96, 432), (768, 192, 100), (1536, 384, 54)])
def test_linear_add_and_mlp_residual_fused_equal_unfused(cin, cout, tokens):
    """res + Linear(x) with the add in the Linear kernel's epilogue (ops.linear_add, ops.mlp(res=)) == linear / mlp followed by
    ops.add, bit for bit, values and gradients (the last shape takes the unfused fallback: K = 1536 runs on the K-split kernel)"""
    from medicalsemseg_amd import ops
    dev, dtype = _dev(), torch.bfloat16
    g = torch.Generator().manual_seed(7)
    x0 = torch.randn(2, tokens, 1, 1, cin, generator=g).to(dev, dtype)
    r0 = torch.randn(2, tokens, 1, 1, cout, generator=g).to(dev, dtype)
    w = torch.nn.Parameter((torch.randn(cout, cin, generator=g) * cin ** -0.5).to(dev))
    b = torch.nn.Parameter((torch.randn(cout, generator=g) * 0.1).to(dev))
    gy = torch.randn(r0.shape, generator=g).to(dev, dtype)
    res = []
    for fused in (True, False):
        x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
        w.grad = b.grad = None
        y = ops.linear_add(x, w, b, r) if fused else ops.add(r, ops.linear(x, w, b))
        y.backward(gy)
        res.append((y.detach().clone(), x.grad.clone(), r.grad.clone(), w.grad.clone(), b.grad.clone()))
    for a_, b_ in zip(res[0], res[1]):
        assert torch.equal(a_, b_)
    if cin == 4 * cout:    # the MLP of a Swin block with the residual: x -> fc1 (cout -> cin) -> GELU -> fc2 (cin -> cout) + res
        w1 = torch.nn.Parameter((torch.randn(cin, cout, generator=g) * cout ** -0.5).to(dev))
        b1 = torch.nn.Parameter((torch.randn(cin, generator=g) * 0.1).to(dev))
        res = []
        for fused in (True, False):
            x = r0.clone().requires_grad_(True)
            for p_ in (w, b, w1, b1):
                p_.grad = None
            xr, xn = ops.layer_norm_res(x, None, None, 1e-5)
            y = ops.mlp(xn, w1, b1, w, b, res=xr) if fused else ops.add(xr, ops.mlp(xn, w1, b1, w, b))
            y.backward(gy)
            res.append((y.detach().clone(), x.grad.clone(), w.grad.clone(), w1.grad.clone()))
        for a_, b_ in zip(res[0], res[1]):
            assert torch.equal(a_, b_)


def test_postproc_kernels_bit_exact(golden_dir):
    """argmax -> uint8, nearest resample (scipy order-0 zoom semantics) and the fold majority vote: bit-exact against the
    numpy oracle, the reference's resample_3d fixture (tests/golden/resample.npz) and ragged sizes"""
    import os
    from medicalsemseg_amd import hip
    from oracle.postproc import argmax_labels, majority_vote, resample_nearest
    dev = _dev()
    g = np.load(os.path.join(golden_dir, "resample.npz"))
    got = hip.resample_nearest_u8(torch.from_numpy(g["vol"]).to(dev), g["out"].shape)
    assert np.array_equal(got.cpu().numpy(), g["out"])
    rng = np.random.default_rng(1)
    for s, t in (((17, 9, 23), (40, 31, 12)), ((33, 40, 29), (33, 40, 29)), ((5, 6, 7), (1, 13, 2)), ((64, 48, 50), (96, 96, 96))):
        v = rng.integers(0, 14, s).astype(np.uint8)
        got = hip.resample_nearest_u8(torch.from_numpy(v).to(dev), t)
        assert np.array_equal(got.cpu().numpy(), resample_nearest(v, t)), (s, t)
    for C, sp in ((3, (7, 9, 11)), (14, (16, 16, 20)), (2, (5, 5, 5))):
        x = rng.standard_normal((C,) + sp).astype(np.float32)
        x[:, 0, 0, :2] = 0.25                      # ties: the first maximum wins
        got = hip.argmax_u8(torch.from_numpy(x).to(dev))
        assert np.array_equal(got.cpu().numpy(), argmax_labels(x)), (C, sp)
    for F_, C, sp in ((5, 4, (9, 10, 11)), (3, 14, (8, 8, 8)), (1, 2, (3, 3, 3))):
        lab = rng.integers(0, C, (F_,) + sp).astype(np.uint8)
        got = hip.majority_vote_u8(torch.from_numpy(lab).to(dev), C)
        assert np.array_equal(got.cpu().numpy(), majority_vote(lab, C)), (F_, C)


def test_device_crop_augment_bit_exact_vs_oracle():
    """N1: the device-side crop + flip + rot90 + intensity kernel reproduces the numpy restatement of the reference's
    transform chain bit for bit for the same parameter rows (crop starts incl. the centre clamp are integer-exact)"""
    from medicalsemseg_amd.data_device import DevicePatchLoader
    from oracle.augment import apply_row, correct_crop_center, crop_start
    dev = _dev()
    rng = np.random.default_rng(3)
    img = rng.standard_normal((2, 40, 52, 44)).astype(np.float32)
    lab = np.zeros((40, 52, 44), dtype=np.uint8)
    lab[5:20, 30:50, 2:12] = 1
    lab[25:38, 0:9, 30:44] = 2
    roi = 16
    ld = DevicePatchLoader(torch.from_numpy(img), torch.from_numpy(lab), roi, 6, 3, dev, seed=11, pos=2.0, neg=1.0,
                           flip_prob=0.5, rot_prob=0.7, shift_prob=0.6, scale_prob=0.6, image_threshold=-10.0)
    seen_rot, seen_flip = set(), set()
    for batch in ld:
        got_i, got_l = batch["image"].cpu().numpy(), batch["label"].cpu().numpy()
        cen = torch.stack(batch["image_transforms"][0]["extra_info"]["center"], 1).numpy()
        for j, row in enumerate(ld.last_rows):
            start = (row.z0, row.y0, row.x0)
            c = tuple(int(v) for v in cen[j])
            assert correct_crop_center(c, (roi,) * 3, lab.shape) == c and crop_start(c, (roi,) * 3) == start
            assert all(0 <= s and s + roi <= n for s, n in zip(start, lab.shape))
            flips = (row.flips & 1, row.flips & 2, row.flips & 4)
            wi, wl = apply_row(img, lab, start, roi, flips, row.rotk, row.shift, row.scale)
            assert np.array_equal(got_i[j], wi), j
            assert np.array_equal(got_l[j, 0], wl), j
            seen_rot.add(row.rotk); seen_flip.add(row.flips)
    assert len(seen_rot) >= 3 and len(seen_flip) >= 4          # the seed exercises the index maps
    # the batch dict feeds the engine unchanged (crop centre record included)
    from medicalsemseg_amd.utils import misc
    rel = misc.get_rel_crop_loc(batch["image_transforms"][0])
    assert tuple(rel.shape) == (6, 3) and float(rel.min()) > 0 and float(rel.max()) < 1


@pytest.mark.parametrize("dtype", DTYPES)
def test_batchnorm_act_vs_torch(dtype):
    """BatchNorm3d + ReLU (training: batch statistics, running-statistics update; eval: running statistics) on the
    InstanceNorm kernels over the merged batch, dense and as a channel slice of a concat buffer, vs torch.nn.BatchNorm3d"""
    from medicalsemseg_amd import layers
    dev = _dev()
    N, C, sp = 3, 32, (8, 12, 16)
    x = gen(N, C, *sp, seed=21, scale=1.5) + 0.3
    r = gen(N, C, *sp, seed=22)
    xr = rnd(dtype, x)
    mk = lambda: torch.nn.BatchNorm3d(C)
    ref, mine = mk(), mk().to(dev)
    with torch.no_grad():
        ref.weight.copy_(1 + 0.1 * gen(C, seed=23)); ref.bias.copy_(0.1 * gen(C, seed=24))
    mine.load_state_dict(ref.state_dict())
    xr = xr.clone().requires_grad_(True)
    y = torch.relu(ref(xr))
    (y * rnd(dtype, r)).sum().backward()
    op = layers.BatchNormAct(mine, 0.0)
    big = torch.zeros(N, *sp, 2 * C, dtype=dtype, device=dev)
    xg = cl(x, dtype, dev)
    a, s = op.fwd(xg, out=big[..., C:])
    check(ncdhw(big[..., C:]), y.detach(), dtype, "batchnorm fwd (into a concat slice)")
    assert float(big[..., :C].float().abs().max()) == 0
    dgrid = torch.zeros(N, *sp, 2 * C, dtype=dtype, device=dev)
    dgrid[..., :C] = cl(r, dtype, dev)
    dy = op.bwd(xg, s, dgrid[..., :C])
    check(ncdhw(dy), xr.grad, dtype, "batchnorm bwd")
    check(mine.weight.grad, ref.weight.grad, dtype, "batchnorm dgamma")
    check(mine.bias.grad, ref.bias.grad, dtype, "batchnorm dbeta")
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert torch.allclose(mine.running_mean.cpu(), ref.running_mean, atol=tol) and torch.allclose(mine.running_var.cpu(), ref.running_var, atol=tol)
    assert int(mine.num_batches_tracked) == 1
    ref.eval(); mine.eval()
    with torch.no_grad():
        ye = torch.relu(ref(rnd(dtype, x)))
    ae, _ = op.fwd(xg)
    check(ncdhw(ae), ye, dtype, "batchnorm eval fwd")
    with pytest.raises(NotImplementedError):
        op.bwd(xg, s, dgrid[..., :C])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,sp", [(128, (12, 12, 12)), (192, (6, 6, 6)), (40, (5, 7, 9)), (1536, (3, 3, 3))])
def test_dwconv3_vs_torch(dtype, C, sp):
    """depthwise Conv3d k3 p1 (groups = C) forward, input gradient, weight + bias gradient (deterministic two-stage sums)
    vs torch.nn.functional.conv3d(groups=C); a second backward accumulates"""
    from medicalsemseg_amd import ops
    dev = _dev()
    N = 2
    x = gen(N, C, *sp, seed=31)
    w = gen(C, 1, 3, 3, 3, seed=32, scale=27 ** -0.5)
    b = gen(C, seed=33)
    r = gen(N, C, *sp, seed=34)
    xr, rr = rnd(dtype, x, r)
    xr = xr.clone().requires_grad_(True)      # rnd() returns the input itself for fp32
    wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    yref = F.conv3d(xr, wr, br, padding=1, groups=C)
    yref.backward(rr)
    wp = torch.nn.Parameter(w.to(dev)); bp = torch.nn.Parameter(b.to(dev))
    xg = cl(x, dtype, dev).requires_grad_(True)
    y = ops.dwconv3(xg, wp, bp)
    check(ncdhw(y), yref.detach(), dtype, "dwconv3 fwd")
    y.backward(cl(r, dtype, dev))
    check(ncdhw(xg.grad), xr.grad, dtype, "dwconv3 dgrad")
    check(wp.grad, wr.grad, dtype, "dwconv3 wgrad")
    check(bp.grad, br.grad, dtype, "dwconv3 bias grad")
    g1 = wp.grad.clone()
    y2 = ops.dwconv3(xg, wp, bp)
    y2.backward(cl(r, dtype, dev))
    assert torch.equal(wp.grad, 2 * g1)          # accumulate path; fixed-order sums -> bit-identical second pass


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("isz,osz", [((2, 2, 2), (16, 16, 16)), ((4, 4, 4), (16, 16, 16)), ((16, 16, 16), (64, 64, 64)),
                                     ((3, 5, 4), (7, 9, 6)), ((8, 8, 8), (8, 8, 8)), ((9, 6, 10), (4, 3, 5))])
def test_interp_trilinear_vs_torch(dtype, isz, osz):
    """F.interpolate(mode='trilinear', align_corners=False) and its adjoint (gather form, deterministic): up, down,
    identity, odd sizes; output / gradient as channel slices of wider buffers"""
    from medicalsemseg_amd import hip
    dev = _dev()
    N, C = 2, 16
    x = gen(N, C, *isz, seed=41)
    r = gen(N, C, *osz, seed=42)
    xr, rr = rnd(dtype, x, r)
    xr = xr.clone().requires_grad_(True)
    yref = F.interpolate(xr, size=osz, mode="trilinear", align_corners=False)
    yref.backward(rr)
    big = torch.zeros(N, *osz, 3 * C, dtype=dtype, device=dev)
    hip.interp_trilinear(cl(x, dtype, dev), big[..., C:2 * C])
    check(ncdhw(big[..., C:2 * C]), yref.detach(), dtype, "trilinear fwd")
    assert float(big[..., :C].float().abs().max()) == 0 and float(big[..., 2 * C:].float().abs().max()) == 0
    gbig = torch.zeros(N, *osz, 2 * C, dtype=dtype, device=dev)
    gbig[..., C:] = cl(r, dtype, dev)
    dx = torch.empty(N, *isz, C, dtype=dtype, device=dev)
    hip.interp_trilinear_bwd(gbig[..., C:], dx)
    check(ncdhw(dx), xr.grad, dtype, "trilinear bwd")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,M,heads,hd", [(4096, 8, 1, 32), (512, 27, 2, 48), (216, 27, 4, 48), (27, 27, 8, 48),
                                          (300, 70, 2, 16), (64, 130, 1, 64)])
def test_kv_attention_vs_torch(dtype, N, M, heads, hd):
    """softmax(q k^T / sqrt(hd)) v with a small key set (SegFormer's spatial-reduction attention): forward, dq, dk, dv vs
    the reference's formulation in torch (segformer_backbone.py:96-117); more than one 64-key LDS tile included"""
    from medicalsemseg_amd import ops
    dev = _dev()
    B, C = 2, heads * hd
    q = gen(B, N, C, seed=51)
    kv = gen(B, M, 2 * C, seed=52)
    r = gen(B, N, C, seed=53)
    qr, kvr, rr = rnd(dtype, q, kv, r)
    qr = qr.clone().requires_grad_(True); kvr = kvr.clone().requires_grad_(True)
    qh = qr.reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    kvh = kvr.reshape(B, M, 2, heads, hd).permute(2, 0, 3, 1, 4)
    attn = ((qh @ kvh[0].transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    yref = (attn @ kvh[1]).transpose(1, 2).reshape(B, N, C)
    yref.backward(rr)
    qg = q.to(dev, dtype).requires_grad_(True); kvg = kv.to(dev, dtype).requires_grad_(True)
    y = ops.kv_attention(qg, kvg, heads)
    check(y, yref.detach(), dtype, "kv attention fwd")
    y.backward(r.to(dev, dtype))
    check(qg.grad, qr.grad, dtype, "kv attention dq")
    check(kvg.grad, kvr.grad, dtype, "kv attention dkv", scale=float(kvr.grad.abs().max()))
    g1 = kvg.grad.clone()
    kvg.grad = None; qg.grad = None
    ops.kv_attention(qg, kvg, heads).backward(r.to(dev, dtype))
    assert torch.equal(kvg.grad, g1)                       # fixed-order sums: bit-reproducible


def test_dropout3d_with_given_mask():
    from medicalsemseg_amd import ops
    dev = _dev()
    x = gen(3, 4, 5, 6, 16, seed=61).to(dev).requires_grad_(True)
    mask = (gen(3, 16, seed=62) > 0).float()
    y = ops.dropout3d(x, 0.25, True, mask)
    want = x.detach() * (mask / 0.75).to(dev).view(3, 1, 1, 1, 16)
    assert torch.allclose(y, want, rtol=1e-6, atol=1e-7)
    y.sum().backward()
    assert torch.allclose(x.grad, (mask / 0.75).to(dev).view(3, 1, 1, 1, 16).expand_as(x), rtol=1e-6)
    assert ops.dropout3d(x, 0.25, False) is x


@pytest.mark.parametrize("dtype", DTYPES)
def test_avg_pool3_no_systematic_bias(dtype):
    """AvgPool3d(3, 1, 1) of the SwInception pooling branch (/root/reference/models/backbones/swinception.py:113-116): unit
    taps and the fp32 sum scaled by 1/27 -- a bf16 tap of 1/27 (0.0371094) would put +0.195 % on every output, forward and
    backward.  Positive inputs make a scale error visible as a mean relative error."""
    from medicalsemseg_amd import ops
    DEV = _dev()
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(2, 10, 12, 14, 16, generator=g) + 0.5).to(DEV).to(dtype).requires_grad_(True)
    y = ops.avg_pool3(x)
    ref = torch.nn.functional.avg_pool3d(x.detach().float().permute(0, 4, 1, 2, 3), 3, 1, 1).permute(0, 2, 3, 4, 1)
    rel = (y.detach().float() - ref) / ref
    r = torch.rand(y.shape, generator=g).to(DEV) + 0.5
    y.backward(r.to(dtype))
    gref = torch.nn.functional.avg_pool3d(r.to(dtype).float().permute(0, 4, 1, 2, 3), 3, 1, 1).permute(0, 2, 3, 4, 1)
    grel = (x.grad.float() - gref) / gref
    tol = 1e-6 if dtype == torch.float32 else 2.0 ** -8
    print(f"[{dtype}] avg_pool3 mean rel err fwd {float(rel.mean()):.2e} bwd {float(grel.mean()):.2e}, max {float(rel.abs().max()):.2e}")
    assert float(rel.abs().max()) < tol and float(grel.abs().max()) < tol
    assert abs(float(rel.mean())) < 2e-4 and abs(float(grel.mean())) < 2e-4      # the bf16-tap form sat at +1.95e-3


def test_eval_batchnorm_large_mean_small_variance():
    """eval-mode BatchNorm with running_mean^2 >> running_var (mean 10, var 1e-2): the folded scale / shift form keeps
    full fp32 accuracy where sums rebuilt from the running statistics lose three digits (E[x^2] - mean^2)"""
    from medicalsemseg_amd import ops
    DEV = _dev()
    C = 16
    bn = torch.nn.BatchNorm3d(C, eps=1e-5).to(DEV).eval()
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        bn.running_mean.copy_(10.0 + torch.rand(C, generator=g).to(DEV))
        bn.running_var.copy_(1e-2 * (1 + torch.rand(C, generator=g).to(DEV)))
        bn.weight.copy_(1 + 0.1 * torch.randn(C, generator=g).to(DEV))
        bn.bias.copy_(0.1 * torch.randn(C, generator=g).to(DEV))
    x = (10.5 + 0.1 * torch.randn(2, 6, 6, 6, C, generator=g)).to(DEV)
    with torch.no_grad():
        y = ops.batch_norm(x, bn, None)
        xd = x.double()
        want = (xd - bn.running_mean.double()) * torch.rsqrt(bn.running_var.double() + bn.eps) * bn.weight.double() + bn.bias.double()
    err = float((y.double() - want).abs().max()) / float(want.abs().max())
    print(f"eval BatchNorm (mean 10, var 1e-2): max err / scale {err:.2e}")
    assert err < 5e-6


@pytest.mark.parametrize("cin,cout,sp,N", [(64, 64, (3, 3, 3), 2), (128, 64, (6, 6, 6), 2), (96, 32, (12, 12, 12), 1), (768, 768, (3, 3, 3), 2)])
def test_resblock_small_grid_forward_backward(monkeypatch, cin, cout, sp, N):
    """layers.ResBlock (MONAI UnetResBlock: Swin-UNETR encoder4 / encoder10 / decoder5 / decoder4) on the 12^3 ... 3^3 grids:
    both 3x3x3 convolutions as split-K partials + one finish kernel each (raw output, statistics, normalise, residual, LeakyReLU;
    3^3 = one quarter-filled tile per sample) and the input gradients on the same path, against the tile kernels
    (MSSEG_NO_K3_SMALL=1) and torch fp32 on the bf16-rounded operands"""
    from medicalsemseg_amd import hip, layers
    dev, dtype = _dev(), torch.bfloat16
    x = gen(N, cin, *sp, seed=1)
    do = gen(N, cout, *sp, seed=2)
    mk = lambda *sh, seed: torch.nn.Parameter((gen(*sh, seed=seed) * (sh[1] * (27 if len(sh) == 5 and sh[2] == 3 else 1)) ** -0.5).to(dev))   # noqa: E731
    w1, w2 = mk(cout, cin, 3, 3, 3, seed=3), mk(cout, cout, 3, 3, 3, seed=4)
    w3 = mk(cout, cin, 1, 1, 1, seed=5) if cin != cout else None
    xg, dog = cl(x, dtype, dev), cl(do, dtype, dev)
    res = {}
    for mode in ("small", "tile"):
        if mode == "tile":
            monkeypatch.setenv("MSSEG_NO_K3_SMALL", "1")
        else:
            monkeypatch.delenv("MSSEG_NO_K3_SMALL", raising=False)
            assert hip.conv3d_k3_small_ok(xg, cin, cout)
        for w in (w1, w2, w3):
            if w is not None:
                w.grad = None
        blk = layers.ResBlock(w1, w2, w3)
        layers.WGRAD_SIDE.begin(xg.device)
        o, saved = blk.fwd(xg)
        dx = blk.bwd(saved, dog, True)
        layers.WGRAD_SIDE.join()
        res[mode] = (o.float().clone(), dx.float().clone(), w1.grad.clone(), w2.grad.clone())
    monkeypatch.delenv("MSSEG_NO_K3_SMALL", raising=False)
    # torch reference
    xr = x.to(dtype).float().requires_grad_(True)
    w1r, w2r = w1.detach().cpu().to(dtype).float().requires_grad_(True), w2.detach().cpu().to(dtype).float().requires_grad_(True)
    a1 = F.leaky_relu(F.instance_norm(F.conv3d(xr, w1r, padding=1)), 0.01)
    y2 = F.instance_norm(F.conv3d(a1, w2r, padding=1))
    r = xr if w3 is None else F.instance_norm(F.conv3d(xr, w3.detach().cpu().to(dtype).float()))
    oref = F.leaky_relu(y2 + r, 0.01)
    oref.backward(do.to(dtype).float())
    errs = {}
    for mode in ("small", "tile"):
        o, dx, g1, g2 = res[mode]
        so = float(oref.detach().abs().max())
        eo = float((ncdhw(o).cpu() - oref.detach()).abs().max()) / so
        ex = float((ncdhw(dx).cpu() - xr.grad).norm() / xr.grad.norm())    # rel. L2: single voxels flip with a LeakyReLU sign
        e1 = float((g1.cpu() - w1r.grad).norm() / w1r.grad.norm())
        e2 = float((g2.cpu() - w2r.grad).norm() / w2r.grad.norm())
        print(f"[{mode}] out {eo:.3e} dx {ex:.3e} dW1 {e1:.3e} dW2 {e2:.3e}")
        errs[mode] = (eo, ex, e1, e2)
    # bf16 InstanceNorm over 27 ... 1728 voxels: the gradient error against fp32 is set by the statistics (the tile kernels
    # show the same figures); the small path must not be worse than the tile path by more than a quarter
    for es, et in zip(errs["small"], errs["tile"]):
        assert es < 0.2 and es < 1.25 * et + 2e-3, (errs["small"], errs["tile"])
    # the two kernel paths agree much closer than either does with fp32
    o_s, dx_s = res["small"][0], res["small"][1]
    o_t, dx_t = res["tile"][0], res["tile"][1]
    assert float((o_s - o_t).abs().max()) / float(o_t.abs().max()) < 1.6e-2
    assert float((dx_s - dx_t).norm() / dx_t.norm()) < 3e-2


@pytest.mark.parametrize("cin,cout,sp", [(96, 48, (32, 32, 48)), (48, 48, (32, 40, 32)), (64, 32, (32, 32, 40))])
def test_resblock_large_grid_shortcut_gradient_accumulated(monkeypatch, cin, cout, sp):
    """layers.ResBlock backward on grids of the 48-channel ping-pong kernel (Swin-UNETR decoder1 / decoder2 / encoder2): the first
    conv's input gradient is added onto the shortcut's gradient by the kernel's accumulate epilogue instead of a separate add pass --
    against the add-pass form (MSSEG_NO_DGRAD_ACCUM=1) and torch fp32 on the bf16-rounded operands"""
    from medicalsemseg_amd import layers
    dev, dtype, N = _dev(), torch.bfloat16, 2
    x = gen(N, cin, *sp, seed=1)
    do = gen(N, cout, *sp, seed=2)
    mk = lambda *sh, seed: torch.nn.Parameter((gen(*sh, seed=seed) * (sh[1] * (27 if sh[2] == 3 else 1)) ** -0.5).to(dev))   # noqa: E731
    w1, w2 = mk(cout, cin, 3, 3, 3, seed=3), mk(cout, cout, 3, 3, 3, seed=4)
    w3 = mk(cout, cin, 1, 1, 1, seed=5) if cin != cout else None
    xg, dog = cl(x, dtype, dev), cl(do, dtype, dev)
    res = {}
    for mode in ("accum", "add"):
        if mode == "add":
            monkeypatch.setenv("MSSEG_NO_DGRAD_ACCUM", "1")
        else:
            monkeypatch.delenv("MSSEG_NO_DGRAD_ACCUM", raising=False)
        for w in (w1, w2, w3):
            if w is not None:
                w.grad = None
        blk = layers.ResBlock(w1, w2, w3)
        layers.WGRAD_SIDE.begin(xg.device)
        o, saved = blk.fwd(xg)
        if mode == "accum":
            assert blk.c1.dgrad_accumulate_ok(saved[1])
        dx = blk.bwd(saved, dog, True)
        layers.WGRAD_SIDE.join()
        res[mode] = (dx.float().clone(), w1.grad.clone())
    monkeypatch.delenv("MSSEG_NO_DGRAD_ACCUM", raising=False)
    assert torch.equal(res["accum"][1], res["add"][1])
    d = float((res["accum"][0] - res["add"][0]).norm() / res["add"][0].norm())
    print(f"accumulate epilogue vs add pass: rel. L2 {d:.3e}")
    assert d < 5e-3
    xr = x.to(dtype).float().requires_grad_(True)
    cpu = lambda w: w.detach().cpu().to(dtype).float()   # noqa: E731
    a1 = F.leaky_relu(F.instance_norm(F.conv3d(xr, cpu(w1), padding=1)), 0.01)
    y2 = F.instance_norm(F.conv3d(a1, cpu(w2), padding=1))
    r = xr if w3 is None else F.instance_norm(F.conv3d(xr, cpu(w3)))
    F.leaky_relu(y2 + r, 0.01).backward(do.to(dtype).float())
    e = {m: float((ncdhw(res[m][0]) - xr.grad).norm() / xr.grad.norm()) for m in res}
    print(f"input gradient vs fp32: {e}")
    # bf16 storage of every intermediate: both forms sit at 4.3e-2 against fp32 (measured); the gate is 2x that, and the new form
    # must not be worse than the add pass
    assert e["accum"] < 9e-2 and e["accum"] < 1.05 * e["add"] + 1e-3


@pytest.mark.parametrize("cin,cmid,sp,N,pool", [(64, 128, (12, 12, 12), 2, True), (256, 128, (12, 12, 12), 1, False),
                                                (128, 256, (6, 6, 6), 2, False), (32, 32, (6, 12, 6), 8, True),
                                                (96, 64, (12, 6, 12), 3, True)])
def test_conv3d_k3_small_grid_twoconv_unit(monkeypatch, cin, cmid, sp, N, pool):
    """The split-K small-grid path (csrc/conv3d_k3_small.hip; the 12^3 / 6^3 levels of the UNet): a TwoConv (conv + IN +
    LeakyReLU twice, MONAI BasicUNet) forward with the second activation written into a concat-buffer slice and max-pooled,
    and its backward -- the input-gradient finish kernel runs the first unit's whole InstanceNorm backward -- against
    (a) torch fp32 on bf16-rounded operands and (b) the unfused kernels (MSSEG_NO_K3_SMALL=1), which must agree much closer."""
    from medicalsemseg_amd import hip
    from medicalsemseg_amd.layers import Conv3, ConvNormAct, InstNormAct
    dev = _dev()
    dtype = torch.bfloat16
    x = gen(N, cin, *sp, seed=1)
    w0 = gen(cmid, cin, 3, 3, 3, seed=2, scale=(cin * 27) ** -0.5)
    w1 = gen(cmid, cmid, 3, 3, 3, seed=3, scale=(cmid * 27) ** -0.5)
    b0, b1 = gen(cmid, seed=4), gen(cmid, seed=5)
    g0, g1 = 1 + 0.2 * gen(cmid, seed=6), 1 + 0.2 * gen(cmid, seed=7)
    be0, be1 = 0.2 * gen(cmid, seed=8), 0.2 * gen(cmid, seed=9)
    psp = tuple(s // 2 for s in sp)
    r_act, r_pool = gen(N, cmid, *sp, seed=10), gen(N, cmid, *psp, seed=11)
    # (a) torch reference
    xr, w0r, w1r = rnd(dtype, x, w0, w1)
    leaves = [t.clone().requires_grad_(True) for t in (xr, w0r, w1r, g0, be0, g1, be1)]
    xr, w0r, w1r, g0r, be0r, g1r, be1r = leaves
    y0 = F.conv3d(xr, w0r, b0, padding=1)
    a0 = F.leaky_relu(F.instance_norm(y0, weight=g0r, bias=be0r, eps=1e-5), 0.1)
    y1 = F.conv3d(a0, w1r, b1, padding=1)
    a1 = F.leaky_relu(F.instance_norm(y1, weight=g1r, bias=be1r, eps=1e-5), 0.1)
    loss = (a1 * r_act).sum() + ((F.max_pool3d(a1, 2) * r_pool).sum() if pool else 0.0)
    loss.backward()

    def run(small):
        if small:
            monkeypatch.delenv("MSSEG_NO_K3_SMALL", raising=False)
        else:
            monkeypatch.setenv("MSSEG_NO_K3_SMALL", "1")
        P = lambda t: torch.nn.Parameter(t.clone().to(dev))   # noqa: E731
        ps = dict(w0=P(w0), b0=P(b0), g0=P(g0), be0=P(be0), w1=P(w1), b1=P(b1), g1=P(g1), be1=P(be1))
        u0 = ConvNormAct(Conv3(ps["w0"], ps["b0"]), InstNormAct(ps["g0"], ps["be0"], 0.1))
        u1 = ConvNormAct(Conv3(ps["w1"], ps["b1"]), InstNormAct(ps["g1"], ps["be1"], 0.1))
        xg = cl(x, dtype, dev)
        cat = torch.full((N,) + tuple(sp) + (cmid + 32,), 3.0, dtype=dtype, device=dev)
        pooled = torch.empty((N,) + psp + (cmid,), dtype=dtype, device=dev) if pool else None
        hip.TIMER.records.clear()
        hip.TIMER.enabled = True
        try:
            A0, s0 = u0.fwd(xg)
            A1, s1 = u1.fwd(A0, cat[..., :cmid], pooled=pooled)
            # gradient of the second activation: direct part + max-pool part (as the encoder's pool-backward kernel forms it)
            da1 = cl(r_act, dtype, dev)
            if pool:
                g = cl(r_pool, dtype, dev)
                hip.maxpool2_bwd(A1, g, da1, accumulate=True)
            da0, red = u1.bwd(s1, da1, True, next_saved=s0, next_cna=u0)
            dx = u0.bwd(s0, da0, True, red=red)
            torch.cuda.synchronize()
        finally:
            hip.TIMER.enabled = False
        keys = set(hip.TIMER.summary())
        hip.TIMER.records.clear()
        assert ("conv3d_k3_small" in keys) == small, keys
        assert float((cat[..., cmid:].float() - 3.0).abs().max()) == 0.0      # the other half of the concat buffer is untouched
        out = dict(a1=ncdhw(A1), pooled=ncdhw(pooled) if pool else None, dx=ncdhw(dx), y0=ncdhw(s0[1]), stats0=s0[2].cpu(),
                   stats1=s1[2].cpu(), **{k: v.grad.detach().cpu() for k, v in ps.items() if v.grad is not None})
        return out

    got, base = run(True), run(False)
    ref = dict(a1=a1.detach(), pooled=F.max_pool3d(a1, 2).detach() if pool else None, dx=xr.grad, w0=w0r.grad, w1=w1r.grad,
               g0=g0r.grad, be0=be0r.grad, g1=g1r.grad, be1=be1r.grad)
    for k in ("a1", "pooled", "dx", "w0", "w1", "g0", "be0", "g1", "be1"):
        if ref[k] is None:
            continue
        s = float(ref[k].abs().max())
        e_ref = float((got[k] - ref[k]).abs().max()) / s
        e_base = float((got[k] - base[k]).abs().max()) / s
        e_bref = float((base[k] - ref[k]).abs().max()) / s
        print(f"{k}: small vs torch {e_ref:.2e}, unfused vs torch {e_bref:.2e}, small vs unfused {e_base:.2e}")
        # two bf16 conv + norm stages: a few bf16 ulps of the scale; never worse than 1.5x the unfused kernels' own error
        assert e_ref < max(3e-2, 1.5 * e_bref), (k, e_ref, e_bref)
    # statistics of the raw outputs (sum, sum of squares per (n, c)): same stored values -> same sums up to fp32 order
    for k in ("stats0", "stats1"):
        assert torch.allclose(got[k], base[k], rtol=2e-3, atol=2e-2 * float(base[k].abs().max()) * 1e-2), k
    assert float((got["y0"] - base["y0"]).abs().max()) / float(base["y0"].abs().max()) < 1e-2


def test_conv3d_k3_split_concat_two_launches():
    """Conv3.fwd_split: conv(cat([a, b])) + bias as two launches of the ping-pong kernel (the second accumulates onto the
    stored bf16 result and forms the InstanceNorm statistics of the sums; csrc/conv3d_k3_pp.hip STATS == 3) -- the inference
    forward of BasicUNet's UpCat convs at the 96^3 / 48^3 levels -- against torch conv3d on the concatenated tensor and
    against the one-launch kernel on a concat buffer."""
    from medicalsemseg_amd import hip
    from medicalsemseg_amd.layers import Conv3
    dev = _dev()
    dt = torch.bfloat16
    N, sp, cout = 2, (32, 48, 64), 32
    xa, xb = gen(N, 32, *sp, seed=1), gen(N, 32, *sp, seed=2)
    w = gen(cout, 64, 3, 3, 3, seed=3, scale=(64 * 27) ** -0.5)
    b = gen(cout, seed=4)
    xar, xbr, wr = rnd(dt, xa, xb, w)
    ref = F.conv3d(torch.cat([xar, xbr], 1), wr, b, padding=1)
    op = Conv3(torch.nn.Parameter(w.to(dev)), torch.nn.Parameter(b.to(dev)))
    assert op.split_ok((N,) + sp, dt, 32, 32)
    A, B = cl(xa, dt, dev), cl(xb, dt, dev)
    hip.TIMER.records.clear()
    hip.TIMER.enabled = True
    try:
        y, stats = op.fwd_split(A, B)
        torch.cuda.synchronize()
    finally:
        hip.TIMER.enabled = False
    assert hip.TIMER.summary().get("conv3d_k3_fwd/v3", {}).get("launches", 0) == 2      # both halves on the ping-pong kernel
    hip.TIMER.records.clear()
    check(ncdhw(y), ref, dt, "split-concat conv")
    yf = y.float()
    want = torch.stack([yf.sum((1, 2, 3)), (yf ** 2).sum((1, 2, 3))], -1)
    assert torch.allclose(stats, want, rtol=2e-4, atol=1e-2 * float(want.abs().max()) * 1e-2)
    cat = torch.cat([A, B], -1).contiguous()
    y1 = op.fwd(cat)
    d = float((y.float() - y1.float()).abs().max()) / float(y1.float().abs().max())
    print(f"split-concat vs one launch: max diff / scale {d:.2e}")
    assert d < 1e-2        # one extra bf16 rounding of the intermediate sum
    # the weight halves are repacked with the rest after an in-place weight update
    with torch.no_grad():
        op.w.mul_(0.5)
    y2, _ = op.fwd_split(A, B)
    check(ncdhw(y2), F.conv3d(torch.cat([xar, xbr], 1), rnd(dt, w * 0.5), b, padding=1), dt, "split-concat conv after a weight update")
