"""Pin the CPU oracle against golden vectors produced by the reference's own modules
(oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import swin as osw
from tests.golden_util import det_fill_, det_tensor


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag,dim,ws,heads", [("h3w6", 48, 6, 3), ("h24w3", 384, 3, 24)])
def test_window_attention(golden_dir, tag, dim, ws, heads):
    g = _load(golden_dir, f"swin_attn_{tag}.npz")
    m = osw.WindowAttention(dim, ws, heads)
    det_fill_(m, "attn_" + tag)
    N = ws ** 3
    x = det_tensor("attn_x_" + tag, (8, N, dim)).requires_grad_(True)
    r = det_tensor("attn_r_" + tag, (8, N, dim))
    mask = osw.shift_region_mask(2 * ws, 2 * ws, 2 * ws, ws, ws // 2)
    for mk, msk in (("nomask", None), ("mask", mask)):
        y = m(x, msk)
        gx, gt = torch.autograd.grad((y * r).sum(), [x, m.relative_position_bias_table])
        _close(y, g[f"y_{mk}"])
        _close(gx, g[f"dx_{mk}"], atol=1e-4)
        _close(gt, g[f"dtable_{mk}"], atol=1e-3)


def test_block(golden_dir):
    g = _load(golden_dir, "swin_block.npz")
    mask = osw.shift_region_mask(12, 12, 12, 6, 3)
    for shift in (0, 3):
        m = osw.SwinTransformerBlock(48, (12, 12, 12), 3, 6, shift)
        det_fill_(m, "blk")
        x = det_tensor("blk_x", (2, 12 ** 3, 48)).requires_grad_(True)
        y = m(x, mask)
        _close(y, g[f"y_shift{shift}"], atol=1e-4)
        if shift:
            gx, = torch.autograd.grad((y * det_tensor("blk_r", y.shape)).sum(), x)
            _close(gx, g["dx_shift3"], atol=1e-3)


def test_layer_mask_and_downsample(golden_dir):
    g = _load(golden_dir, "swin_layer.npz")
    mask = osw.shift_region_mask(12, 12, 12, 6, 3)
    np.testing.assert_array_equal(mask.numpy().astype(np.int8), g["mask"])
    layer = osw.BasicLayer(48, (10, 12, 12), 2, 3, 6)
    det_fill_(layer, "layer")
    x = det_tensor("layer_x", (1, 10 * 12 * 12, 48))
    xo, xd, S, H, W = layer(x, 10, 12, 12)
    _close(xo, g["x_out"], atol=1e-4)
    _close(xd, g["x_down"], atol=1e-4)
    assert [10, 12, 12, S, H, W] == list(g["dims"])


@pytest.mark.parametrize("tag,vol", [("v24", (24, 24, 24)), ("v20", (20, 20, 20))])
def test_encoder(golden_dir, tag, vol):
    g = _load(golden_dir, f"swin_encoder_{tag}.npz")
    m = osw.SwinTransformerNNFormer(vol, (2, 2, 2), 1, 32, (2, 2), (2, 4), (6, 3))
    det_fill_(m, "enc")
    x = det_tensor("enc_x_" + tag, (2, 1) + vol).requires_grad_(True)
    outs = m((x, None, None))
    for i, o in enumerate(outs):
        _close(o, g[f"out{i}"], rtol=1e-3, atol=1e-4)
    loss = sum((o * det_tensor(f"enc_r{i}_" + tag, o.shape)).sum() for i, o in enumerate(outs))
    loss.backward()
    _close(x.grad, g["dx"], rtol=1e-3, atol=1e-3)
    _close(m.layers[0].blocks[1].attn.qkv.weight.grad, g["d_qkv_w"], rtol=1e-3, atol=2e-2)
    _close(m.layers[0].blocks[1].attn.relative_position_bias_table.grad, g["d_table"], rtol=1e-3, atol=1e-2)
    _close(m.layers[1].downsample.reduction.weight.grad[:8], g["d_merge_w"], rtol=1e-3, atol=1e-2)
