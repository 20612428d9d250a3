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


def _unetr_params():
    """the deterministic fills oracle/gen_golden.py applied to the reference's blocks (golden_util.det_fill_ rules)"""
    def mat(tag, shape):
        return det_tensor(tag, shape, 1.0 / np.sqrt(int(np.prod(shape[1:])))).float()   # det_fill_ copies into fp32 parameters
    return {
        "conv_w": mat("unetr_conv.block.weight", (32, 16, 3, 3, 3)), "conv_b": det_tensor("unetr_conv.block.bias", (32,), 0.1),
        "deconv_w": mat("unetr_deconv.block.weight", (32, 16, 2, 2, 2)), "deconv_b": det_tensor("unetr_deconv.block.bias", (16,), 0.1),
        "blk_dw": mat("unetr_block.block.0.block.weight", (32, 16, 2, 2, 2)), "blk_db": det_tensor("unetr_block.block.0.block.bias", (16,), 0.1),
        "blk_cw": mat("unetr_block.block.1.block.weight", (16, 16, 3, 3, 3)), "blk_cb": det_tensor("unetr_block.block.1.block.bias", (16,), 0.1),
        "blk_bn_w": det_tensor("unetr_block.block.2.weight", (16,), 0.1, 1.0), "blk_bn_b": det_tensor("unetr_block.block.2.bias", (16,), 0.1),
    }


def test_unetr_conv_blocks(golden_dir):
    """The stock torch ops the oracle is built from (conv3d k3 p1, conv_transpose3d k2 s2, eval BatchNorm + ReLU)
    reproduce the reference's own UNETR decoder blocks (/root/reference/models/segmentors/unetr.py:9-52)."""
    import torch.nn.functional as F
    g = _load(golden_dir, "unetr_blocks.npz")
    P = _unetr_params()
    x = det_tensor("unetr_x", (2, 16, 12, 12, 12)).requires_grad_(True)
    w, b = P["conv_w"].clone().requires_grad_(True), P["conv_b"].clone().requires_grad_(True)
    y = F.conv3d(x, w, b, padding=1)
    (y * det_tensor("unetr_r", tuple(y.shape))).sum().backward()
    _close(y, g["conv_y"]); _close(x.grad, g["conv_dx"]); _close(w.grad, g["conv_dw"], 1e-4, 1e-4); _close(b.grad, g["conv_db"], 1e-4, 1e-4)
    x2 = det_tensor("unetr_x2", (2, 32, 6, 6, 6)).requires_grad_(True)
    w2, b2 = P["deconv_w"].clone().requires_grad_(True), P["deconv_b"].clone().requires_grad_(True)
    y2 = F.conv_transpose3d(x2, w2, b2, stride=2)
    (y2 * det_tensor("unetr_r2", tuple(y2.shape))).sum().backward()
    _close(y2, g["deconv_y"]); _close(x2.grad, g["deconv_dx"]); _close(w2.grad, g["deconv_dw"], 1e-4, 1e-4)
    _close(b2.grad, g["deconv_db"], 1e-4, 1e-4)
    with torch.no_grad():
        t = F.conv3d(F.conv_transpose3d(x2, P["blk_dw"], P["blk_db"], stride=2), P["blk_cw"], P["blk_cb"], padding=1)
        t = torch.relu(t / np.sqrt(1.0 + 1e-5) * P["blk_bn_w"].view(1, -1, 1, 1, 1) + P["blk_bn_b"].view(1, -1, 1, 1, 1))
    _close(t, g["block_y"])


def test_sliding_window_loop_vs_reference_loop(golden_dir):
    """oracle/sliding_window.py against the reference's OWN loop (/root/reference/engine/utils.py:19-159, run by
    oracle/gen_golden.py with the MONAI helper names bound to the oracle's restatements): padding, window order, the
    relative `centers` (incl. the unsqueeze quirk at sw_batch_size 1), blend order and the final slicing"""
    from oracle.sliding_window import sliding_window_inference
    from tests.golden_util import SW_CASES, sw_predictor
    g = _load(golden_dir, "sliding_window_ref.npz")
    for tag, vol, roi, sb, ov, mode, cval in SW_CASES:
        x = det_tensor("sw_x_" + tag, vol)
        aff = det_tensor("sw_aff_" + tag, (vol[0], 3))
        y = sliding_window_inference(x, aff, roi, sb, sw_predictor, overlap=ov, mode=mode, cval=cval)
        assert tuple(y.shape) == g["out_" + tag].shape, tag
        np.testing.assert_allclose(y.numpy(), g["out_" + tag], rtol=1e-6, atol=1e-6, err_msg=tag)


def test_drop_path_and_trunc_normal_vs_reference_layers(golden_dir):
    """the reference's vendored DropPath / trunc_normal_ under fixed CPU seeds: the oracle restatements and torch's own
    trunc_normal_ (what the product's modules initialise with) reproduce them bit for bit"""
    from oracle.layers import drop_path, trunc_normal_
    g = _load(golden_dir, "layers_ref.npz")
    x = torch.from_numpy(g["dp_x"])
    torch.manual_seed(7)
    y = drop_path(x, 0.2, training=True)
    np.testing.assert_array_equal(y.numpy(), g["dp_y"])
    np.testing.assert_array_equal(drop_path(x, 0.2, training=False).numpy(), g["dp_eval"])
    kept = np.abs(g["dp_y"]).reshape(16, -1).max(1) > 0
    assert 0 < kept.sum() < 16                      # the seed exercises both branches
    np.testing.assert_allclose(g["dp_y"][kept], g["dp_x"][kept] / 0.8, rtol=1e-6)
    torch.manual_seed(3)
    np.testing.assert_array_equal(trunc_normal_(torch.empty(64, 48), std=0.02).numpy(), g["tn"])
    torch.manual_seed(4)
    np.testing.assert_array_equal(trunc_normal_(torch.empty(257), mean=0.5, std=1.0, a=-1.0, b=2.0).numpy(), g["tn2"])
    torch.manual_seed(3)
    np.testing.assert_array_equal(torch.nn.init.trunc_normal_(torch.empty(64, 48), std=0.02).numpy(), g["tn"])
    assert np.abs(g["tn"]).max() <= 2.0 and abs(float(g["tn"].std()) - 0.02) < 2e-3
    assert g["tn2"].min() >= -1.0 and g["tn2"].max() <= 2.0


def test_postproc_oracle(golden_dir):
    """nearest resample against the reference's resample_3d fixture and scipy's zoom; majority vote known answers"""
    from scipy import ndimage
    from oracle.postproc import argmax_labels, majority_vote, resample_nearest
    g = _load(golden_dir, "resample.npz")
    np.testing.assert_array_equal(resample_nearest(g["vol"], g["out"].shape), g["out"])
    rng = np.random.default_rng(0)
    for _ in range(40):
        s, t = tuple(rng.integers(1, 20, 3)), tuple(rng.integers(1, 25, 3))
        v = rng.integers(0, 5, s).astype(np.uint8)
        ref = ndimage.zoom(v, tuple(float(a) / float(b) for a, b in zip(t, s)), order=0, prefilter=False)
        if ref.shape == t:
            np.testing.assert_array_equal(resample_nearest(v, t), ref)
    # majority vote: a foreground class needs two votes to beat the background's initial one; ties -> lower class index
    folds = np.array([[1, 1, 2, 0, 2], [1, 0, 2, 0, 1], [2, 0, 1, 0, 1], [0, 0, 1, 3, 2], [0, 0, 0, 3, 2]], dtype=np.uint8).reshape(5, 1, 1, 5)
    np.testing.assert_array_equal(majority_vote(folds, 4).ravel(), [1, 0, 1, 3, 2])
    x = rng.standard_normal((3, 4, 5, 6)).astype(np.float32)
    np.testing.assert_array_equal(argmax_labels(x), x.argmax(0).astype(np.uint8))


def test_swin_official_vs_reference_file(golden_dir):
    """oracle/swin_official.py against the reference's own swin_unetr_official.py (vendored MONAI SwinUNETR, run by
    oracle/gen_golden.py): encoder features + gradients at 28^3 (padded 7-windows, clamped windows with the sliced index,
    duplicated-sub-grid patch merging, un-affine proj_out) and the whole network at 64^3"""
    from oracle import swin_official as so
    g = _load(golden_dir, "swin_official_encoder.npz")
    vit = so.SwinTransformer(1, 24, (7, 7, 7), (2, 2, 2), (2, 2, 2, 2), (3, 6, 12, 24))
    det_fill_(vit, "swo_vit.")
    x = det_tensor("swo_x28", (1, 1, 28, 28, 28)).requires_grad_(True)
    outs = vit(x, True)
    for i, o in enumerate(outs):
        _close(o, g[f"out{i}"], rtol=1e-3, atol=1e-4)
    sum((o * det_tensor(f"swo_r{i}", o.shape)).sum() for i, o in enumerate(outs)).backward()
    blk = vit.layers1[0].blocks[1]
    _close(x.grad, g["dx"], rtol=1e-3, atol=1e-3)
    _close(blk.attn.qkv.weight.grad, g["d_qkv_w"], rtol=1e-3, atol=2e-2)
    _close(blk.attn.qkv.bias.grad, g["d_qkv_b"], rtol=1e-3, atol=2e-2)
    _close(blk.attn.relative_position_bias_table.grad, g["d_table"], rtol=1e-3, atol=1e-2)
    _close(vit.layers1[0].downsample.reduction.weight.grad, g["d_merge_w"], rtol=1e-3, atol=1e-2)
    _close(vit.layers2[0].blocks[1].attn.relative_position_bias_table.grad, g["d_table_l2"], rtol=1e-3, atol=1e-2)
    gn = _load(golden_dir, "swin_official_net.npz")
    net = so.SwinUNETR((64, 64, 64), 1, 3, feature_size=24)
    det_fill_(net, "swo_net.")
    y = net(det_tensor("swo_x64", (1, 1, 64, 64, 64)))
    _close(y[:, :, ::2, ::2, ::2], gn["logits_s2"], rtol=1e-3, atol=1e-4)
    assert abs(float(y.double().sum()) - float(gn["logits_sum"])) < 1e-3 * float(gn["logits_abs"])
    (y * det_tensor("swo_ry", y.shape)).sum().backward()
    _close(net.out.conv.conv.weight.grad, gn["d_out_w"], rtol=1e-3, atol=1e-2)
    _close(net.encoder1.layer.conv1.conv.weight.grad[:12], gn["d_enc1_w"], rtol=1e-3, atol=1e-2)
    _close(net.swinViT.patch_embed.proj.weight.grad, gn["d_patch_w"], rtol=2e-3, atol=2e-2)
    _close(net.swinViT.layers4[0].blocks[0].mlp.linear1.weight.grad[:96, :96], gn["d_l4_fc"], rtol=2e-3, atol=2e-2)


def test_unetrc_oracle_vs_reference_class(golden_dir):
    """oracle/unetrc.py (restatement of the reference's UNETR conv decoder, /root/reference/models/segmentors/unetr.py:9-52,
    195-289) against the reference class itself: same state-dict keys, training-mode logits, gradient probes, BatchNorm
    running statistics, eval-mode logits"""
    from oracle.unetrc import UNETRC
    from tests.golden_util import UNETRC_PROBES, ToyTokenEncoder, probe
    g = _load(golden_dir, "unetrc_ref.npz")
    torch.set_num_threads(8)
    net = UNETRC(ToyTokenEncoder(1, 48, (32, 32, 32), (16, 16, 16)), in_chans=1, output_dim=2)
    det_fill_(net, "unetrc.")
    net.train()
    x = det_tensor("unetrc_x", (2, 1, 32, 32, 32))
    y = net(x)
    assert np.allclose(y.detach().numpy(), g["logits"], rtol=1e-4, atol=1e-4)
    (y * det_tensor("unetrc_r", tuple(y.shape))).sum().backward()
    params = dict(net.named_parameters())
    for k in UNETRC_PROBES:
        want = g["g:" + k]
        got = probe(params[k].grad).numpy()
        assert np.abs(got - want).max() <= 2e-4 * max(np.abs(want).max(), 1e-3), k
    bn = net.decoder9_upsampler[1].block[1]
    assert np.allclose(bn.running_mean.numpy(), g["rm"], rtol=1e-4, atol=1e-5)
    assert np.allclose(bn.running_var.numpy(), g["rv"], rtol=1e-4, atol=1e-5)
    assert float(bn.num_batches_tracked) == float(g["nbt"])
    net.eval()
    with torch.no_grad():
        assert np.allclose(net(x).numpy(), g["logits_eval"], rtol=1e-4, atol=1e-4)


def test_swindepth_oracle_vs_reference_file(golden_dir):
    """oracle SwinDepth (oracle/swin.py with the depthwise-conv + BatchNorm MLP) against the reference's own
    models/backbones/swindepth.py: training-mode features, gradients, running statistics, eval-mode features"""
    g = _load(golden_dir, "swindepth_encoder.npz")
    vol = (24, 24, 24)
    m = osw.SwinTransformerNNFormer(vol, (2, 2, 2), 1, 32, (2, 2), (2, 4), (6, 3), mlp="depth")
    det_fill_(m, "sd")
    m.train()
    x = det_tensor("sd_x", (2, 1) + vol).requires_grad_(True)
    outs = m((x, None, None))
    for i, o in enumerate(outs):
        assert np.allclose(o.detach().numpy(), g[f"out{i}"], rtol=1e-3, atol=2e-4), i
    sum((o * det_tensor(f"sd_r{i}", o.shape)).sum() for i, o in enumerate(outs)).backward()
    mlp = m.layers[0].blocks[1].mlp
    rel = lambda a, b: float(np.abs(a.detach().numpy() - b).max() / max(np.abs(b).max(), 1e-6))
    assert rel(x.grad, g["dx"]) < 2e-3
    assert rel(mlp.dwc2.weight.grad, g["d_dwc2_w"]) < 2e-3 and rel(mlp.bn2.weight.grad, g["d_bn2_w"]) < 2e-3
    assert rel(mlp.fc1.weight.grad, g["d_fc1_w"]) < 2e-3
    assert np.allclose(mlp.bn3.running_mean.numpy(), g["rm"], atol=1e-5) and np.allclose(mlp.bn3.running_var.numpy(), g["rv"], atol=1e-5)
    m.eval()
    with torch.no_grad():
        for i, o in enumerate(m((x.detach(), None, None))):
            assert np.allclose(o.numpy(), g[f"eval{i}"], rtol=1e-3, atol=2e-4), i


def test_swinception_oracle_vs_reference_file(golden_dir):
    """oracle SwInception (oracle/swin.py with the Inception-head MLP: 4- / 25- / 8- / 51-channel Conv3d + BatchNorm3d + GELU
    branches, average-pool branch, concat + Linear) against the reference's own models/backbones/swinception.py:
    training-mode features, gradient probes, running statistics, eval-mode features"""
    g = _load(golden_dir, "swinception_encoder.npz")
    vol = (24, 24, 24)
    m = osw.SwinTransformerNNFormer(vol, (2, 2, 2), 1, 32, (2, 2), (2, 4), (6, 3), mlp="inception")
    det_fill_(m, "si")
    m.train()
    x = det_tensor("si_x", (2, 1) + vol).requires_grad_(True)
    outs = m((x, None, None))
    for i, o in enumerate(outs):
        assert np.allclose(o.detach().numpy(), g[f"out{i}"], rtol=1e-3, atol=2e-4), i
    sum((o * det_tensor(f"si_r{i}", o.shape)).sum() for i, o in enumerate(outs)).backward()
    mlp = m.layers[0].blocks[1].mlp
    b = mlp.branches
    rel = lambda a, c: float(np.abs(a.detach().numpy() - c).max() / max(np.abs(c).max(), 1e-6))
    probes = dict(dx=x.grad, d_b1_w=b[0].branch1x1.conv.weight.grad, d_b3_2_w=b[1].branch3x3_2.conv.weight.grad,
                  d_b5_2_w=b[2].branch3x3dbl_2.conv.weight.grad, d_b7_1_w=b[3].branch3x3trpl_1.conv.weight.grad,
                  d_b7_4_bn_w=b[3].branch3x3trpl_4.bn.weight.grad, d_b7_4_bn_b=b[3].branch3x3trpl_4.bn.bias.grad,
                  d_pool_w=b[4].branch_pool_2.conv.weight.grad, d_fc_w=mlp.fc.weight.grad, d_fc_b=mlp.fc.bias.grad,
                  d_fc_w_l1=m.layers[1].blocks[0].mlp.fc.weight.grad)
    errs = {k: rel(v, g[k]) for k, v in probes.items()}
    assert max(errs.values()) < 2e-3, errs
    bn = b[2].branch3x3dbl_3.bn
    assert np.allclose(bn.running_mean.numpy(), g["rm"], atol=1e-5) and np.allclose(bn.running_var.numpy(), g["rv"], atol=1e-5)
    m.eval()
    with torch.no_grad():
        for i, o in enumerate(m((x.detach(), None, None))):
            assert np.allclose(o.numpy(), g[f"eval{i}"], rtol=1e-3, atol=2e-4), i


def test_segformer3d_oracle_vs_reference_files(golden_dir):
    """oracle/segformer.py against the reference's own MixVisionTransformer + SegFormerHeadOfficial (training-mode logits,
    encoder features, gradient probes, BatchNorm running statistics, eval-mode logits)"""
    from oracle import segformer as OS
    from tests.golden_util import SEGFORMER_CFG as c, probe
    g = _load(golden_dir, "segformer3d_ref.npz")
    torch.set_num_threads(8)
    enc = OS.MixVisionTransformer(1, c["embed_dim"], c["num_heads"], (4, 4, 4, 4), True, c["depths"], (8, 4, 2, 1))
    net = OS.SegFormerHeadOfficial(enc, [c["embed_dim"] * 2 ** i for i in range(4)], c["classes"], 0.0, c["embedding_dim"])
    det_fill_(net, "segf.")
    net.train()
    x = det_tensor("segf_x", (2, 1) + c["vol"])
    sub = lambda t: t[:, :, ::2, ::2, ::2]
    feats = enc((x, None, None))
    assert np.allclose(sub(feats[1]).detach().numpy(), g["feat1_s2"], rtol=1e-3, atol=1e-4)
    for i in (2, 3, 4):
        assert np.allclose(feats[i].detach().numpy(), g[f"feat{i}"], rtol=1e-3, atol=1e-4), i
    y = net((x, None, None))
    assert np.allclose(sub(y).detach().numpy(), g["logits_s2"], rtol=1e-3, atol=1e-4)
    (y * det_tensor("segf_r", tuple(y.shape))).sum().backward()
    P = dict(net.named_parameters())
    for k in [k[2:] for k in g.files if k.startswith("g:")]:
        want = g["g:" + k]
        got = probe(P[k].grad).numpy()
        assert np.abs(got - want).max() <= 2e-3 * max(np.abs(want).max(), 1e-3), k
    assert np.allclose(net.linear_fuse.bn.running_mean.numpy(), g["rm"], atol=1e-5)
    assert np.allclose(net.linear_fuse.bn.running_var.numpy(), g["rv"], atol=1e-5)
    net.eval()
    with torch.no_grad():
        assert np.allclose(sub(net((x, None, None))).numpy(), g["logits_eval_s2"], rtol=1e-3, atol=1e-4)


def test_swin_segformer_oracle_vs_reference_files(golden_dir):
    """oracle/segformer.py SegFormerHead around oracle/swin.py's encoder against the reference's own 'SwinSegFormer' wiring
    (models/model_builder.py:173-189; tests/golden/swin_segformer_ref.npz): training-mode logits, gradient probes, the four
    BatchNorms' running statistics, eval-mode logits"""
    from oracle import segformer as OS, swin as O
    from tests.golden_util import SWIN_SEGFORMER_CFG as c, probe
    g = _load(golden_dir, "swin_segformer_ref.npz")
    torch.set_num_threads(8)
    enc = O.SwinTransformerNNFormer(c["vol"], patch_size=(2, 2, 2), in_chans=1, embed_dim=c["embed_dim"], depths=tuple(c["depths"]),
                                    num_heads=tuple(c["num_heads"]), window_size=tuple(c["window_size"]))
    net = OS.SegFormerHead(enc, [c["embed_dim"] * 2 ** i for i in range(5)], c["classes"], 0.0, c["embedding_dim"])
    det_fill_(net, "swsf.")
    net.train()
    x = det_tensor("swsf_x", (2, 1) + c["vol"])
    y = net((x, None, None))
    assert np.allclose(y.detach().numpy(), g["logits"], rtol=1e-3, atol=1e-4)
    (y * det_tensor("swsf_r", tuple(y.shape))).sum().backward()
    P = dict(net.named_parameters())
    for k in c["probes"]:
        want = g["g:" + k]
        got = probe(P[k].grad).numpy()
        assert np.abs(got - want).max() <= 2e-3 * max(np.abs(want).max(), 1e-3), k
    for i in range(4):
        bn = getattr(net, f"linear_fuse_{i}").bn
        assert np.allclose(bn.running_mean.numpy(), g[f"rm{i}"], atol=1e-5) and np.allclose(bn.running_var.numpy(), g[f"rv{i}"], atol=1e-5)
    net.eval()
    with torch.no_grad():
        assert np.allclose(net((x, None, None)).numpy(), g["logits_eval"], rtol=1e-3, atol=1e-4)
