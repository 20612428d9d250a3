"""Generate ``tests/golden/*.npz`` by running the REFERENCE's own modules on CPU.

Runs only in the build container (needs ``/root/reference``); the GPU box only sees the
committed fixtures.  Nothing from the reference is copied: its modules are imported from
where they lie, evaluated on deterministic inputs/weights (``tests/golden_util.py``), and only
the resulting arrays are stored.

``timm`` and ``monai`` are not installed here.  The reference's backbones use four timm names
(``DropPath``, ``to_3tuple``, ``trunc_normal_``, ``to_2tuple``) for which the reference vendors its
own copies under ``models/layers``; an in-memory ``timm.models.layers`` module maps to those.
``monai.utils.ensure_tuple_rep`` is a pure tuple helper; the MONAI block classes named by the
imports are never instantiated on this path and get inert placeholders.

    python oracle/gen_golden.py            # rewrites tests/golden/
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
from tests.golden_util import (SEGFORMER_CFG, SW_CASES, SWIN_SEGFORMER_CFG, UNETRC_PROBES, ToyTokenEncoder, det_fill_,  # noqa: E402
                               det_tensor, probe, sw_predictor)


def _install_import_shims():
    sys.path.insert(0, REF)
    from models.layers.drop_path import DropPath
    from models.layers.weight_init import trunc_normal_

    def _ntuple(n):
        def f(x):
            return tuple(x) if isinstance(x, (tuple, list)) else (x,) * n
        return f

    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tl = types.ModuleType("timm.models.layers")
    tl.DropPath, tl.trunc_normal_ = DropPath, trunc_normal_
    tl.to_2tuple, tl.to_3tuple = _ntuple(2), _ntuple(3)
    timm.models, tm.layers = tm, tl
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})

    def ensure_tuple_rep(tup, dim):
        if isinstance(tup, (tuple, list)):
            if len(tup) != dim:
                raise ValueError("bad length")
            return tuple(tup)
        return (tup,) * dim

    class _Inert:  # named by imports, never instantiated on this path
        def __getitem__(self, k):
            raise RuntimeError("inert placeholder")

    names = ["monai", "monai.utils", "monai.networks", "monai.networks.layers", "monai.networks.blocks",
             "monai.networks.blocks.unetr_block"]
    mods = {n: types.ModuleType(n) for n in names}
    mods["monai.utils"].ensure_tuple_rep = ensure_tuple_rep
    mods["monai.networks.layers"].Conv = _Inert()
    mods["monai.networks.blocks.unetr_block"].UnetrBasicBlock = _Inert
    sys.modules.update(mods)


def _save(name, **arrays):
    out = os.path.join(REPO, "tests", "golden", name)
    np.savez_compressed(out, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                for k, v in arrays.items()})
    print("wrote", out, {k: tuple(v.shape) for k, v in arrays.items()})


def gen_window_attention(ref):
    for tag, dim, ws, heads, nwin in [("h3w6", 48, 6, 3, 8), ("h24w3", 384, 3, 24, 8)]:
        torch.manual_seed(0)
        m = ref.WindowAttention(dim, window_size=(ws,) * 3, num_heads=heads, qkv_bias=True)
        det_fill_(m, "attn_" + tag)
        N = ws ** 3
        x = det_tensor("attn_x_" + tag, (nwin, N, dim)).requires_grad_(True)
        r = det_tensor("attn_r_" + tag, (nwin, N, dim))
        # a region mask like BasicLayer's: windows of a (2ws)^3 volume, shift ws//2
        from oracle.swin import shift_region_mask
        mask = shift_region_mask(2 * ws, 2 * ws, 2 * ws, ws, ws // 2)
        res = {}
        for mk, msk in (("nomask", None), ("mask", mask)):
            y, _ = m(x, mask=msk)
            g, = torch.autograd.grad((y * r).sum(), x)
            pg = torch.autograd.grad((m(x, mask=msk)[0] * r).sum(),
                                     [m.qkv.weight, m.relative_position_bias_table])
            res.update({f"y_{mk}": y, f"dx_{mk}": g, f"dtable_{mk}": pg[1]})
            if dim <= 48:
                res[f"dqkvw_{mk}"] = pg[0]
        _save(f"swin_attn_{tag}.npz", **res)


def gen_block(ref):
    dim, heads, ws, res = 48, 3, 6, (12, 12, 12)
    # the layer's shifted-window mask, as BasicLayer.forward builds it
    from oracle.swin import shift_region_mask
    mask = shift_region_mask(12, 12, 12, ws, ws // 2)
    out = {}
    for shift in (0, 3):
        m = ref.SwinTransformerBlock(dim, res, heads, window_size=ws, shift_size=shift, drop_path=0.0)
        det_fill_(m, "blk")
        x = det_tensor("blk_x", (2, 12 ** 3, dim)).requires_grad_(True)
        r = det_tensor("blk_r", (2, 12 ** 3, dim))
        y, _ = m(x, mask)
        g, = torch.autograd.grad((y * r).sum(), x)
        out[f"y_shift{shift}"] = y
        if shift:
            out[f"dx_shift{shift}"] = g
    _save("swin_block.npz", **out)


def gen_basic_layer_mask(ref):
    """Capture the mask BasicLayer.forward builds by intercepting the first block."""
    layer = ref.BasicLayer(dim=48, input_resolution=(10, 12, 12), depth=2, num_heads=3, window_size=6,
                           drop_path=0.0, downsample=ref.PatchMerging)
    seen = {}
    orig = layer.blocks[0].forward

    def spy(x, mask_matrix, **kw):
        seen["mask"] = mask_matrix.clone()
        return orig(x, mask_matrix, **kw)
    layer.blocks[0].forward = spy
    det_fill_(layer, "layer")
    x = det_tensor("layer_x", (1, 10 * 12 * 12, 48))
    xo, S, H, W, xd, Ws, Wh, Ww, _ = layer(x, 10, 12, 12)
    _save("swin_layer.npz", mask=seen["mask"].to(torch.int8), x_out=xo, x_down=xd, dims=np.array([S, H, W, Ws, Wh, Ww]))


def gen_encoder(ref):
    for tag, vol in (("v24", (24, 24, 24)), ("v20", (20, 20, 20))):
        m = ref.SwinTransformerNNFormer(pretrain_img_size=vol, patch_size=(2, 2, 2), in_chans=1, embed_dim=32,
                                        depths=[2, 2], num_heads=[2, 4], window_size=[6, 3],
                                        drop_path_rate=0.0)
        m.eval()
        det_fill_(m, "enc")
        x = det_tensor("enc_x_" + tag, (2, 1) + vol).requires_grad_(True)
        outs = m((x, None, None))
        loss = sum((o * det_tensor(f"enc_r{i}_" + tag, o.shape)).sum() for i, o in enumerate(outs))
        loss.backward()
        gw = m.layers[0].blocks[1].attn.qkv.weight.grad
        gt = m.layers[0].blocks[1].attn.relative_position_bias_table.grad
        gm = m.layers[1].downsample.reduction.weight.grad
        _save(f"swin_encoder_{tag}.npz", dx=x.grad, d_qkv_w=gw, d_table=gt, d_merge_w=gm[:8],
              **{f"out{i}": o for i, o in enumerate(outs)})


def gen_swindepth():
    """the reference's SwinDepth encoder (models/backbones/swindepth.py:400-691; learned class vectors off, as build_model
    passes cfg.learned_cls_vectors = False): depthwise-conv + BatchNorm MLP in TRAINING mode (batch statistics, running
    statistics updated), stochastic depth 0; features + gradients, then the eval-mode features on the updated statistics"""
    import models.backbones.swindepth as SD
    vol = (24, 24, 24)
    m = SD.SwinDepth(pretrain_img_size=vol, patch_size=(2, 2, 2), in_chans=1, embed_dim=32, depths=[2, 2],
                     num_heads=[2, 4], window_size=[6, 3], drop_path_rate=0.0, use_learned_cls_vectors=False,
                     out_indices=(0, 1))
    det_fill_(m, "sd")
    m.train()
    x = det_tensor("sd_x", (2, 1) + vol).requires_grad_(True)
    outs = m((x, None, None))
    loss = sum((o * det_tensor(f"sd_r{i}", o.shape)).sum() for i, o in enumerate(outs))
    loss.backward()
    mlp = m.layers[0].blocks[1].mlp
    out = dict(dx=x.grad, d_dwc2_w=mlp.dwc2.weight.grad, d_dwc2_b=mlp.dwc2.bias.grad, d_bn2_w=mlp.bn2.weight.grad,
               d_bn2_b=mlp.bn2.bias.grad, d_fc1_w=mlp.fc1.weight.grad, d_fc2_w=m.layers[1].blocks[0].mlp.fc2.weight.grad,
               rm=mlp.bn3.running_mean, rv=mlp.bn3.running_var, **{f"out{i}": o for i, o in enumerate(outs)})
    m.eval()
    with torch.no_grad():
        for i, o in enumerate(m((x.detach(), None, None))):
            out[f"eval{i}"] = o
    _save("swindepth_encoder.npz", **out)


def gen_swinception():
    """the reference's SwInception encoder (models/backbones/swinception.py:609-791; learned class vectors off): Inception
    head MLP (1x1 / 3x3 / double / triple 3x3 / average-pool branches of Conv3d + BatchNorm3d + GELU, 4- and 25- / 8- and
    51-channel convs at embed_dim 32) in TRAINING mode, stochastic depth 0; features + gradient probes + running statistics,
    then the eval-mode features on the updated statistics"""
    import models.backbones.swinception as SI
    vol = (24, 24, 24)
    m = SI.SwInception(pretrain_img_size=vol, patch_size=(2, 2, 2), in_chans=1, embed_dim=32, depths=[2, 2],
                       num_heads=[2, 4], window_size=[6, 3], drop_path_rate=0.0, use_learned_cls_vectors=False,
                       out_indices=(0, 1))
    det_fill_(m, "si")
    m.train()
    x = det_tensor("si_x", (2, 1) + vol).requires_grad_(True)
    outs = m((x, None, None))
    loss = sum((o * det_tensor(f"si_r{i}", o.shape)).sum() for i, o in enumerate(outs))
    loss.backward()
    mlp = m.layers[0].blocks[1].mlp
    b = mlp.branches
    out = dict(dx=x.grad, d_b1_w=b[0].branch1x1.conv.weight.grad, d_b3_2_w=b[1].branch3x3_2.conv.weight.grad,
               d_b5_2_w=b[2].branch3x3dbl_2.conv.weight.grad, d_b7_1_w=b[3].branch3x3trpl_1.conv.weight.grad,
               d_b7_4_bn_w=b[3].branch3x3trpl_4.bn.weight.grad, d_b7_4_bn_b=b[3].branch3x3trpl_4.bn.bias.grad,
               d_pool_w=b[4].branch_pool_2.conv.weight.grad, d_fc_w=mlp.fc.weight.grad, d_fc_b=mlp.fc.bias.grad,
               d_fc_w_l1=m.layers[1].blocks[0].mlp.fc.weight.grad,
               rm=b[2].branch3x3dbl_3.bn.running_mean, rv=b[2].branch3x3dbl_3.bn.running_var,
               **{f"out{i}": o for i, o in enumerate(outs)})
    m.eval()
    with torch.no_grad():
        for i, o in enumerate(m((x.detach(), None, None))):
            out[f"eval{i}"] = o
    _save("swinception_encoder.npz", **out)


def gen_segformer3d():
    """the reference's SegFormer3D (models/backbones/segformer_backbone.py MixVisionTransformer + models/segmentors/
    segformer_head_official.py SegFormerHeadOfficial, wired as model_builder.py:190-205) in TRAINING mode (BatchNorm batch
    statistics; dropout 0 and stochastic depth 0 so that the pass is deterministic): logits, encoder features, gradients;
    then the eval-mode logits on the updated running statistics"""
    import models.backbones.segformer_backbone as SB
    import models.segmentors.segformer_head_official as SH
    c = SEGFORMER_CFG
    enc = SB.MixVisionTransformer(img_size=c["vol"], patch_size=16, in_chans=1, embed_dim=c["embed_dim"], depths=c["depths"],
                                  num_heads=c["num_heads"], sr_ratios=[8, 4, 2, 1], qkv_bias=True)
    net = SH.SegFormerHeadOfficial(encoder=enc, in_channels=[c["embed_dim"] * 2 ** i for i in range(4)],
                                   num_classes=c["classes"], dropout_ratio=0.0, embedding_dim=c["embedding_dim"])
    det_fill_(net, "segf.")
    net.train()
    x = det_tensor("segf_x", (2, 1) + c["vol"])
    feats = enc((x, None, None))
    y = net((x, None, None))
    (y * det_tensor("segf_r", tuple(y.shape))).sum().backward()
    P = dict(net.named_parameters())
    keys = ["encoder.patch_embed1.proj.weight", "encoder.block1.0.attn.q.weight", "encoder.block1.0.attn.kv.weight",
            "encoder.block1.0.attn.sr.weight", "encoder.block1.1.mlp.dwconv.dwconv.weight", "encoder.block2.0.attn.sr.bias",
            "encoder.block4.0.attn.kv.bias", "encoder.patch_embed3.proj.weight", "encoder.norm2.weight",
            "linear_c4.proj.weight", "linear_c1.proj.bias", "linear_fuse.conv.weight", "linear_fuse.bn.weight",
            "linear_pred.weight", "linear_pred.bias"]
    sub = lambda t: t[:, :, ::2, ::2, ::2].contiguous()        # fixtures stay small: every second voxel per axis
    out = dict(logits_s2=sub(y), feat1_s2=sub(feats[1]), feat2=feats[2], feat3=feats[3], feat4=feats[4],
               **{"g:" + k: probe(P[k].grad) for k in keys})
    out["rm"], out["rv"] = net.linear_fuse.bn.running_mean, net.linear_fuse.bn.running_var
    net.eval()
    with torch.no_grad():
        out["logits_eval_s2"] = sub(net((x, None, None)))
    _save("segformer3d_ref.npz", **out)


def gen_swin_segformer(ref):
    """the reference's 'SwinSegFormer' branch (models/model_builder.py:173-189): SwinTransformerNNFormer encoder +
    models/segmentors/segformer_head.py SegFormerHead (progressive fusion of the five feature maps: Linear per map,
    trilinear x2, Conv3d 1x1x1 + BatchNorm3d(eps 1e-3) + GELU per level, Dropout3d, prediction conv at full resolution) in
    TRAINING mode with dropout 0 and stochastic depth 0: logits, gradient probes, running statistics, eval logits"""
    import models.segmentors.segformer_head as SH
    c = SWIN_SEGFORMER_CFG
    enc = ref.SwinTransformerNNFormer(pretrain_img_size=c["vol"], patch_size=(2, 2, 2), in_chans=1, embed_dim=c["embed_dim"],
                                      depths=c["depths"], num_heads=c["num_heads"], window_size=c["window_size"],
                                      qkv_bias=True, drop_path_rate=0.0)
    net = SH.SegFormerHead(encoder=enc, in_channels=[c["embed_dim"] * 2 ** i for i in range(5)], num_classes=c["classes"],
                           dropout_ratio=0.0, embedding_dim=c["embedding_dim"])
    det_fill_(net, "swsf.")
    net.train()
    x = det_tensor("swsf_x", (2, 1) + c["vol"])
    y = net((x, None, None))
    (y * det_tensor("swsf_r", tuple(y.shape))).sum().backward()
    P = dict(net.named_parameters())
    out = dict(logits=y, **{"g:" + k: probe(P[k].grad) for k in c["probes"]})
    for i in range(4):
        bn = getattr(net, f"linear_fuse_{i}").bn
        out[f"rm{i}"], out[f"rv{i}"] = bn.running_mean, bn.running_var
    net.eval()
    with torch.no_grad():
        out["logits_eval"] = net((x, None, None))
    _save("swin_segformer_ref.npz", **out)


def gen_param_order(ref):
    """ordered (name, shape) lists of `named_parameters()` of the reference's own model classes: what torch.optim.AdamW's
    state dict indexes by position (the reference's checkpoints, utils/misc.py:268-283).  optim.FlatAdamW maps such a
    state onto this build's modules by position, so their parameter ORDER must be the reference's."""
    import json
    import models.backbones.segformer_backbone as SB
    import models.backbones.swindepth as SD
    import models.backbones.swinception as SI
    import models.segmentors.segformer_head as SH
    import models.segmentors.segformer_head_official as SHO
    import models.segmentors.unetr as UN
    kw = dict(pretrain_img_size=(32, 32, 32), patch_size=(2, 2, 2), in_chans=1, embed_dim=16, depths=[2, 2], num_heads=[1, 2],
              window_size=[4, 4])
    fams = {"swin_nnformer": ref.SwinTransformerNNFormer(**kw),
            "swindepth": SD.SwinDepth(**kw, use_learned_cls_vectors=False, out_indices=(0, 1)),
            "swinception": SI.SwInception(**kw, use_learned_cls_vectors=False, out_indices=(0, 1))}
    enc = SB.MixVisionTransformer(img_size=64, patch_size=16, in_chans=1, embed_dim=32, depths=[1, 1, 1, 1],
                                  num_heads=[1, 2, 4, 8], sr_ratios=[8, 4, 2, 1], qkv_bias=True)
    fams["segformer3d"] = SHO.SegFormerHeadOfficial(encoder=enc, in_channels=[32, 64, 128, 256], num_classes=3, embedding_dim=64)
    c = SWIN_SEGFORMER_CFG
    enc = ref.SwinTransformerNNFormer(pretrain_img_size=c["vol"], patch_size=(2, 2, 2), in_chans=1, embed_dim=c["embed_dim"],
                                      depths=c["depths"], num_heads=c["num_heads"], window_size=c["window_size"], qkv_bias=True)
    fams["swin_segformer"] = SH.SegFormerHead(encoder=enc, in_channels=[c["embed_dim"] * 2 ** i for i in range(5)],
                                              num_classes=c["classes"], embedding_dim=c["embedding_dim"])
    fams["unetrc"] = UN.UNETRC(ToyTokenEncoder(1, 48, (32, 32, 32), (16, 16, 16)), in_chans=1, output_dim=2)
    out = {k: [[n, list(p.shape)] for n, p in m.named_parameters()] for k, m in fams.items()}
    path = os.path.join(REPO, "tests", "golden", "param_order.json")
    with open(path, "w") as fh:
        json.dump(out, fh)
    print("wrote", path, {k: len(v) for k, v in out.items()})


def gen_lr_and_misc():
    from models.optimizers.lr_scheduler import LinearWarmupCosineAnnealingLR
    import utils.misc as misc
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=4e-4)
    sch = LinearWarmupCosineAnnealingLR(opt, warmup_epochs=40, max_epochs=200)
    lrs = []
    for _ in range(200):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    aff = det_tensor("aff", (3, 4, 4))
    t = {"orig_size": [torch.tensor([100., 120.]), torch.tensor([110., 90.]), torch.tensor([64., 80.])],
         "extra_info": {"center": [torch.tensor([10., 30.]), torch.tensor([55., 45.]), torch.tensor([32., 8.])]}}
    _save("lr_misc.npz", lrs=np.array(lrs, dtype=np.float64), aff_in=aff, aff_xyz=misc.get_affine_xyz(aff),
          rel_crop=misc.get_rel_crop_loc(t))
    vol = (np.arange(5 * 6 * 7) % 4).reshape(5, 6, 7).astype(np.uint8)
    _save("resample.npz", vol=vol, out=misc.resample_3d(vol, (9, 6, 11)))


def gen_unetr_conv_blocks():
    """conv / transposed-conv blocks of the reference's own UNETR decoder (models/segmentors/unetr.py:9-52, pure torch):
    SingleConv3DBlock (Conv3d k3 p1 + bias), SingleDeconv3DBlock (ConvTranspose3d k2 s2 + bias) and Deconv3DBlock in eval
    mode (deconv -> conv -> BatchNorm3d with its initial running statistics -> ReLU), forward and gradients."""
    import models.segmentors.unetr as U
    x = det_tensor("unetr_x", (2, 16, 12, 12, 12)).requires_grad_(True)
    conv = U.SingleConv3DBlock(16, 32, 3)
    det_fill_(conv, "unetr_conv.")
    y = conv(x)
    r = det_tensor("unetr_r", tuple(y.shape))
    (y * r).sum().backward()
    out = dict(conv_y=y, conv_dx=x.grad.clone(), conv_dw=conv.block.weight.grad, conv_db=conv.block.bias.grad)
    x2 = det_tensor("unetr_x2", (2, 32, 6, 6, 6)).requires_grad_(True)
    dec = U.SingleDeconv3DBlock(32, 16)
    det_fill_(dec, "unetr_deconv.")
    y2 = dec(x2)
    r2 = det_tensor("unetr_r2", tuple(y2.shape))
    (y2 * r2).sum().backward()
    out.update(deconv_y=y2, deconv_dx=x2.grad.clone(), deconv_dw=dec.block.weight.grad, deconv_db=dec.block.bias.grad)
    blk = U.Deconv3DBlock(32, 16).eval()
    det_fill_(blk, "unetr_block.")
    with torch.no_grad():
        out["block_y"] = blk(x2.detach())
    _save("unetr_blocks.npz", **out)


def gen_unetrc():
    """the reference's own UNETRC (models/segmentors/unetr.py:195-289, pure torch) around a toy token encoder: training
    mode (BatchNorm3d batch statistics + running-statistics update), logits, probes of the gradients of decoder and encoder
    parameters and of the input-side token maps' producer, then an eval-mode forward on the updated running statistics"""
    import models.segmentors.unetr as U
    net = U.UNETRC(ToyTokenEncoder(1, 48, (32, 32, 32), (16, 16, 16)), in_chans=1, output_dim=2)
    det_fill_(net, "unetrc.")
    net.train()
    x = det_tensor("unetrc_x", (2, 1, 32, 32, 32))
    y = net(x)
    r = det_tensor("unetrc_r", tuple(y.shape))
    (y * r).sum().backward()
    params = dict(net.named_parameters())
    out = dict(logits=y)
    for k in UNETRC_PROBES:
        out["g:" + k] = probe(params[k].grad)
    bn = net.decoder9_upsampler[1].block[1]
    out["rm"], out["rv"], out["nbt"] = bn.running_mean, bn.running_var, bn.num_batches_tracked.float()
    net.eval()
    with torch.no_grad():
        out["logits_eval"] = net(x)
    _save("unetrc_ref.npz", **out)


def gen_sliding_window_loop():
    """The reference's OWN loop (engine/utils.py:19-159: padding, window order, `centers`, blend, final slicing) run with the
    four MONAI helper names it imports bound in memory to oracle/sliding_window.py's restatements (MONAI is absent)."""
    import enum
    from oracle import sliding_window as osw

    class BlendMode(enum.Enum):
        CONSTANT = "constant"
        GAUSSIAN = "gaussian"

    class PytorchPadMode(enum.Enum):
        CONSTANT = "constant"
        REFLECT = "reflect"
        REPLICATE = "replicate"
        CIRCULAR = "circular"

    def look_up_option(opt, supported):
        return opt if isinstance(opt, supported) else supported(opt)

    mods = {n: types.ModuleType(n) for n in ["monai.data", "monai.data.utils", "monai.inferers", "monai.inferers.utils"]}
    mods["monai.data.utils"].compute_importance_map = lambda ps, mode="constant", sigma_scale=0.125, device=None: \
        osw.compute_importance_map(ps, getattr(mode, "value", mode), sigma_scale)
    mods["monai.data.utils"].dense_patch_slices = osw.dense_patch_slices
    mods["monai.data.utils"].get_valid_patch_size = osw.get_valid_patch_size
    mods["monai.inferers.utils"]._get_scan_interval = osw.get_scan_interval
    sys.modules.update(mods)
    mu = sys.modules["monai.utils"]
    mu.BlendMode, mu.PytorchPadMode, mu.look_up_option = BlendMode, PytorchPadMode, look_up_option
    mu.fall_back_tuple = osw.fall_back_tuple
    mu.optional_import = lambda *a, **k: (None, False)
    import importlib
    ref_utils = importlib.import_module("engine.utils")
    out = {}
    for tag, vol, roi, sb, ov, mode, cval in SW_CASES:
        x = det_tensor("sw_x_" + tag, vol)
        aff = det_tensor("sw_aff_" + tag, (vol[0], 3))
        y = ref_utils.sliding_window_inference(x, aff, roi, sb, sw_predictor, overlap=ov, mode=mode, cval=cval)
        out["out_" + tag] = y
    _save("sliding_window_ref.npz", **out)


def gen_layers():
    """the reference's vendored DropPath (models/layers/drop_path.py:15-45) and trunc_normal_ (weight_init.py:17-64) under
    fixed CPU seeds"""
    from models.layers.drop_path import DropPath
    from models.layers.weight_init import trunc_normal_
    x = det_tensor("dp_x", (16, 5, 7))
    dp = DropPath(0.2).train()
    torch.manual_seed(7)
    y = dp(x)
    dp_eval = DropPath(0.2).eval()(x)
    torch.manual_seed(3)
    t = trunc_normal_(torch.empty(64, 48), std=0.02)
    torch.manual_seed(4)
    t2 = trunc_normal_(torch.empty(257), mean=0.5, std=1.0, a=-1.0, b=2.0)
    _save("layers_ref.npz", dp_x=x, dp_y=y, dp_eval=dp_eval, tn=t, tn2=t2)


def gen_swin_official():
    """The reference's own models/segmentors/swin_unetr_official.py (vendored MONAI SwinUNETR: window 7, Linear patch merging
    with the duplicated sub-grids, index slicing when the window is clamped, un-affine proj_out) with the MONAI block names
    it imports bound to oracle/blocks.py's restatements.  Encoder features + gradients and the whole network's logits."""
    import importlib
    from oracle import blocks as ob

    class _Conv:
        CONV = "conv"

        def __getitem__(self, key):
            assert key[0] == "conv" and key[1] == 3
            return torch.nn.Conv3d

    def basic(spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name, res_block=True):
        assert spatial_dims == 3 and res_block and norm_name == "instance"
        return ob.UnetrBasicBlock(in_channels, out_channels, kernel_size, stride)

    def up(spatial_dims, in_channels, out_channels, kernel_size, upsample_kernel_size, norm_name, res_block=True):
        assert spatial_dims == 3 and res_block and norm_name == "instance"
        return ob.UnetrUpBlock(in_channels, out_channels, kernel_size, upsample_kernel_size)

    def outb(spatial_dims, in_channels, out_channels):
        return ob.UnetOutBlock(in_channels, out_channels)

    def optional_import(module, name="", **kw):
        try:
            m = importlib.import_module(module)
            return (getattr(m, name) if name else m), True
        except ImportError:
            return None, False

    def look_up_option(opt, supported, default=None):
        if opt in supported:
            return opt
        raise ValueError(opt)

    sys.modules["monai.networks.blocks"].UnetOutBlock = outb
    sys.modules["monai.networks.blocks"].UnetrBasicBlock = basic
    sys.modules["monai.networks.blocks"].UnetrUpBlock = up
    sys.modules["monai.networks.layers"].Conv = _Conv()
    sys.modules["monai.networks.layers"].get_act_layer = lambda act: torch.nn.GELU()
    sys.modules["monai.utils"].optional_import = optional_import
    sys.modules["monai.utils"].look_up_option = look_up_option
    for m in ("models.blocks.mlp", "models.blocks.patch_embeddings", "models.segmentors.swin_unetr_official"):
        sys.modules.pop(m, None)
    R = importlib.import_module("models.segmentors.swin_unetr_official")
    # (a) encoder at 28^3 (token grid 14 -> 7 -> 4 -> 2 -> 1: padded windows, clamped windows with sliced index)
    vit = R.SwinTransformer(in_chans=1, embed_dim=24, window_size=(7, 7, 7), patch_size=(2, 2, 2), depths=(2, 2, 2, 2),
                            num_heads=(3, 6, 12, 24), spatial_dims=3)
    det_fill_(vit, "swo_vit.")
    x = det_tensor("swo_x28", (1, 1, 28, 28, 28)).requires_grad_(True)
    outs = vit(x, normalize=True)
    loss = sum((o * det_tensor(f"swo_r{i}", o.shape)).sum() for i, o in enumerate(outs))
    loss.backward()
    blk = vit.layers1[0].blocks[1]
    _save("swin_official_encoder.npz", dx=x.grad, d_qkv_w=blk.attn.qkv.weight.grad, d_qkv_b=blk.attn.qkv.bias.grad,
          d_table=blk.attn.relative_position_bias_table.grad, d_merge_w=vit.layers1[0].downsample.reduction.weight.grad,
          d_table_l2=vit.layers2[0].blocks[1].attn.relative_position_bias_table.grad,
          **{f"out{i}": o for i, o in enumerate(outs)})
    # (b) the whole network at 64^3 (the deepest feature map must keep more than one voxel for InstanceNorm), feature size 24 (head dim 8)
    net = R.SwinUNETR(img_size=(64, 64, 64), in_channels=1, out_channels=3, feature_size=24)
    det_fill_(net, "swo_net.")
    x2 = det_tensor("swo_x64", (1, 1, 64, 64, 64))
    y = net(x2)
    (y * det_tensor("swo_ry", y.shape)).sum().backward()
    _save("swin_official_net.npz", logits_s2=y[:, :, ::2, ::2, ::2], logits_sum=y.double().sum(), logits_abs=y.double().abs().sum(),
          d_out_w=net.out.conv.conv.weight.grad,
          d_enc1_w=net.encoder1.layer.conv1.conv.weight.grad[:12],
          d_patch_w=net.swinViT.patch_embed.proj.weight.grad,
          d_l4_fc=net.swinViT.layers4[0].blocks[0].mlp.linear1.weight.grad[:96, :96])


def main():
    """`python oracle/gen_golden.py` rewrites every fixture; `python oracle/gen_golden.py swin_segformer param_order`
    only the named ones (function names without the gen_ prefix)."""
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (build container only)")
    torch.set_num_threads(8)
    _install_import_shims()
    import models.backbones.swin_nnformer as ref
    only = set(sys.argv[1:])
    jobs = [("window_attention", lambda: gen_window_attention(ref)), ("block", lambda: gen_block(ref)),
            ("basic_layer_mask", lambda: gen_basic_layer_mask(ref)), ("encoder", lambda: gen_encoder(ref)),
            ("swindepth", gen_swindepth), ("swinception", gen_swinception), ("segformer3d", gen_segformer3d),
            ("swin_segformer", lambda: gen_swin_segformer(ref)), ("param_order", lambda: gen_param_order(ref)),
            ("lr_and_misc", gen_lr_and_misc), ("unetr_conv_blocks", gen_unetr_conv_blocks), ("unetrc", gen_unetrc),
            ("sliding_window_loop", gen_sliding_window_loop), ("layers", gen_layers),
            ("swin_official", gen_swin_official)]      # swin_official last: it rebinds the MONAI placeholders
    unknown = only - {n for n, _ in jobs}
    if unknown:
        raise SystemExit(f"unknown fixture(s) {sorted(unknown)}")
    for name, fn in jobs:
        if not only or name in only:
            fn()


if __name__ == "__main__":
    main()
