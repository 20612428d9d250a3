"""CPU tests of the host logic that surrounds the kernels: CLI, LR schedule, meters, sliding-window geometry, the
engine loop (driven with the CPU oracle injected as model / criterion), and the N>1 paths on gloo (world_size 2)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arguments_defaults_and_list_collapsing():
    from medicalsemseg_amd.utils.arguments import get_args
    a = get_args([])
    assert a.model == "UNETR_Official" and a.vol_size == 96 and a.patch_size == 16 and a.window_size == 6
    assert a.depths == (2, 2, 2, 2) and a.num_heads == (3, 6, 12, 24) and a.lr == 4e-4 and a.seed == 13
    assert a.pin_mem is True and a.neptune_logging is True and a.gradient_clipping is None and a.backend == "nccl"
    b = get_args("--model UNet --vol_size 64 64 32 --window_size 6 6 6 3 --no_pin_memory --qkv_bias --output_dim 2".split())
    assert b.vol_size == (64, 64, 32) and b.window_size == (6, 6, 6, 3) and b.pin_mem is False and b.qkv_bias


def test_lr_schedule_matches_reference_trace(golden_dir):
    from medicalsemseg_amd.optim import LinearWarmupCosineAnnealingLR
    g = np.load(os.path.join(golden_dir, "lr_misc.npz"))
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=4e-4)
    sch = LinearWarmupCosineAnnealingLR(opt, warmup_epochs=40, max_epochs=200)
    lrs = []
    for _ in range(200):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    np.testing.assert_allclose(np.array(lrs), g["lrs"], rtol=1e-9, atol=1e-15)


def test_misc_helpers_match_reference(golden_dir):
    from medicalsemseg_amd.utils import misc
    g = np.load(os.path.join(golden_dir, "lr_misc.npz"))
    np.testing.assert_array_equal(misc.get_affine_xyz(torch.from_numpy(g["aff_in"])).numpy(), g["aff_xyz"])
    t = {"orig_size": [torch.tensor([100., 120.]), torch.tensor([110., 90.]), torch.tensor([64., 80.])],
         "extra_info": {"center": [torch.tensor([10., 30.]), torch.tensor([55., 45.]), torch.tensor([32., 8.])]}}
    np.testing.assert_allclose(misc.get_rel_crop_loc(t).numpy(), g["rel_crop"])
    m = misc.SmoothedValue(window_size=3)
    for v in (1.0, 2.0, 3.0, 4.0):
        m.update(v)
    assert m.global_avg == 2.5 and m.value == 4.0 and m.max == 4.0 and abs(m.avg - 3.0) < 1e-6


def test_sliding_window_geometry_matches_oracle_and_known_answers():
    from medicalsemseg_amd.engine import utils as P
    from oracle import sliding_window as O
    starts = P.window_starts((512,) * 3, (96,) * 3, P.get_scan_interval((512,) * 3, (96,) * 3, 3, 0.5))
    assert len(starts) == 1000 and starts[-1] == (416, 416, 416) and starts[1] == (0, 0, 48)
    for img, roi, ov in [((100, 70, 96), (96, 64, 96), 0.5), ((130, 130, 130), (64, 64, 64), 0.25), ((96,) * 3, (96,) * 3, 0.5)]:
        iv = P.get_scan_interval(img, roi, 3, ov)
        assert iv == O.get_scan_interval(img, roi, 3, ov)
        assert [tuple(s.start for s in sl) for sl in O.dense_patch_slices(img, roi, iv)] == P.window_starts(img, roi, iv)
    imp = P.importance_map((96, 96, 96), "gaussian", 0.125)
    assert torch.equal(imp, O.compute_importance_map((96, 96, 96), "gaussian", 0.125))
    assert imp.max() == 1.0 and imp[48, 48, 48] == 1.0 and imp.min() > 0
    assert torch.allclose(imp[1:], imp[1:].flip(0)) and torch.allclose(imp[:, 1:], imp[:, 1:].flip(1))
    assert torch.equal(P.importance_map((8, 8, 8), "constant"), torch.ones(8, 8, 8))


def test_oracle_known_answers_dice_ce():
    from oracle.losses import dice_ce_loss, dice_metric
    K, V = 3, 4 * 4 * 4
    labels = torch.zeros(1, 1, 4, 4, 4)
    labels[0, 0, :2] = 1
    labels[0, 0, 3] = 2
    counts = [float((labels == c).sum()) for c in range(K)]
    loss = dice_ce_loss(torch.zeros(1, K, 4, 4, 4), labels, 1e-5, 1e-5)
    dice = np.mean([1 - (2 * vc / K + 1e-5) / (V / K ** 2 + vc + 1e-5) for vc in counts])
    assert abs(float(loss) - (dice + np.log(K))) < 1e-6
    onehot = torch.nn.functional.one_hot(labels.long().squeeze(1), K).movedim(-1, 1).float()
    s, nn_ = dice_metric(onehot * 10, labels)
    assert torch.allclose(s, torch.ones(1, K)) and nn_.sum() == K
    s2, _ = dice_metric(onehot.roll(1, 1) * 10, labels)
    assert torch.all(s2 == 0)
    s3, nn3 = dice_metric(onehot * 10, torch.zeros_like(labels))
    assert torch.isnan(s3[0, 1]) and nn3[0, 1] == 0


class _OracleCriterion(torch.nn.Module):
    """CPU oracle injected into the product engine loop (tests only)."""

    def forward(self, logits, labels):
        from oracle.losses import dice_ce_loss
        return dice_ce_loss(logits, labels)

    def hard_dice(self, logits, labels):
        from oracle.losses import dice_metric
        return dice_metric(logits, labels)


def test_engine_train_loop_with_oracle_model_on_cpu():
    """config 1 plumbing: UNet-small, 64^3... (here 32^3 to stay fast), batch 2, CPU: the engine loop, meters,
    optimizer step and metric bookkeeping run end to end and the loss goes down."""
    from medicalsemseg_amd.data import SyntheticLoader
    from medicalsemseg_amd.engine.train import train_one_epoch
    from medicalsemseg_amd.optim import add_weight_decay
    from medicalsemseg_amd.utils.arguments import get_args
    from oracle.blocks import UNET_FEATURES, BasicUNet
    cfg = get_args("--model UNetSmall --output_dim 2 --vol_size 32".split())
    torch.manual_seed(0)
    model = BasicUNet(1, 2, UNET_FEATURES["UNetSmall"])
    opt = torch.optim.AdamW(add_weight_decay(model, 1e-5), lr=2e-3, betas=(0.9, 0.95), eps=1e-6)
    loader = SyntheticLoader(4, 2, 32, 1, 2, seed=1)
    scaler = torch.amp.GradScaler("cpu", enabled=False)
    s0 = train_one_epoch(model, loader, opt, _OracleCriterion(), torch.device("cpu"), 0, scaler, cfg)
    s1 = train_one_epoch(model, loader, opt, _OracleCriterion(), torch.device("cpu"), 1, scaler, cfg)
    assert set(s0) == {"train/lr", "train/loss", "train/mDice", "train/class0Dice", "train/class1Dice"}
    assert s1["train/loss"] < s0["train/loss"] and 0 <= s1["train/mDice"] <= 1


def test_build_model_surface():
    from medicalsemseg_amd.models.model_builder import build_model
    from medicalsemseg_amd.utils.arguments import get_args
    m = build_model(get_args("--model UNet --output_dim 3".split()))
    assert sum(p.numel() for p in m.parameters()) == 5749443
    sw = build_model(get_args("--model nnFormerUNETR --patch_size 2 --window_size 6 6 6 3 --qkv_bias --output_dim 3".split()))
    n_enc = sum(p.numel() for p in sw.encoder.parameters())
    assert n_enc == 15362430, n_enc   # the reference encoder's parameter count
    sf = build_model(get_args("--model SwinSegFormer --patch_size 2 --window_size 6 6 6 3 --qkv_bias --output_dim 3".split()))
    assert sum(p.numel() for p in sf.encoder.parameters()) == 15362430 and sf.linear_fuse_0.conv.weight.shape == (512, 1024, 1, 1, 1)
    with pytest.raises(NotImplementedError):
        build_model(get_args("--model FocalNetUNETR".split()))
    with pytest.raises(ValueError):
        build_model(get_args([]))   # the reference's default 'UNETR_Official' matches no branch either


def test_parameter_order_matches_reference_classes(golden_dir):
    """optim.FlatAdamW maps a torch.optim.AdamW state (the reference's checkpoints) onto the flat moments BY POSITION, so
    `named_parameters()` of every model family must enumerate in the reference's order (tests/golden/param_order.json:
    ordered names + shapes taken from the reference's own classes by oracle/gen_golden.py).  SwInception computes on
    zero-padded channel counts: its names must match, its shapes are the padded ones (state-dict hooks translate)."""
    import json
    from medicalsemseg_amd.models import segformer3d as PS, swin_unetr as P, unetrc as PC
    from medicalsemseg_amd.optim import add_weight_decay
    from tests.golden_util import SWIN_SEGFORMER_CFG as c, ToyTokenEncoder
    with open(os.path.join(golden_dir, "param_order.json")) as fh:
        ref = json.load(fh)
    kw = dict(patch_size=(2, 2, 2), in_chans=1, embed_dim=16, depths=(2, 2), num_heads=(1, 2), window_size=(4, 4))
    enc = PS.MixVisionTransformer(64, 16, 1, 32, (1, 2, 4, 8), (4, 4, 4, 4), True, 0.0, (1, 1, 1, 1), (8, 4, 2, 1))
    senc = P.SwinTransformerNNFormer(c["vol"], (2, 2, 2), 1, c["embed_dim"], tuple(c["depths"]), tuple(c["num_heads"]),
                                     tuple(c["window_size"]))
    fams = {"swin_nnformer": P.SwinTransformerNNFormer((32,) * 3, **kw), "swindepth": P.SwinDepth((32,) * 3, **kw),
            "swinception": P.SwInception((32,) * 3, **kw),
            "segformer3d": PS.SegFormerHeadOfficial(enc, [32, 64, 128, 256], 3, 0.1, 64),
            "swin_segformer": PS.SegFormerHead(senc, [c["embed_dim"] * 2 ** i for i in range(5)], c["classes"], 0.1, c["embedding_dim"]),
            "unetrc": PC.UNETRC(ToyTokenEncoder(1, 48, (32, 32, 32), (16, 16, 16)), 1, 2)}
    for fam, net in fams.items():
        got = [(n, list(p.shape)) for n, p in net.named_parameters()]
        want = [(n, s) for n, s in ref[fam]]
        assert [n for n, _ in got] == [n for n, _ in want], fam
        if fam != "swinception":
            assert got == want, fam
        # the two AdamW groups (timm add_weight_decay: [no_decay, decay]) then enumerate alike as well
        groups = add_weight_decay(net, 1e-5)
        names = {id(p): n for n, p in net.named_parameters()}
        order = [[names[id(p)] for p in g["params"]] for g in groups]
        wn = [n for n, s in want if len(s) <= 1 or n.endswith(".bias")], [n for n, s in want if not (len(s) <= 1 or n.endswith(".bias"))]
        assert order[0] == wn[0] and order[1] == wn[1], fam


def test_convert_sync_batchnorm_reaches_every_batchnorm_holder():
    """run_training.py under --distributed: /root/reference/run_training.py:83 converts every BatchNorm"""
    from medicalsemseg_amd import layers, parallel
    from medicalsemseg_amd.models import segformer3d as PS, swin_unetr as P, unetrc as PC
    from medicalsemseg_amd.models.unet import UNet
    from tests.golden_util import ToyTokenEncoder
    kw = dict(patch_size=(2, 2, 2), in_chans=1, embed_dim=16, depths=(2, 1), num_heads=(1, 2), window_size=(4, 4))
    net = P.SwinUNETRCustom(P.SwinDepth((32,) * 3, **kw), 1, 3, (32,) * 3, 16, (2, 2, 2))
    assert parallel.convert_sync_batchnorm(net) == 3                      # one _DepthMlp per block
    assert all(m.sync_group is True for m in net.modules() if hasattr(m, "sync_group"))
    assert parallel.convert_sync_batchnorm(net, None) == 3 and all(m.sync_group is None for m in net.modules() if hasattr(m, "sync_group"))
    enc = PS.MixVisionTransformer(64, 16, 1, 32, (1, 2, 4, 8), (4, 4, 4, 4), True, 0.0, (1, 1, 1, 1), (8, 4, 2, 1))
    assert parallel.convert_sync_batchnorm(PS.SegFormerHeadOfficial(enc, [32, 64, 128, 256], 3, 0.1, 64)) == 1
    u = PC.UNETRC(ToyTokenEncoder(1, 48, (32, 32, 32), (16, 16, 16)), 1, 2)
    n = parallel.convert_sync_batchnorm(u)
    bns = [o.op.norm for ops_ in list(u._branch.values()) + list(u._trunk.values()) for o in ops_ if hasattr(o, "op") and hasattr(o.op, "norm")]
    assert bns and all(isinstance(b, layers.BatchNormAct) and b.group is True for b in bns) and n == 1 + len(bns)
    u._build_ops()                                                        # what .to(device) triggers: the switch must survive
    assert all(o.op.norm.group is True for ops_ in u._trunk.values() for o in ops_ if hasattr(o, "op") and hasattr(o.op, "norm"))
    assert parallel.convert_sync_batchnorm(UNet(1, 2)) == 0               # InstanceNorm: nothing to exchange


def test_swin_unetr_state_dict_matches_oracle_layout():
    from medicalsemseg_amd.models import swin_unetr as P
    from oracle import swin as O
    kw = dict(patch_size=(2, 2, 2), in_chans=1, embed_dim=16, depths=(2, 2), num_heads=(1, 2), window_size=(4, 4))
    ref = O.SwinUNETRCustom(O.SwinTransformerNNFormer((32,) * 3, **kw), 1, 3, 16, 2)
    net = P.SwinUNETRCustom(P.SwinTransformerNNFormer((32,) * 3, **kw), 1, 3, (32,) * 3, 16, (2, 2, 2))
    a, b = net.state_dict(), ref.state_dict()
    assert sorted(a) == sorted(b) and all(a[k].shape == b[k].shape for k in a)


def _gloo_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from medicalsemseg_amd import parallel
    from medicalsemseg_amd.utils import misc
    parallel.init_from_env("gloo")
    flat = torch.full((1000,), float(rank + 1))
    parallel.all_reduce_flat_grads(flat)
    lo, hi = parallel.shard_windows(1000)
    m = misc.SmoothedValue()
    m.update(float(rank + 1), n=rank + 1)
    m.synchronize_between_processes()
    # GradSync with a two-phase backward: the suffix goes out in start(), the rest in finish(); stub optimiser / model
    # objects stand in for FlatAdamW / UNet (both GPU-only), the protocol is what is under test
    class _Opt:
        flat_grad = torch.arange(1000, dtype=torch.float32) * (rank + 1)
        _gscale = torch.ones(1)

        @staticmethod
        def early_suffix_offset(late):
            return 300

    class _Net:
        deferred = False

        def defer_backward_tail(self, on=True):
            self.deferred = on

        def tail_parameters(self):
            return []

        def backward_tail(self):
            pass

    net = _Net()
    gs = parallel.GradSync(_Opt, net)
    gs.start()
    after_start = (_Opt.flat_grad[299].item(), _Opt.flat_grad[300].item())   # head untouched so far
    gs.finish()
    gsync = (gs.overlapped, net.deferred, gs.split, after_start[0], float(_Opt.flat_grad[299]), float(_Opt.flat_grad[999]),
             float(_Opt._gscale))
    q.put((rank, float(flat[0]), parallel.all_reduce_mean(float(rank)), (lo, hi), m.count, m.total, gsync))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_gloo_world_size_2_flat_allreduce_and_window_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [3.0, 3.0]            # sum of 1 and 2
    assert [r[2] for r in res] == [0.5, 0.5]            # mean of ranks
    assert res[0][3] == (0, 500) and res[1][3] == (500, 1000)
    assert all(r[4] == 3 and r[5] == 5.0 for r in res)  # meter: counts 1+2, totals 1*1 + 2*2
    for r in res:
        ov, deferred, split, head_before, head_after, tail_after, gscale = r[6]
        assert ov and deferred and split == 300
        assert head_before == 299.0 * (r[0] + 1)         # start() reduces only flat_grad[300:]
        assert head_after == 299.0 * 3 and tail_after == 999.0 * 3 and gscale == 0.5


def test_window_sharding_is_a_partition():
    from medicalsemseg_amd.parallel import shard_windows
    for n, ws in [(1000, 8), (27, 4), (5, 8), (64, 3)]:
        spans = [shard_windows(n, ws, r) for r in range(ws)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_swin_unetr_official_load_from_maps_pretrained_keys():
    """SwinUNETR.load_from: the reference's pretrained-weight mapper (swin_unetr_official.py:232-280, :617-649), incl. the
    checkpoint's mlp.fc1 / fc2 -> linear1 / linear2 renaming"""
    import torch
    from medicalsemseg_amd.models.swin_unetr_official import SwinUNETR
    net = SwinUNETR((64, 64, 64), 1, 3, feature_size=12)
    g = torch.Generator().manual_seed(0)
    sd = {}
    for name, t in list(net.swinViT.named_parameters()) + list(net.swinViT.named_buffers()):
        src = "module." + name.replace("mlp.linear1", "mlp.fc1").replace("mlp.linear2", "mlp.fc2")
        sd[src] = (torch.randn(t.shape, generator=g) if t.is_floating_point() else t.clone())
    before_dec = net.decoder3.conv_block.conv1.conv.weight.clone()
    net.load_from({"state_dict": sd})
    assert torch.equal(net.swinViT.layers3[0].blocks[1].mlp.linear2.weight, sd["module.layers3.0.blocks.1.mlp.fc2.weight"])
    assert torch.equal(net.swinViT.layers1[0].downsample.reduction.weight, sd["module.layers1.0.downsample.reduction.weight"])
    assert torch.equal(net.swinViT.patch_embed.proj.bias, sd["module.patch_embed.proj.bias"])
    assert torch.equal(net.swinViT.layers4[0].blocks[0].attn.relative_position_bias_table,
                       sd["module.layers4.0.blocks.0.attn.relative_position_bias_table"])
    assert torch.equal(net.decoder3.conv_block.conv1.conv.weight, before_dec)      # the conv decoder is untouched
    cfg_ok = __import__("medicalsemseg_amd.models.model_builder", fromlist=["build_model"])
    from medicalsemseg_amd.utils.arguments import get_args
    cfg = get_args("--model SwinUNETR --vol_size 64 --hidden_dim 24 --output_dim 3".split())
    m = cfg_ok.build_model(cfg)
    assert type(m).__name__ == "SwinUNETR" and m.swinViT.layers1[0].blocks[0].attn.qkv.weight.shape == (72, 24)


def test_unetrc_state_dict_keys_equal_reference_layout():
    """product UNETRC keeps the key layout of /root/reference/models/segmentors/unetr.py (the oracle restatement is pinned
    against the reference class by tests/test_oracle_golden.py, weights filled by key name)"""
    from medicalsemseg_amd.models.unetrc import UNETRC
    from oracle.unetrc import UNETRC as OracleUNETRC
    from tests.golden_util import ToyTokenEncoder
    a = UNETRC(ToyTokenEncoder(1, 48), 1, 2)
    b = OracleUNETRC(ToyTokenEncoder(1, 48), 1, 2)
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb)
    assert all(sa[k].shape == sb[k].shape for k in sa)


def test_segformer3d_and_swindepth_state_dict_keys_and_builder():
    """product SegFormer3D / SwinDepth keep the reference's key layout (their oracles are pinned against the reference files,
    weights filled by key name) and build_model wires the branches of model_builder.py:120-171,190-205"""
    import argparse
    from medicalsemseg_amd.models.model_builder import build_model
    from oracle import segformer as OS, swin as OW
    cfg = argparse.Namespace(model="SegFormer3D", vol_size=64, patch_size=16, in_chans=1, hidden_dim=32, depths=[2, 1, 1, 1],
                             num_heads=[1, 2, 4, 8], qkv_bias=True, output_dim=3, compute_dtype="f32")
    net = build_model(cfg)
    ref = OS.SegFormerHeadOfficial(OS.MixVisionTransformer(1, 32, (1, 2, 4, 8), (4, 4, 4, 4), True, (2, 1, 1, 1), (8, 4, 2, 1)),
                                   [32, 64, 128, 256], 3, 0.1, 512)
    assert sorted(net.state_dict()) == sorted(ref.state_dict())
    assert all(net.state_dict()[k].shape == v.shape for k, v in ref.state_dict().items())
    cfg = argparse.Namespace(model="SwinDepth", vol_size=(24, 24, 24), patch_size=(2, 2, 2), in_chans=1, hidden_dim=32,
                             depths=[2, 2], num_heads=[2, 4], window_size=[6, 3], qkv_bias=True, mlp_ratio=4.0, output_dim=2,
                             compute_dtype="bf16")
    net = build_model(cfg)
    enc = OW.SwinTransformerNNFormer((24, 24, 24), (2, 2, 2), 1, 32, (2, 2), (2, 4), (6, 3), mlp="depth")
    assert sorted(net.encoder.state_dict()) == sorted(enc.state_dict())


def test_swinception_state_dict_has_reference_shapes_and_round_trips():
    """product SwInception computes on zero-padded channel counts (multiples of 8) but its state dict shows and accepts the
    reference's shapes (key layout of oracle/swin.py, which is pinned against the reference file); loading re-zeroes the
    padding; build_model wires the branch of model_builder.py:67-119"""
    import argparse
    from medicalsemseg_amd.models.model_builder import build_model
    from oracle import swin as OW
    from tests.golden_util import det_fill_
    cfg = argparse.Namespace(model="SwInception", vol_size=(24, 24, 24), patch_size=(2, 2, 2), in_chans=1, hidden_dim=32,
                             depths=[2, 2], num_heads=[2, 4], window_size=[6, 3], qkv_bias=True, mlp_ratio=4.0, output_dim=2,
                             compute_dtype="f32")
    net = build_model(cfg)
    ref = OW.SwinTransformerNNFormer((24, 24, 24), (2, 2, 2), 1, 32, (2, 2), (2, 4), (6, 3), mlp="inception")
    det_fill_(ref, "si")
    sd, rd = net.encoder.state_dict(), ref.state_dict()
    assert sorted(sd) == sorted(rd)
    assert all(sd[k].shape == v.shape for k, v in rd.items())
    mlp = net.encoder.layers[0].blocks[1].mlp
    conv = mlp.branches[2].branch3x3dbl_2.conv            # reference 4 -> 4 channels, computed as 8 -> 8
    assert tuple(conv.weight.shape) == (8, 8, 3, 3, 3) and tuple(mlp.fc.weight.shape) == (32, 5 * 32)
    with torch.no_grad():
        conv.weight.add_(1.0)                               # dirty the padding; a load must restore the zeros
    net.encoder.load_state_dict(rd)
    back = net.encoder.state_dict()
    assert all(torch.equal(back[k], v) for k, v in rd.items())
    assert float(conv.weight[4:].abs().max()) == 0.0 and float(conv.weight[:, 4:].abs().max()) == 0.0
    w = mlp.fc.weight.view(32, 5, 32)
    assert float(w[:, :, 25:].abs().max()) == 0.0 and torch.equal(w[:, :, :25].reshape(32, 125), rd["layers.0.blocks.1.mlp.fc.weight"])
    bn = mlp.branches[0].branch1x1.bn
    assert float(bn.weight[25:].abs().max()) == 0.0 and float(bn.bias[25:].abs().max()) == 0.0


def test_nifti_writer_round_trip_and_header(tmp_path):
    """NIfTI-1 single-file writer (what nib.save(nib.Nifti1Image(arr, affine)) produces for the reference's test outputs):
    header fields at their specified offsets, Fortran-ordered data, sform = affine, quaternion of a flipped-x affine"""
    import struct
    from medicalsemseg_amd.utils.nifti import load_nifti, save_nifti
    rng = np.random.default_rng(0)
    arr = rng.integers(0, 5, size=(5, 6, 7)).astype(np.uint8)
    aff = np.array([[-1.5, 0, 0, 10.0], [0, 1.5, 0, -20.0], [0, 0, 2.0, 5.0], [0, 0, 0, 1.0]])
    for name in ("seg.nii", "seg.nii.gz"):
        p = str(tmp_path / name)
        save_nifti(p, arr, aff)
        back, a2 = load_nifti(p)
        assert back.dtype == np.uint8 and np.array_equal(back, arr) and np.allclose(a2, aff)
    raw = open(tmp_path / "seg.nii", "rb").read()
    assert len(raw) == 352 + arr.size
    assert struct.unpack("<i", raw[0:4])[0] == 348 and raw[344:348] == b"n+1" + bytes(1)
    assert struct.unpack("<8h", raw[40:56]) == (3, 5, 6, 7, 1, 1, 1, 1)
    assert struct.unpack("<2h", raw[70:74]) == (2, 8)                               # DT_UINT8, 8 bits
    assert np.allclose(struct.unpack("<4f", raw[76:92]), (-1.0, 1.5, 1.5, 2.0))      # qfac, voxel sizes
    assert struct.unpack("<f", raw[108:112])[0] == 352.0
    assert struct.unpack("<2h", raw[252:256]) == (0, 2)                              # qform unknown, sform aligned
    assert np.allclose(struct.unpack("<3f", raw[256:268]), (0.0, 1.0, 0.0))          # 180 degrees about y after the z flip
    assert raw[352:352 + 5] == arr[:, 0, 0].tobytes()                                # first index fastest
    vol = rng.standard_normal((3, 4, 2)).astype(np.float32)
    save_nifti(str(tmp_path / "img.nii.gz"), vol, np.eye(4))
    back, _ = load_nifti(str(tmp_path / "img.nii.gz"))
    assert back.dtype == np.float32 and np.array_equal(back, vol)


def test_grad_buffer_epochs_and_sub_grid_word():
    """host logic of the lazy gradient buffers (layers._grad_buf: who overwrites, who accumulates) and of the packed sub-grid word
    of the PatchMerging gather (hip.merge_subs) -- no kernel involved"""
    import torch
    from medicalsemseg_amd import hip, layers

    class Owner:                       # what optim.FlatAdamW exposes to _grad_buf
        _gepoch = 0

    o = Owner()
    p = torch.nn.Parameter(torch.zeros(3))
    # no gradient yet: a fresh buffer, overwrite
    g, acc = layers._grad_buf(p)
    assert acc is False and g is p.grad
    # ordinary optimiser (no owner): always accumulate into the existing buffer
    assert layers._grad_buf(p)[1] is True and layers._grad_buf(p)[1] is True
    # owned, first kernel write ever: the slice was zero-filled this epoch, accumulate; the parameter becomes kernel-written
    p._msseg_gowner, p._msseg_gepoch = o, -1
    assert layers._grad_buf(p)[1] is True and p._msseg_kgrad is True and p._msseg_gepoch == 0
    assert layers._grad_buf(p)[1] is True                     # a second writer in the same epoch adds
    o._gepoch = 1                                             # zero_grad(): a new epoch, nothing filled
    assert layers._grad_buf(p)[1] is False                    # first writer overwrites
    assert layers._grad_buf(p)[1] is True                     # later writers (gradient accumulation, shared weights) add
    o._gepoch = 2
    assert layers._grad_buf(p)[1] is False
    # the reference's sub-grid order with its duplicates: (a, b, c) -> a | b << 1 | c << 2, slot s at bits 3s..3s+2
    sub = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 0), (0, 0, 1), (1, 1, 1)]
    word = hip.merge_subs(sub)
    assert [(word >> (3 * s)) & 7 for s in range(8)] == [0, 1, 2, 4, 5, 2, 4, 7]
