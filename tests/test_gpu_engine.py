"""GPU tests of the engine-level paths: sliding-window inference vs the CPU oracle, the training / validation
loops on the product model, and the run_training.py driver end to end (synthetic data)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _pair(out_ch=2):
    from medicalsemseg_amd.models.unet import UNET_FEATURES, UNet
    from oracle.blocks import BasicUNet
    torch.manual_seed(0)
    ref = BasicUNet(1, out_ch, UNET_FEATURES["UNetSmall"]).eval()
    net = UNet(1, out_ch, UNET_FEATURES["UNetSmall"], compute_dtype=torch.float32)
    net.load_state_dict(ref.state_dict())
    return ref, net.to(DEV).eval()


@pytest.mark.parametrize("vol,roi,overlap,sw_batch,mode", [((40, 48, 56), (32, 32, 32), 0.5, 1, "gaussian"),
                                                            ((40, 48, 56), (32, 32, 32), 0.25, 4, "constant"),
                                                            ((24, 40, 32), (32, 32, 32), 0.5, 2, "gaussian")])
def test_sliding_window_inference_matches_oracle(vol, roi, overlap, sw_batch, mode):
    from medicalsemseg_amd.engine.utils import sliding_window_inference as sw_hip
    from oracle.sliding_window import sliding_window_inference as sw_ref
    ref, net = _pair()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 1, *vol, generator=g)
    aff = torch.ones(1, 3)
    with torch.no_grad():
        want = sw_ref(x, aff, roi, sw_batch, ref, overlap=overlap, mode=mode, cval=-1.5)
        got = sw_hip(x.to(DEV), aff.to(DEV), roi, sw_batch, net, overlap=overlap, mode=mode, cval=-1.5)
    assert got.shape == want.shape == (1, 2, *vol)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-4, atol=2e-4)
    # Dice of the argmax maps (north_star: within 1e-3)
    a, b = got.argmax(1).cpu(), want.argmax(1)
    dice = 2.0 * float(((a == 1) & (b == 1)).sum()) / max(float((a == 1).sum() + (b == 1).sum()), 1.0)
    assert dice > 1 - 1e-3


def test_engine_loops_on_gpu():
    from medicalsemseg_amd.data import SyntheticLoader
    from medicalsemseg_amd.engine.train import train_one_epoch
    from medicalsemseg_amd.engine.val import run_validation
    from medicalsemseg_amd.losses import DiceCELoss
    from medicalsemseg_amd.models.model_builder import build_model
    from medicalsemseg_amd.optim import FlatAdamW, add_weight_decay
    from medicalsemseg_amd.utils.arguments import get_args
    cfg = get_args("--model UNetSmall --output_dim 2 --vol_size 32 --gradient_clipping 1.0 --batch_size_val 2".split())
    torch.manual_seed(0)
    model = build_model(cfg).to(DEV)
    opt = FlatAdamW(add_weight_decay(model, 1e-5), lr=2e-3, betas=(0.9, 0.95), eps=1e-6)
    crit = DiceCELoss()
    scaler = torch.amp.GradScaler("cuda", enabled=False)
    loader = SyntheticLoader(6, 2, 32, 1, 2, seed=1)
    s0 = train_one_epoch(model, loader, opt, crit, torch.device(DEV), 0, scaler, cfg)
    s1 = train_one_epoch(model, loader, opt, crit, torch.device(DEV), 1, scaler, cfg)
    assert s1["train/loss"] < s0["train/loss"]
    v = run_validation(model, SyntheticLoader(1, 1, 48, 1, 2, seed=3, with_crop_info=False), crit, torch.device(DEV), 1, cfg)
    assert set(v) >= {"val/loss", "val/mDice"} and np.isfinite(v["val/loss"])


def test_swinception_trains_through_the_engine_and_padding_stays_zero():
    """cfg.model == 'SwInception' through build_model / train_one_epoch (hipGraph replay): the loss falls, the zero-padded
    channel entries of the Inception head (models/swin_unetr.py `_PaddedState`) are still exactly zero after AdamW steps with
    weight decay and gradient clipping, and the state dict keeps the reference's shapes"""
    from medicalsemseg_amd.data import SyntheticLoader
    from medicalsemseg_amd.engine.train import train_one_epoch
    from medicalsemseg_amd.losses import DiceCELoss
    from medicalsemseg_amd.models.model_builder import build_model
    from medicalsemseg_amd.models.swin_unetr import _BasicConv3d, _InceptionFc
    from medicalsemseg_amd.optim import FlatAdamW, add_weight_decay
    from medicalsemseg_amd.utils.arguments import get_args
    cfg = get_args("--model SwInception --output_dim 2 --vol_size 48 --patch_size 2 --hidden_dim 32 --depths 2 2 --num_heads 2 4 "
                   "--window_size 6 3 --qkv_bias --gradient_clipping 1.0".split())
    torch.manual_seed(0)
    model = build_model(cfg).to(DEV)
    opt = FlatAdamW(add_weight_decay(model, 1e-2), lr=2e-3, betas=(0.9, 0.95), eps=1e-6)
    crit = DiceCELoss()
    scaler = torch.amp.GradScaler("cuda", enabled=False)
    loader = SyntheticLoader(5, 2, 48, 1, 2, seed=1)
    s0 = train_one_epoch(model, loader, opt, crit, torch.device(DEV), 0, scaler, cfg)
    s1 = train_one_epoch(model, loader, opt, crit, torch.device(DEV), 1, scaler, cfg)
    assert np.isfinite(s1["train/loss"]) and s1["train/loss"] < s0["train/loss"]
    checked = 0
    for m in model.modules():
        if isinstance(m, _BasicConv3d):
            own = dict(m.named_parameters()); own.update(dict(m.named_buffers()))
            for name, real in m._real.items():
                if name.startswith("bn.running"):
                    continue                     # running statistics of a padding channel are (0, decaying 1): sliced off on save
                t = own[name].detach().clone()
                t[tuple(slice(0, n) for n in real)] = 0
                assert float(t.abs().max()) == 0.0, name
                checked += 1
        elif isinstance(m, _InceptionFc):
            w = m.weight.detach().view(m.weight.shape[0], m.nb, -1)
            assert w.shape[2] == (m.branch + 7) // 8 * 8
            if w.shape[2] > m.branch:
                assert float(w[:, :, m.branch:].abs().max()) == 0.0
            checked += 1
    assert checked > 40
    sd = model.state_dict()
    assert tuple(sd["encoder.layers.0.blocks.0.mlp.branches.1.branch3x3_2.conv.weight"].shape) == (25, 4, 3, 3, 3)
    assert tuple(sd["encoder.layers.1.blocks.1.mlp.fc.weight"].shape) == (64, 5 * 51)


def test_train_epoch_graph_replay_equals_eager(monkeypatch):
    """the hipGraph replay of forward+loss+backward inside train_one_epoch is the eager step, launch for launch"""
    from medicalsemseg_amd.data import SyntheticLoader
    from medicalsemseg_amd.engine.train import train_one_epoch
    from medicalsemseg_amd.losses import DiceCELoss
    from medicalsemseg_amd.models.model_builder import build_model
    from medicalsemseg_amd.optim import FlatAdamW, add_weight_decay
    from medicalsemseg_amd.utils.arguments import get_args
    cfg = get_args("--model UNetSmall --output_dim 2 --vol_size 32 --gradient_clipping 1.0".split())

    def run(eager):
        if eager:
            monkeypatch.setenv("MSSEG_NO_TRAIN_GRAPH", "1")
        else:
            monkeypatch.delenv("MSSEG_NO_TRAIN_GRAPH", raising=False)
        torch.manual_seed(0)
        model = build_model(cfg).to(DEV)
        opt = FlatAdamW(add_weight_decay(model, 1e-5), lr=2e-3, betas=(0.9, 0.95), eps=1e-6)
        scaler = torch.amp.GradScaler("cuda", enabled=False)
        stats = [train_one_epoch(model, SyntheticLoader(4, 2, 32, 1, 2, seed=1), opt, DiceCELoss(), torch.device(DEV), e,
                                 scaler, cfg)["train/loss"] for e in range(2)]
        return stats, opt.flat_param.clone()

    (la, pa), (lb, pb) = run(True), run(False)
    # same kernels in the same order; the DiceCE partial sums use float atomics, so runs agree to rounding only
    assert la == pytest.approx(lb, rel=2e-4)
    # (Adam turns a rounding-level difference of a near-zero gradient into a full +-lr step: bound the mean tightly,
    # the maximum by lr * steps)
    assert float((pa - pb).abs().mean()) < 2e-4 and float((pa - pb).abs().max()) < 2e-3 * 8


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_phase_backward_equals_single_phase(dtype):
    """defer_backward_tail: head (decoder + bottom level) in autograd's backward, encoder levels 3..0 in backward_tail();
    the gradients are those of the one-phase backward bit for bit, and the flat-buffer suffix the head finishes is
    complete before the tail runs (that suffix is what parallel.GradSync all-reduces under the tail)."""
    from medicalsemseg_amd.models.unet import UNET_FEATURES, UNet
    from medicalsemseg_amd.optim import FlatAdamW, add_weight_decay
    torch.manual_seed(3)
    net = UNet(1, 3, UNET_FEATURES["UNetSmall"], compute_dtype=dtype).to(DEV)
    opt = FlatAdamW(add_weight_decay(net, 1e-5), lr=1e-3)
    x = torch.randn(2, 1, 32, 32, 32, device=DEV)
    dy = torch.randn(2, 3, 32, 32, 32, device=DEV)
    net((x, None, None)).backward(dy)
    ref = opt.flat_grad.clone()
    opt.zero_grad()

    net.defer_backward_tail(True)
    o = opt.early_suffix_offset(net.tail_parameters())
    assert 0 < o < opt.flat_grad.numel()
    # (FlatAdamW.zero_grad() is lazy: it opens a new epoch in which the first kernel to write a gradient overwrites it; the
    #  buffer keeps the previous step's values until then -- mark the tail's slices to see that the head leaves them alone)
    tail_ids = {id(p) for p in net.tail_parameters()}
    for p, gv in opt._views:
        if id(p) in tail_ids:
            gv.fill_(-7.0)
    net((x, None, None)).backward(dy)
    head = opt.flat_grad.clone()
    assert torch.equal(head[o:], ref[o:])                       # everything behind the split is final
    for p, gv in opt._views:                                    # nothing of the tail has been written yet
        if id(p) in tail_ids:
            assert float((gv + 7.0).abs().max()) == 0.0
    net.backward_tail()
    assert torch.equal(opt.flat_grad, ref)
    net.backward_tail()                                         # idempotent: nothing pending
    assert torch.equal(opt.flat_grad, ref)
    net.defer_backward_tail(False)
    opt.zero_grad()
    net((x, None, None)).backward(dy)
    assert torch.equal(opt.flat_grad, ref)


def test_lazy_zero_grad_equals_zero_filled_gradients(monkeypatch):
    """FlatAdamW.zero_grad() without the fill (the first kernel that writes a gradient in an epoch overwrites it) against the
    zero-filled buffer with accumulating kernels (MSSEG_EAGER_ZERO_GRAD=1): the same parameters bit for bit after three
    clipped steps -- with a torch-autograd-written parameter in the optimiser (its slice keeps the fill), with a parameter no
    kernel touches in the second step, and with two backward passes before one of the steps (gradient accumulation)"""
    from medicalsemseg_amd.losses import DiceCELoss
    from medicalsemseg_amd.models.unet import UNET_FEATURES, UNet
    from medicalsemseg_amd.optim import FlatAdamW, add_weight_decay

    def run(eager):
        if eager:
            monkeypatch.setenv("MSSEG_EAGER_ZERO_GRAD", "1")
        else:
            monkeypatch.delenv("MSSEG_EAGER_ZERO_GRAD", raising=False)
        torch.manual_seed(5)
        net = UNet(1, 2, UNET_FEATURES["UNetSmall"], compute_dtype=torch.bfloat16).to(DEV)
        extra = torch.nn.Parameter(torch.full((3,), 0.5, device=DEV))           # written by torch autograd only
        unused = torch.nn.Parameter(torch.full((5,), 0.25, device=DEV))         # written in the first step only
        groups = add_weight_decay(net, 1e-5)
        groups[0]["params"] += [extra, unused]
        opt = FlatAdamW(groups, lr=1e-3)
        crit = DiceCELoss()
        g = torch.Generator().manual_seed(9)
        for it in range(3):
            for rep in range(2 if it == 1 else 1):                                  # step 1 accumulates two backward passes
                x = torch.randn(2, 1, 32, 32, 32, generator=g).to(DEV)
                y = torch.randint(0, 2, (2, 1, 32, 32, 32), generator=g).float().to(DEV)
                loss = crit(net((x, None, None)), y) + (extra * extra).sum() * 0.1
                if it == 0:
                    loss = loss + (unused * unused).sum()
                loss.backward()
            opt.clip_grad_norm_(1.0)
            opt.step()
            opt.zero_grad()
        return opt.flat_param.clone()

    lazy, eager = run(False), run(True)
    assert torch.equal(lazy, eager)


def test_run_training_driver_synthetic(tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, "run_training.py"), "--synthetic", "--model", "UNetSmall", "--output_dim", "2",
           "--vol_size", "32", "--n_images_per_batch", "2", "--synthetic_steps", "3", "--epochs", "2", "--val_interval", "2",
           "--synthetic_val_size", "48", "--warmup_epochs", "1", "--output_dir", str(tmp_path), "--save_ckpt_freq", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(tmp_path / "checkpoint-1.pth") and os.path.exists(tmp_path / "log.txt")
    ck = torch.load(tmp_path / "checkpoint-1.pth", map_location="cpu", weights_only=True)
    assert "conv_0.conv_0.conv.weight" in ck["model"] and ck["epoch"] == 1


def test_run_evaluation_and_run_test_drivers(tmp_path):
    """train -> checkpoint -> run_evaluation.py (eval_model, metrics json) and run_test.py (test_model, saved label maps)"""
    base = ["--synthetic", "--model", "UNetSmall", "--output_dim", "2", "--vol_size", "32", "--synthetic_val_size", "48",
            "--output_dir", str(tmp_path)]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_training.py"), *base, "--n_images_per_batch", "2",
                        "--synthetic_steps", "3", "--epochs", "2", "--val_interval", "2", "--warmup_epochs", "1",
                        "--save_ckpt_freq", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ck = str(tmp_path / "checkpoint-1.pth")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_evaluation.py"), *base, "--resume", ck, "--synthetic_steps", "2",
                        "--batch_size_val", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    import json
    ev = json.load(open(tmp_path / "eval.json"))
    assert set(ev) >= {"eval/loss", "eval/mDice"} and np.isfinite(ev["eval/loss"])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "run_test.py"), *base, "--resume", ck, "--synthetic_steps", "2",
                        "--save_eval_output"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    pred = np.load(tmp_path / "test_output" / "Fold0" / "pred" / "synthetic_0_0.npy")
    assert pred.dtype == np.uint8 and pred.shape == (48, 48, 48)


def test_run_training_driver_device_data_path(tmp_path):
    """the driver fed by the device-side crop + augmentation loader (N1): fg/bg crops, flips, rot90, intensity jitter"""
    cmd = [sys.executable, os.path.join(ROOT, "run_training.py"), "--synthetic", "--model", "UNetSmall", "--output_dim", "3",
           "--vol_size", "32", "--n_images_per_batch", "2", "--synthetic_steps", "4", "--epochs", "2", "--val_interval", "2",
           "--synthetic_val_size", "48", "--warmup_epochs", "1", "--output_dir", str(tmp_path), "--save_ckpt_freq", "2",
           "--t_rand_crop_fgbg", "--t_flip_prob", "0.5", "--t_rot_prob", "0.5", "--t_intensity_shift_prob", "0.5",
           "--t_intensity_scale_prob", "0.5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert os.path.exists(tmp_path / "checkpoint-1.pth")


def test_run_training_two_ranks_share_one_gpu_over_gloo(tmp_path):
    """data-parallel control flow of the driver (weight broadcast, GradSync with the two-phase backward, meter and
    loss reductions, rank-0 checkpointing) with two gloo ranks on this one GPU; RCCL needs one GPU per rank"""
    env = dict(os.environ, MSSEG_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "run_training.py"), "--synthetic", "--backend", "gloo",
           "--model", "UNetSmall", "--output_dim", "2", "--vol_size", "32", "--n_images_per_batch", "2",
           "--synthetic_steps", "3", "--epochs", "2", "--val_interval", "2", "--synthetic_val_size", "48",
           "--warmup_epochs", "1", "--gradient_clipping", "1.0", "--output_dir", str(tmp_path), "--save_ckpt_freq", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    ck = torch.load(tmp_path / "checkpoint-1.pth", map_location="cpu", weights_only=True)
    assert ck["epoch"] == 1 and all(torch.isfinite(v).all() for v in ck["model"].values())


def test_sliding_window_vs_reference_loop_golden(golden_dir):
    """the product's loop (batched gather / blend kernels, generic-predictor path) against outputs of the REFERENCE's own
    sliding_window_inference (/root/reference/engine/utils.py:19-159, tests/golden/sliding_window_ref.npz): padding,
    non-cubic volumes, sw_batch_size 1 (centers quirk) and short last batches, overlap 0.25 / 0.5, both blend modes"""
    from medicalsemseg_amd.engine.utils import sliding_window_inference
    from tests.golden_util import SW_CASES, det_tensor, sw_predictor
    g = np.load(os.path.join(golden_dir, "sliding_window_ref.npz"))
    for tag, vol, roi, sb, ov, mode, cval in SW_CASES:
        x = det_tensor("sw_x_" + tag, vol).to(DEV)
        aff = det_tensor("sw_aff_" + tag, (vol[0], 3)).to(DEV)
        y = sliding_window_inference(x, aff, roi, sb, sw_predictor, overlap=ov, mode=mode, cval=cval)
        assert tuple(y.shape) == g["out_" + tag].shape, tag
        np.testing.assert_allclose(y.cpu().numpy(), g["out_" + tag], rtol=1e-5, atol=1e-5, err_msg=tag)


def test_eval_model_and_test_model_vs_oracle(tmp_path):
    """engine.test.eval_model (built-in sliding window and a MONAI-style `inferer(inputs=, network=)`) and test_model
    (device argmax -> uint8, nearest resample to the original grid, saved maps) against the CPU oracle pipeline"""
    import functools
    from medicalsemseg_amd.data import SyntheticLoader
    from medicalsemseg_amd.engine.test import eval_model, majority_vote, test_model
    from medicalsemseg_amd.engine.utils import sliding_window_inference as sw_hip
    from medicalsemseg_amd.losses import DiceCELoss
    from medicalsemseg_amd.utils.arguments import get_args
    from oracle.losses import class_means_and_mdice, dice_ce_loss, dice_metric
    from oracle.postproc import argmax_labels, hausdorff95, hausdorff_mean, resample_nearest
    from oracle.sliding_window import sliding_window_inference as sw_ref
    ref, net = _pair(out_ch=3)
    cfg = get_args(f"--model UNetSmall --output_dim 3 --vol_size 32 --batch_size_val 2 --val_infer_overlap 0.5 "
                   f"--save_eval_output --t_voxel_spacings --output_dir {tmp_path}".split())
    loader = lambda: SyntheticLoader(2, 1, (48, 32, 64), 1, 3, seed=5, with_crop_info=False)   # noqa: E731
    # oracle pipeline: loop + network + loss + hard Dice on the CPU
    losses, mdices, maps, hds = [], [], [], []
    onehot = lambda m: np.stack([m == c for c in range(3)], 0)[None]   # noqa: E731
    with torch.no_grad():
        for b in loader():
            aff = torch.ones(1, 3)
            out = sw_ref(b["image"], aff, (32, 32, 32), 2, ref, overlap=0.5, mode="gaussian")
            losses.append(float(dice_ce_loss(out, b["label"])))
            mdices.append(float(class_means_and_mdice(*dice_metric(out, b["label"]))[1]))
            maps.append(argmax_labels(out[0].numpy()))
            hds.append(hausdorff_mean(hausdorff95(onehot(maps[-1]), onehot(b["label"][0, 0].numpy())))[0])
    crit = DiceCELoss()
    r1 = eval_model(None, net, loader(), crit, torch.device(DEV), cfg)
    inferer = lambda inputs, network: sw_hip(inputs, None, (32, 32, 32), 2, network, overlap=0.5, mode="gaussian")   # noqa: E731
    r2 = eval_model(inferer, net, loader(), crit, torch.device(DEV), cfg)
    for r in (r1, r2):
        assert abs(r["eval/loss"] - np.mean(losses)) < 1e-4
        assert abs(r["eval/mDice"] - np.mean(mdices)) < 1e-3
        # the reference's eval dict carries the Hausdorff-95 meter too (engine/test.py:20,64); an arg-max tie flipped by the
        # ~1e-5 logit differences can move a surface voxel: compare at that level, the exact test is test_hausdorff95_*
        assert "eval/mHdorffDist" in r and abs(r["eval/mHdorffDist"] - np.mean(hds)) < 0.05 * max(np.mean(hds), 1.0), (r, hds)
    for i, m in enumerate(maps):     # saved label maps == oracle arg max (ties aside: fp32 logits differ by ~1e-5)
        got = np.load(os.path.join(tmp_path, f"pred_synthetic_{i}_0.npy"))
        assert got.dtype == np.uint8 and (got != m).mean() < 1e-4

    class WithSpacing:    # the loader plus the Spacingd record test_model reads the original size from
        def __iter__(self):
            for b in loader():
                b["image_transforms"] = [{"class": ["Spacingd"], "orig_size": [torch.tensor([60]), torch.tensor([41]), torch.tensor([70])]}]
                yield b
    assert test_model(net, WithSpacing(), torch.device(DEV), cfg) is None
    out_dir = os.path.join(tmp_path, "test_output", "Fold0")
    for i, m in enumerate(maps):
        pred = np.load(os.path.join(out_dir, "pred", f"synthetic_{i}_0.npy"))
        rs = np.load(os.path.join(out_dir, "rs", f"synthetic_{i}_0.npy"))
        assert (pred != m).mean() < 1e-4
        assert rs.shape == (60, 41, 70) and np.array_equal(rs, resample_nearest(pred, (60, 41, 70)))
    # file names with a NIfTI extension (the real datasets: "img0001.nii.gz"): NIfTI-1 outputs in pred/, img/ and rs/
    class AsNifti:
        def __iter__(self):
            for i, b in enumerate(WithSpacing()):
                b["image_meta_dict"]["filename_or_obj"] = [f"/data/img{i:04d}.nii.gz"]
                b["image_meta_dict"]["affine"] = torch.diag(torch.tensor([-1.5, 1.5, 2.0, 1.0]))[None] + torch.tensor(
                    [[[0, 0, 0, 7.0], [0, 0, 0, -3.0], [0, 0, 0, 11.0], [0, 0, 0, 0]]])
                yield b
    from medicalsemseg_amd.utils.nifti import load_nifti
    test_model(net, AsNifti(), torch.device(DEV), cfg)
    for i, m in enumerate(maps):
        pred, aff = load_nifti(os.path.join(out_dir, "pred", f"{i:04d}.nii.gz"))
        assert pred.dtype == np.uint8 and (pred != m).mean() < 1e-4
        assert np.allclose(aff, np.diag([-1.5, 1.5, 2.0, 1.0]))            # translation zeroed as the reference does
        img, _ = load_nifti(os.path.join(out_dir, "img", f"{i:04d}.nii.gz"))
        assert img.shape == (48, 32, 64) and img.dtype == np.float32
        rs, _ = load_nifti(os.path.join(out_dir, "rs", f"{i:04d}.nii.gz"))
        assert rs.shape == (60, 41, 70)
    voted = majority_vote([maps[0], maps[0], maps[1]], 3).cpu().numpy()
    from oracle.postproc import majority_vote as mv_ref
    assert np.array_equal(voted, mv_ref(np.stack([maps[0], maps[0], maps[1]]), 3))


def test_hausdorff95_bit_exact_vs_scipy_oracle():
    """metrics.hausdorff95 (device surfaces, exact integer distance passes, histogram order statistics) against
    oracle/postproc.py (scipy binary_erosion / distance_transform_edt / np.percentile, MONAI's published algorithm;
    reference call site /root/reference/engine/test.py:31,48-51): random blobs, a class missing from both maps (NaN), a class
    only the prediction has (inf), a one-voxel class, objects on the volume border, non-cubic volume, batch of 2; then
    MONAI's "mean" reduction."""
    from medicalsemseg_amd import metrics
    from oracle.postproc import hausdorff95, hausdorff_mean
    rng = np.random.default_rng(3)
    D, H, W, C = 37, 52, 44, 6

    def blobs(seed_shift):
        m = np.zeros((D, H, W), np.uint8)
        for c in (1, 2, 3):
            for _ in range(2):
                ctr = rng.integers(0, [D, H, W])
                rad = rng.integers(3, 11, 3)
                zz, yy, xx = np.ogrid[:D, :H, :W]
                m[((zz - ctr[0]) / rad[0]) ** 2 + ((yy - ctr[1]) / rad[1]) ** 2 + ((xx - ctr[2]) / rad[2]) ** 2 <= 1.0] = c
        return m

    gt = np.stack([blobs(0), blobs(1)])
    pred = gt.copy()
    for b in range(2):                       # perturb: shift one class, carve another, add noise voxels
        pred[b] = np.where(np.roll(gt[b], (2, -3, 1), (0, 1, 2)) == 1, 1, np.where(gt[b] == 1, 0, gt[b]))
        pred[b][rng.random((D, H, W)) < 0.002] = 2
    pred[0][5, 6, 7] = 4                     # class 4: one voxel, prediction only -> inf
    pred[1][pred[1] == 3] = 0                # class 3 missing from the prediction of sample 1 -> inf there
    gt[1][0:4, 0:5, 0:6] = 2                 # an object in the corner (volume border = surface)
    # class 5: in neither map -> NaN
    onehot = lambda m: np.stack([m == c for c in range(C)], 1)   # noqa: E731
    want = hausdorff95(onehot(pred), onehot(gt))
    got = metrics.hausdorff95(torch.from_numpy(pred).to(DEV), torch.from_numpy(gt)[:, None].float().to(DEV), C)
    print("hausdorff95 oracle:", want.tolist(), "device:", got.tolist())
    assert got.shape == want.shape == (2, C)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(want[:, 5]).all()
    # a class only one map has: all-inf distances, which np.percentile (numpy >= 1.22) turns into NaN (inf - inf in its lerp)
    assert np.isnan(want[0, 4]) and np.isnan(want[1, 3])
    fin = np.isfinite(want)
    assert fin.sum() == 7 and np.array_equal(got[fin], want[fin]), (got, want)        # bit-exact doubles
    assert metrics.hausdorff_mean(got) == hausdorff_mean(want)


def test_overlapped_gradient_exchange_equals_plain_exchange_two_ranks():
    """2 gloo ranks on this GPU: three data-parallel optimiser steps with the all-reduce of the finished gradient suffix
    running under the backward tail (eager, and as graph A1 | collective | graph A2 | collective | graph B) leave the
    same bits as one all-reduce after the whole backward (tools/dp_overlap_check.py); and 2 ranks x B=2 with GradSync
    reproduce 1 process x B=4 (fp32 compute mode, three AdamW steps) to reduction-order tolerance"""
    env = dict(os.environ, MSSEG_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(29650 + os.getpid() % 200),
                        os.path.join(ROOT, "tools", "dp_overlap_check.py")], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DP_OVERLAP_OK" in r.stdout and "DP_EQUIV" in r.stdout
    print([l for l in r.stdout.splitlines() if "DP_EQUIV" in l])


def test_resume_from_reference_layout_checkpoint(tmp_path):
    """a checkpoint written the way the reference writes it (/root/reference/utils/misc.py:268-283: torch.optim.AdamW
    state dict over timm's two parameter groups, `cfg` stored as an argparse.Namespace) resumes into FlatAdamW: weights,
    both moments in param-group order, step count and epoch"""
    import argparse
    from medicalsemseg_amd.models.unet import UNET_FEATURES, UNet
    from medicalsemseg_amd.optim import FlatAdamW, add_weight_decay
    from medicalsemseg_amd.utils import misc
    from oracle.blocks import BasicUNet
    torch.manual_seed(1)
    ref = BasicUNet(1, 2, UNET_FEATURES["UNetSmall"])
    topt = torch.optim.AdamW(add_weight_decay(ref, 1e-5), lr=1e-3, betas=(0.9, 0.95), eps=1e-6)
    for _ in range(2):
        ref(torch.randn(1, 1, 32, 32, 32)).square().mean().backward()
        topt.step()
        topt.zero_grad()
    path = str(tmp_path / "ref_ckpt.pth")
    torch.save({"model": ref.state_dict(), "optimizer": topt.state_dict(), "epoch": 4, "scaler": None, "scheduler": None,
                "cfg": argparse.Namespace(model="UNetSmall", lr=1e-3)}, path)
    net = UNet(1, 2, UNET_FEATURES["UNetSmall"], compute_dtype=torch.float32).to(DEV)
    opt = FlatAdamW(add_weight_decay(net, 1e-5), lr=1e-3, betas=(0.9, 0.95), eps=1e-6)
    cfg = argparse.Namespace(resume=path, start_epoch=0, eval=False)
    misc.load_model(cfg, net, opt)
    assert cfg.start_epoch == 5 and opt._step == 2
    ref_params = [p for g in topt.param_groups for p in g["params"]]
    off = 0
    for (p, _), rp in zip(opt._views, ref_params):
        k = p.numel()
        assert torch.equal(p.detach().cpu(), rp.detach())
        st = topt.state[rp]
        assert torch.equal(opt.exp_avg[off:off + k].cpu(), st["exp_avg"].reshape(-1))
        assert torch.equal(opt.exp_avg_sq[off:off + k].cpu(), st["exp_avg_sq"].reshape(-1))
        off += k
    # one more step on both sides stays in agreement (bias correction uses the restored step count)
    x = torch.randn(1, 1, 32, 32, 32)
    ref(x).square().mean().backward()
    topt.step()
    net((x.to(DEV), None, None)).float().square().mean().backward()
    opt.step()
    num = sum(float(((p.detach().cpu() - rp.detach()) ** 2).sum()) for (p, _), rp in zip(opt._views, ref_params))
    den = sum(float((rp.detach() ** 2).sum()) for rp in ref_params)
    assert (num / den) ** 0.5 < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_unetrc_vs_reference_golden(dtype):
    """UNETR conv decoder (row A13) against the reference's own class (tests/golden/unetrc_ref.npz, generated by
    oracle/gen_golden.py from /root/reference/models/segmentors/unetr.py): training-mode BatchNorm on the InstanceNorm
    kernels, concat buffers written in place, gradients into decoder AND encoder parameters, running statistics, eval"""
    from medicalsemseg_amd.models.unetrc import UNETRC
    from tests.golden_util import UNETRC_PROBES, ToyTokenEncoder, det_fill_, det_tensor, probe
    g = np.load(os.path.join(ROOT, "tests", "golden", "unetrc_ref.npz"))
    net = UNETRC(ToyTokenEncoder(1, 48, (32, 32, 32), (16, 16, 16)), 1, 2, compute_dtype=dtype)
    det_fill_(net, "unetrc.")
    net = net.to(DEV).train()
    x = det_tensor("unetrc_x", (2, 1, 32, 32, 32)).to(DEV)
    y = net(x)
    want = torch.from_numpy(g["logits"])
    scale = float(want.abs().max())
    err = float((y.float().cpu() - want).abs().max()) / scale
    (y.float() * det_tensor("unetrc_r", tuple(y.shape)).to(DEV)).sum().backward()
    params = dict(net.named_parameters())
    worst = 0.0
    for k in UNETRC_PROBES:
        w = torch.from_numpy(g["g:" + k])
        got = probe(params[k].grad).float().cpu()
        if float(w.abs().max()) < 1e-6:      # biases in front of a training-mode BatchNorm: exactly zero here
            assert float(got.abs().max()) < 1e-3, k
            continue
        e = float((got - w).norm() / w.norm())
        print(f"   {k}: rel-L2 {e:.3e} (|g| {float(w.norm()):.3e})")
        worst = max(worst, e)
    print(f"[{dtype}] UNETRC 32^3: logits err/scale {err:.3e}, worst grad-probe rel-L2 {worst:.3e}")
    bn = net.decoder9_upsampler[1].block[1]
    assert float(bn.num_batches_tracked) == float(g["nbt"])
    # Gradient tolerance: this network (ReLU after BatchNorm over as few as 16 values per channel at the 2^3 level) is
    # ill-conditioned in its gradients -- the oracle's stock torch ops run in fp32 on the GPU differ from the CPU golden by
    # 4e-4 ... 5.4e-3 on these probes (tools/unetrc_cond.py) while the logits agree to 4e-6; the HIP path shows the same
    # 1.5e-3 ... 4.6e-3.  The gate is therefore the logits (2e-4) and the head / BatchNorm kernels' own exact tests.
    if dtype == torch.float32:
        assert err < 2e-4 and worst < 1e-2
        assert np.allclose(bn.running_mean.cpu().numpy(), g["rm"], rtol=1e-3, atol=1e-4)
        assert np.allclose(bn.running_var.cpu().numpy(), g["rv"], rtol=1e-3, atol=1e-4)
    else:
        assert err < 5e-2 and worst < 0.5      # bf16 drift on the same ill-conditioned probes, reported above
    net.eval()
    with torch.no_grad():
        ye = net(x).float().cpu()
    we = torch.from_numpy(g["logits_eval"])
    ee = float((ye - we).abs().max()) / float(we.abs().max())
    assert ee < (2e-4 if dtype == torch.float32 else 6e-2), ee


def test_sync_batchnorm_two_ranks_share_one_gpu_over_gloo():
    """SyncBatchNorm (reference: run_training.py:83 converts every BatchNorm under DDP): two ranks with half a batch each
    reproduce the whole-batch BatchNorm -- output, input gradient, affine gradients after the gradient exchange, running
    statistics"""
    env = dict(os.environ, MSSEG_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(29700 + os.getpid() % 200),
                        os.path.join(ROOT, "tools", "syncbn_check.py")], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "SYNCBN_CHECK_OK" in r.stdout
