#!/usr/bin/env python
"""Benchmarks of the hot path (BASELINE.json).  Default = the headline: 96^3 volumes/s of UNet (MONAI BasicUNet 1->3)
forward + DiceCE + backward + AdamW, bf16, per-GPU batch 2 (configs[1]; configs[2] under torchrun = weak scaling).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --workload swin_unetr        # configs[3]: Swin-UNETR-48 (reference encoder), same step
    python bench.py --workload sliding_window    # configs[4]: 512^3 sliding-window inference, roi 96^3, overlap 0.5

Prints ONE JSON line on rank 0; `config.workload`, the FLOP / byte model, `roofline` (dominant kernel, measured with HIP
events in this process) and `cpu_baseline` (the oracle on the host cores, bounded sample) all follow `--workload`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_BF16_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_PEAK_F32_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0

# Algorithmic work per unit (one 96^3 sample = one "vol" for training, one
# 512^3 volume for inference).  fwd+bwd = 3x forward FLOPs; fwd+bwd bytes = 3x forward bytes (X + 2 dY + dX + 2 W).
WORK = {
    "unet": {"gflop_per_vol": 757.0, "gb_per_vol_bf16": 2.06,
             "metric": "96^3 vols/sec fwd+bwd (train)",
             "name": "UNet base (MONAI BasicUNet 32-32-64-128-256-32) 1->{c}cls, {s}^3 patches, DiceCE + AdamW, per-GPU batch {b}"},
    "swin_unetr": {"gflop_per_vol": 3 * 631.0, "gb_per_vol_bf16": 3 * 1.88,
                   "metric": "96^3 vols/sec fwd+bwd (train), Swin-UNETR-48",
                   "name": "Swin-UNETR 48-feat (reference encoder swin_nnformer: depths 2-2-2-2, heads 3-6-12-24, windows 6-6-6-3, "
                           "patch 2 + UNETR decoder) 1->{c}cls, {s}^3 patches, DiceCE + AdamW, per-GPU batch {b}"},
    # MONAI / official variant (window 7 on a 49^3 padded token grid): the conv decoder dominates as above; the attention
    # and MLP FLOPs differ by < 2 % of the total, the SwinUNETRCustom figure is kept as the algorithmic model
    "swin_unetr_official": {"gflop_per_vol": 3 * 631.0, "gb_per_vol_bf16": 3 * 1.88,
                            "metric": "96^3 vols/sec fwd+bwd (train), Swin-UNETR-48 (MONAI / official variant)",
                            "name": "Swin-UNETR 48-feat (models/segmentors/swin_unetr_official.py: window 7, depths 2-2-2-2, heads "
                                    "3-6-12-24, Linear patch merging) 1->{c}cls, {s}^3 patches, DiceCE + AdamW, per-GPU batch {b}"},
    # no published FLOP model; matmul + conv FLOPs of one fwd+bwd step are counted on the CPU
    # oracle (torch.utils.flop_counter) in the cpu_baseline leg and reported as model_tflops when that leg runs
    "segformer3d": {"gflop_per_vol": None, "gb_per_vol_bf16": None,
                    "metric": "96^3 vols/sec fwd+bwd (train), SegFormer3D",
                    "name": "SegFormer3D (MixVisionTransformer 48-dim, depths 2-2-2-2, heads 1-2-4-8, sr 8-4-2-1 + SegFormerHeadOfficial "
                            "512) 1->{c}cls, {s}^3 patches, DiceCE + AdamW, per-GPU batch {b}"},
    "swin_depth": {"gflop_per_vol": None, "gb_per_vol_bf16": None,
                   "metric": "96^3 vols/sec fwd+bwd (train), SwinDepth-48",
                   "name": "SwinDepth 48-feat (reference encoder swindepth: depthwise-conv + BatchNorm MLP, depths 2-2-2-2, heads "
                           "3-6-12-24, windows 6-6-6-3, patch 2 + UNETR decoder) 1->{c}cls, {s}^3 patches, DiceCE + AdamW, per-GPU batch {b}"},
    "swinception": {"gflop_per_vol": None, "gb_per_vol_bf16": None,
                    "metric": "96^3 vols/sec fwd+bwd (train), SwInception-48",
                    "name": "SwInception 48-feat (reference encoder swinception: Inception-head MLP of Conv3d + BatchNorm3d + GELU "
                            "branches, depths 2-2-2-2, heads 3-6-12-24, windows 6-6-6-3, patch 2 + UNETR decoder) 1->{c}cls, {s}^3 "
                            "patches, DiceCE + AdamW, per-GPU batch {b}"},
    "sliding_window": {"gflop_per_vol": 252.4e3, "gb_per_vol_bf16": 714.0,
                       "metric": "512^3 sliding-window vols/sec",
                       "name": "UNet base 1->{c}cls, {v}^3 volume, roi {s}^3, overlap 0.5, gaussian, {w} windows, sw_batch {b}"},
}

# timer group (an entry point of the C ABI, split by the kernel variant its planner picks) -> (description, the main
# kernel's name in the library's own event pairs: msseg_ktimer_*)
KERNEL_NAMES = {
    "conv3d_k3_fwd/v3": ("k3pp_kernel (conv3d k3 fwd + dgrad, bf16, 32-input-channel stages, 4x4x16 tiles, LDS-DMA ping-pong)",
                         ("k3pp_kernel",)),
    "conv3d_k3_fwd/v4": ("k3c48_kernel (conv3d k3 fwd + dgrad, bf16, 48 input channels per launch, 16-wide cout blocks, ping-pong)",
                         ("k3c48_kernel",)),
    "conv3d_k3_fwd/v0": ("igemm_fwd_kernel<27,DIRECT,STORE,4,8,16,8> (conv3d k3 fwd + dgrad, 4x8x16 tiles)", ("igemm_fwd_kernel<27,4x8x16>",)),
    "conv3d_k3_fwd/v1": ("igemm_fwd_kernel<27,DIRECT,STORE,4,4,8,4> (conv3d k3 fwd + dgrad, 4x4x8 tiles)", ("igemm_fwd_kernel<27,4x4x8>",)),
    "conv3d_k3_fwd/v2": ("igemm_fwd_kernel<27,DIRECT,STORE,2,4,8,4> (conv3d k3 fwd + dgrad, 2x4x8 tiles)", ("igemm_fwd_kernel<27,2x4x8>",)),
    "conv3d_k3_wgrad/v3": ("k3wg_pp_kernel (conv3d k3 weight gradient, 32-channel block pairs, ping-pong)", ("k3wg_pp_kernel",)),
    "conv3d_k3_wgrad/v0": ("igemm_wgrad_kernel<27> (conv3d k3 weight gradient, generic)", ("igemm_wgrad_kernel<27>",)),
    "conv3d_k3_small": ("k3s_kernel (conv3d k3 fwd + dgrad on 12^3 / 6^3 grids, split-K over workgroups)", ("k3s_kernel",)),
}
MFMA_BOUND_FLOP_PER_BYTE = 60.0   # groups above this algorithmic intensity are priced against the MFMA peak, the rest against HBM


def host_cores():
    """(threads the CPU leg uses, cores this process may run on).  The GPU box grants one GPU's CPU share (16 cores); more
    torch threads than that oversubscribe the share and slow the oracle down, so the leg uses at most 16 and says so."""
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    return min(avail, 16), avail


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def synth_batch(batch, size, n_cls, device, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 1, size, size, size, generator=g)
    # three nested spheres -> non-degenerate Dice
    ax = torch.linspace(-1, 1, size)
    r = (ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2).sqrt()
    y = torch.zeros(size, size, size)
    for c in range(1, n_cls):
        y[r < 0.9 * (n_cls - c) / (n_cls - 1)] = c
    y = y[None, None].repeat(batch, 1, 1, 1, 1)
    return x.to(device), y.to(device)


def cpu_baseline_train(workload, batch, size, n_cls, budget_s=40.0):
    """The CPU oracle (torch fp32, host cores) on the same synthetic training step; bounded sample."""
    from oracle.losses import dice_ce_loss
    cores, avail = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    if workload == "unet":
        from oracle.blocks import BasicUNet
        net = BasicUNet(1, n_cls)
        what = "oracle/ BasicUNet"
    elif workload == "swin_unetr_official":
        from oracle import swin_official as O
        net = O.SwinUNETR((size,) * 3, 1, n_cls, feature_size=48)
        what = "oracle/ swin_official.SwinUNETR(48)"
    elif workload == "segformer3d":
        from oracle import segformer as O
        net = O.SegFormerHeadOfficial(O.MixVisionTransformer(1, 48, (1, 2, 4, 8), (4, 4, 4, 4), True, (2, 2, 2, 2), (8, 4, 2, 1)),
                                      [48, 96, 192, 384], n_cls, 0.1, 512)
        what = "oracle/ segformer.SegFormerHeadOfficial(MixVisionTransformer 48)"
    elif workload == "swin_depth":
        from oracle import swin as O
        net = O.SwinUNETRCustom(O.SwinTransformerNNFormer((size,) * 3, mlp="depth"), 1, n_cls, 48, 2)
        what = "oracle/ SwinUNETRCustom(SwinDepth 48)"
    elif workload == "swinception":
        from oracle import swin as O
        net = O.SwinUNETRCustom(O.SwinTransformerNNFormer((size,) * 3, mlp="inception"), 1, n_cls, 48, 2)
        what = "oracle/ SwinUNETRCustom(SwInception 48)"
    else:
        from oracle import swin as O
        net = O.SwinUNETRCustom(O.SwinTransformerNNFormer((size,) * 3), 1, n_cls, 48, 2)
        what = "oracle/ SwinUNETRCustom(SwinTransformerNNFormer 48)"
    opt = torch.optim.AdamW(net.parameters(), lr=4e-4, betas=(0.9, 0.95), eps=1e-6)
    x, y = synth_batch(batch, size, n_cls, "cpu", 13)
    times = []
    t_all = time.perf_counter()
    flops = None
    WARM, TIMED = 2, 5            # median of >= 5 steps after 2 warm-ups
    for i in range(WARM + TIMED):
        t0 = time.perf_counter()
        if i == 0 and WORK[workload]["gflop_per_vol"] is None:
            from torch.utils.flop_counter import FlopCounterMode
            with FlopCounterMode(display=False) as fc:
                loss = dice_ce_loss(net((x, None, None)), y)
                loss.backward()
            flops = fc.get_total_flops()
        else:
            loss = dice_ce_loss(net((x, None, None)), y)
            loss.backward()
        opt.step()
        opt.zero_grad()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget_s and i >= WARM:   # slow models: keep the leg bounded, say what was timed
            break
    nwarm = min(WARM, len(times) - 1)
    steady = times[nwarm:]
    med = sorted(steady)[len(steady) // 2]
    extra = {} if flops is None else {"counted_gflop_per_vol": round(flops / batch / 1e9, 1)}
    return {**extra, "value": round(batch / med, 4), "unit": "vol/s", "cores": cores, "cores_available": avail,
            "cpu_model": cpu_model(), "kind": "port",
            "sample": f"median of {len(steady)} timed step(s) after {nwarm} warm-up(s) of the same B={batch} {size}^3 "
                      f"fwd+DiceCE+bwd+AdamW step, {what} fp32 on torch-CPU with {cores} threads, {med:.2f} s/step"}


def cpu_baseline_sw(size, n_cls, n_windows, sw_batch=4, budget_s=25.0):
    """The oracle's window forward (+ its blend) on a bounded sample of the 512^3 job: >= 20 windows of 96^3, extrapolated
    to the job's window count."""
    from oracle.blocks import BasicUNet
    from oracle.sliding_window import compute_importance_map
    cores, avail = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = BasicUNet(1, n_cls).eval()
    imp = compute_importance_map((size,) * 3, "gaussian", 0.125)
    g = torch.Generator().manual_seed(13)
    acc = torch.zeros(n_cls, size, size, size)
    cnt = torch.zeros(n_cls, size, size, size)
    done, t_all = 0, time.perf_counter()
    with torch.no_grad():
        for _ in range(2):
            net(torch.randn(1, 1, size, size, size, generator=g))   # warm-ups
        t0 = time.perf_counter()
        while done < 20 or (time.perf_counter() - t_all < budget_s and done < 40):
            seg = net(torch.randn(sw_batch, 1, size, size, size, generator=g))
            for j in range(sw_batch):
                acc += imp * seg[j]
                cnt += imp
            done += sw_batch
        dt = time.perf_counter() - t0
    per_win = dt / done
    return {"value": round(1.0 / (per_win * n_windows), 6), "unit": "vol/s", "cores": cores, "cores_available": avail,
            "cpu_model": cpu_model(), "kind": "port",
            "sample": f"{done} windows of {size}^3 (forward of oracle/ BasicUNet fp32 on torch-CPU in batches of {sw_batch} + "
                      f"weighted blend), {per_win:.3f} s/window, extrapolated to the {n_windows} windows of one volume"}


def roofline_from_timer(summ, ksumm, dtype, step_ms, instr_steps, top=5):
    """The roofline object of the step's DOMINANT group -- the timer group with the largest measured total time (every
    entry point of the library is bracketed by HIP events on the launch stream while the timer is on) -- plus the top
    five groups.  `achieved` of the dominant group = its algorithmic flops (or bytes) / the duration of its MAIN kernel
    alone (the library's own event pairs around that launch, `ksumm`); where the library records no pair for the group the
    entry point's duration is used, which includes its small follow-up kernels."""
    if not summ:
        return None
    mfma_peak = MFMA_PEAK_BF16_TFLOPS if dtype == "bf16" else MFMA_PEAK_F32_TFLOPS
    esz_scale = 1.0
    total = sum(v["total_ms"] for v in summ.values())

    def price(kid, v):
        desc, knames = KERNEL_NAMES.get(kid, (kid, ()))
        ms = v["total_ms"]
        kern_ms = sum(ksumm[n]["total_ms"] for n in knames if n in ksumm) if ksumm else 0.0
        kern_n = sum(ksumm[n]["launches"] for n in knames if n in ksumm) if ksumm else 0
        own = kern_ms if (kern_ms > 0 and kern_n >= v["launches"]) else None   # the main kernel(s) alone, else the entry point
        t = (own if own is not None else ms) * 1e-3
        g = {"name": desc, "group": kid, "launches_per_step": round(v["launches"] / instr_steps, 1),
             "share_of_step": round((ms / instr_steps) / step_ms, 3), "avg_ms": round(ms / v["launches"], 4),
             "kernel_avg_ms": round(own / v["launches"], 4) if own is not None else None}
        fl, by = v["flops"], v["bytes"]
        if fl > 0 and (by <= 0 or fl / by >= MFMA_BOUND_FLOP_PER_BYTE):
            tf = fl / t / 1e12
            g.update(bound="mfma", achieved=round(tf, 2), unit="TFLOP/s", frac=round(tf / mfma_peak, 4))
        elif by > 0:
            gbs = by * esz_scale / t / 1e9
            g.update(bound="hbm", achieved=round(gbs, 1), unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4))
        else:
            g.update(bound=None, achieved=None, unit=None, frac=None)
        return g

    order = sorted(summ, key=lambda k: -summ[k]["total_ms"])
    groups = [price(k, summ[k]) for k in order[:top]]
    kid = order[0]
    k, top = summ[kid], groups[0]
    traffic = tshape = None
    try:   # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/README.md): FETCH_SIZE x2 + WRITE_SIZE
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            ent = json.load(fh).get(kid)
        if ent and dtype == "bf16":
            traffic, tshape = ent["hbm_bytes_per_launch"], ent.get("shape")
    except (OSError, ValueError):
        pass
    r = {"bound": top["bound"] or "hbm", "kernel": top["name"], "selected_by": "largest measured total time over the timer groups",
         "achieved": top["achieved"], "peak": mfma_peak if top["bound"] == "mfma" else HBM_PEAK_GBS, "unit": top["unit"],
         "frac": top["frac"], "traffic": traffic, "traffic_shape": tshape, "launches": k["launches"],
         "avg_ms": top["kernel_avg_ms"] if top["kernel_avg_ms"] is not None else top["avg_ms"],
         "avg_ms_measures": "the main kernel alone (library event pair around its launch)" if top["kernel_avg_ms"] is not None
                            else "the entry point (its follow-up kernels included)",
         "flops_per_launch_avg": round(k["flops"] / k["launches"]),
         "algorithmic_bytes_per_launch_avg": round(k["bytes"] / k["launches"]),
         "share_of_step": top["share_of_step"], "instrumented_share_of_step": round((total / instr_steps) / step_ms, 3),
         "groups": groups}
    return r


def measure_sliding_window(args, dev, dtype, world, rank, steps, warmup, with_roofline=True):
    """512^3 sliding-window inference (roi 96^3, overlap 0.5, gaussian; 1000 windows) with the UNet in eval mode: the second
    half of BASELINE.json's metric.  N > 1: window batches dealt round-robin to the ranks, one all-gather of logits per
    step, every rank blends.  Returns the result dict on rank 0 (None elsewhere)."""
    from medicalsemseg_amd import hip
    from medicalsemseg_amd.engine import utils as U
    from medicalsemseg_amd.models.unet import UNet
    net = UNet(1, args.classes, compute_dtype=dtype).to(dev).eval()
    g = torch.Generator().manual_seed(13)            # the same volume on every rank (shard_ranks contract)
    vol = torch.randn(1, 1, args.sw_size, args.sw_size, args.sw_size, generator=g).to(dev)
    aff = torch.ones(1, 3, device=dev)
    n_win = len(U.window_starts((args.sw_size,) * 3, (args.size,) * 3,
                                U.get_scan_interval((args.sw_size,) * 3, (args.size,) * 3, 3, 0.5)))

    def run():
        with torch.no_grad():
            return U.sliding_window_inference(vol, aff, (args.size,) * 3, args.sw_batch, net, overlap=0.5, mode="gaussian",
                                              shard_ranks=world > 1)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(warmup, 1)):
        out = run()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = run()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return None
    value = steps / dt
    w = WORK["sliding_window"]
    nsteps = -(-n_win // (args.sw_batch * world))
    res = {"metric": w["metric"] if args.sw_size == 512 else f"{args.sw_size}^3 sliding-window vols/sec",
           "value": round(value, 4), "unit": "vol/s", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 2), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": args.dtype, "data": "synthetic", "launch": "hipGraph replay of the window forward",
           "config": {"workload": w["name"].format(c=args.classes, v=args.sw_size, s=args.size, w=n_win, b=args.sw_batch),
                      "parallelism": f"window batches round-robin over {world} rank(s)" + (
                          ", one all-gather of compact logits per step, under the next step's forward" if world > 1 else ""),
                      "windows_per_rank": [sum(1 for s in range(nsteps) for j in range(args.sw_batch)
                                               if (s * world + r) * args.sw_batch + j < n_win) for r in range(world)],
                      "all_gather_bytes_per_rank_per_volume": (0 if world == 1 else
                          nsteps * world * args.sw_batch * args.size ** 3 * args.classes * (2 if args.dtype == "bf16" else 4)),
                      "out_mean": round(float(out.mean()), 5)}}
    scale = (args.sw_size / 512.0) ** 3
    res["model_tflops"] = round(value * w["gflop_per_vol"] * n_win / 1000.0 / 1e3, 2)
    res["hbm_roofline_frac_algorithmic"] = round(value * w["gb_per_vol_bf16"] * scale / HBM_PEAK_GBS, 4)
    if with_roofline:
        # dominant group: instrumented eager window batches right after the timed region
        win = torch.zeros(args.sw_batch, args.size, args.size, args.size, 1, dtype=dtype, device=dev)
        hip.TIMER.records.clear()
        hip.ktimer_summary()
        hip.TIMER.enabled = True
        hip.ktimer_enable(True)
        nrep = 3
        for _ in range(nrep):
            net.infer_cl(win)
        torch.cuda.synchronize()
        hip.TIMER.enabled = False
        hip.ktimer_enable(False)
        batch_ms = dt / steps * 1e3 / nsteps
        res["roofline"] = roofline_from_timer(hip.TIMER.summary(), hip.ktimer_summary(), args.dtype, batch_ms, nrep)
    return res, n_win


def bench_sliding_window(args, dev, dtype, world, rank):
    r = measure_sliding_window(args, dev, dtype, world, rank, args.steps, args.warmup)
    if rank == 0:
        res, n_win = r
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_sw(args.size, args.classes, n_win)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def dist_facts(dev, world, rank, nbytes):
    """what the collective library saw: makes the first multi-GPU run self-validating"""
    import torch.distributed as dist
    facts = {"world_size": world}
    if world == 1:
        return facts
    facts["backend"] = dist.get_backend()
    try:
        v = torch.cuda.nccl.version()
        facts["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:  # noqa: BLE001
        facts["rccl_version"] = f"unavailable ({type(e).__name__})"
    idx = torch.tensor([torch.cuda.current_device()], dtype=torch.int64, device=dev)
    allidx = [torch.zeros_like(idx) for _ in range(world)]
    dist.all_gather(allidx, idx)
    facts["device_index_per_rank"] = [int(t.item()) for t in allidx]
    name = torch.cuda.get_device_name(dev)
    facts["device_name_rank0"] = name
    # one all-reduce of a gradient-sized fp32 buffer, event-timed outside the graphs (5 after 2 warm-ups, max over ranks)
    buf = torch.zeros(max(nbytes // 4, 1), dtype=torch.float32, device=dev)
    for _ in range(2):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dist.barrier()
    e0.record()
    for _ in range(5):
        dist.all_reduce(buf)
    e1.record()
    torch.cuda.synchronize()
    t = torch.tensor([e0.elapsed_time(e1) / 5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    facts["grad_allreduce_ms"] = round(float(t.item()), 4)
    facts["grad_allreduce_bytes"] = int(buf.numel() * 4)
    facts["grad_allreduce_busbw_GBs"] = round(2 * (world - 1) / world * buf.numel() * 4 / (float(t.item()) * 1e-3) / 1e9, 1)
    return facts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="do not replay the step from a captured hipGraph")
    ap.add_argument("--split-graph", action="store_true",
                    help="single GPU: use the multi-GPU replay structure (graph A | eager gap | graph B)")
    ap.add_argument("--workload", default="unet", choices=["unet", "swin_unetr", "swin_unetr_official", "sliding_window", "segformer3d", "swin_depth",
                                                         "swinception"],
                    help="unet = the headline (BASELINE configs[1]); swin_unetr = configs[3]; sliding_window = configs[4]; "
                         "segformer3d / swin_depth / swinception = the SURVEY 8(f) N3 / N4 model families")
    ap.add_argument("--no-sliding-window", action="store_true",
                    help="default workload only: skip the 512^3 sliding-window leg (the second half of BASELINE.json's metric)")
    ap.add_argument("--sw-steps", type=int, default=3, help="timed 512^3 volumes of the sliding-window leg of the default line")
    ap.add_argument("--all-groups", action="store_true", help="list every timer group in roofline.groups, not the top five")
    ap.add_argument("--sw-size", type=int, default=512)
    ap.add_argument("--sw-batch", type=int, default=8)   # windows per forward
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 3 if args.workload == "sliding_window" else 20
    if args.warmup is None:
        args.warmup = 1 if args.workload == "sliding_window" else 5

    from medicalsemseg_amd import hip, parallel
    from medicalsemseg_amd.losses import DiceCELoss
    from medicalsemseg_amd.models.unet import UNet
    from medicalsemseg_amd.optim import FlatAdamW, add_weight_decay

    parallel.init_from_env()
    world, rank = parallel.world_size(), parallel.rank()
    if world != max(args.gpus, 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MSSEG_BENCH_ONE_DEVICE"):   # rehearsal of the N > 1 control flow on a one-GPU box (with gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    hip.load_library()

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(13 + rank)
    if args.workload == "sliding_window":
        return bench_sliding_window(args, dev, dtype, world, rank)
    if args.workload == "swin_unetr":
        from medicalsemseg_amd.models.swin_unetr import SwinTransformerNNFormer, SwinUNETRCustom
        enc = SwinTransformerNNFormer((args.size,) * 3, (2, 2, 2), 1, 48, (2, 2, 2, 2), (3, 6, 12, 24), (6, 6, 6, 3),
                                      drop_path_rate=0.0, compute_dtype=dtype)
        net = SwinUNETRCustom(enc, 1, args.classes, (args.size,) * 3, 48, (2, 2, 2), compute_dtype=dtype).to(dev)
        args.no_graph = args.no_graph or bool(os.environ.get("MSSEG_SWIN_NO_GRAPH"))
    elif args.workload in ("swin_depth", "swinception"):
        from medicalsemseg_amd.models.swin_unetr import SwInception, SwinDepth, SwinUNETRCustom
        enc_cls = SwinDepth if args.workload == "swin_depth" else SwInception
        enc = enc_cls((args.size,) * 3, (2, 2, 2), 1, 48, (2, 2, 2, 2), (3, 6, 12, 24), (6, 6, 6, 3), drop_path_rate=0.0,
                        compute_dtype=dtype)
        net = SwinUNETRCustom(enc, 1, args.classes, (args.size,) * 3, 48, (2, 2, 2), compute_dtype=dtype).to(dev)
        if world > 1:
            enc.sync_batchnorm(True)      # the reference converts every BatchNorm under DDP (run_training.py:83)
        args.no_graph = args.no_graph or world > 1    # SyncBatchNorm's collectives (and their host-side count) stay eager
    elif args.workload == "segformer3d":
        from medicalsemseg_amd.models.segformer3d import MixVisionTransformer, SegFormerHeadOfficial
        enc = MixVisionTransformer(args.size, 16, 1, 48, (1, 2, 4, 8), (4, 4, 4, 4), True, 0.0, (2, 2, 2, 2), (8, 4, 2, 1),
                                   compute_dtype=dtype)
        net = SegFormerHeadOfficial(enc, [48, 96, 192, 384], args.classes, 0.1, 512, compute_dtype=dtype).to(dev)
        if world > 1:
            net.sync_group = True
        args.no_graph = args.no_graph or world > 1
    elif args.workload == "swin_unetr_official":
        from medicalsemseg_amd.models.swin_unetr_official import SwinUNETR
        net = SwinUNETR((args.size,) * 3, 1, args.classes, feature_size=48, compute_dtype=dtype).to(dev)
        args.no_graph = args.no_graph or bool(os.environ.get("MSSEG_SWIN_NO_GRAPH"))
    else:
        net = UNet(1, args.classes, compute_dtype=dtype).to(dev)
    opt = FlatAdamW(add_weight_decay(net, 1e-5), lr=4e-4, betas=(0.9, 0.95), eps=1e-6)
    if world > 1:   # replicas start from rank 0's weights (as run_training.py does); data stays per-rank
        from medicalsemseg_amd import layers as _layers
        torch.distributed.broadcast(opt.flat_param, src=0)
        _layers.bump_weights_epoch()
    crit = DiceCELoss(smooth_nr=1e-5, smooth_dr=1e-5)
    x, y = synth_batch(args.batch, args.size, args.classes, dev, 13 + rank)

    # Data parallel: parallel.GradSync averages the flat gradient buffer over the ranks; with a two-phase backward the
    # first (large) all-reduce runs under the tail of the backward.  --split-graph rehearses the same step structure on
    # one GPU (no collective is issued on a single rank).
    gsync = parallel.GradSync(opt, net)
    if args.split_graph and hasattr(net, "defer_backward_tail") and not os.environ.get("MSSEG_NO_GRAD_OVERLAP"):
        net.defer_backward_tail(True)
    two_phase = bool(getattr(net, "_defer_tail", False))

    def step():
        out = net((x, None, None))
        loss = crit(out, y)
        loss.backward()
        if two_phase:
            gsync.start()
            net.backward_tail()
        gsync.finish()
        opt.step()
        opt.zero_grad()
        return loss

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        loss = step()
    sync()
    # Replay the step from captured hipGraphs; the work per replay is exactly the eager step's.  Single GPU: ONE graph
    # (forward + loss + backward + AdamW).  Multi-GPU (or --split-graph): graph A = forward + loss + backward head, the
    # RCCL all-reduce of the finished gradient suffix as an ordinary eager call under graph A2 = backward tail, the
    # all-reduce of the rest, then graph B = AdamW + zero_grad -- the collectives stay outside the graphs.
    graph = None
    graph_b = None
    graph_tail = None
    split = world > 1 or args.split_graph

    def part_a():
        out = net((x, None, None))
        loss = crit(out, y)
        loss.backward()
        return loss

    def part_b():
        opt.step()
        opt.zero_grad()

    def replay():
        graph.replay()
        if graph_b is not None:
            if graph_tail is not None:
                gsync.start()
                graph_tail.replay()
            gsync.finish()
            graph_b.replay()

    if not args.no_graph:
        try:
            from medicalsemseg_amd import layers
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step()
            torch.cuda.current_stream().wait_stream(side)
            layers.PACK_REGISTRY.prepare()
            layers.bump_weights_epoch()   # capture must include the weight re-packing kernels
            graph = torch.cuda.CUDAGraph()
            if not split:
                with torch.cuda.graph(graph):
                    static_loss = step()
            else:
                with torch.cuda.graph(graph):
                    static_loss = part_a()
                if two_phase:
                    gsync.start()
                    graph_tail = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph_tail, pool=graph.pool()):
                        net.backward_tail()
                gsync.finish()
                graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_b):
                    part_b()
                layers.bump_weights_epoch()
                opt._gscale.fill_(1.0)    # finish() above folded 1/world into the scale; the captured step did not run
            replay()
            sync()
        except Exception as e:  # noqa: BLE001
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            graph = graph_b = graph_tail = None
            torch.cuda.synchronize()
    t0 = time.perf_counter()
    if graph is not None:
        for _ in range(args.steps):
            replay()
        loss = static_loss
    else:
        hip.TIMER.enabled = True
        hip.ktimer_enable(True)
        for _ in range(args.steps):
            loss = step()
    sync()
    hip.ktimer_enable(False)
    dt = time.perf_counter() - t0
    hip.TIMER.enabled = False
    instr_steps = args.steps
    if graph is not None:
        instr_steps = min(args.steps, 5)
        # per-kernel HIP-event timing needs eager launches: instrument a few extra steps right after the timed region
        hip.TIMER.records.clear()
        hip.ktimer_summary()
        hip.TIMER.enabled = True
        hip.ktimer_enable(True)
        for _ in range(instr_steps):
            step()
        sync()
        hip.TIMER.enabled = False
        hip.ktimer_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss_v = float(loss.detach())
    facts = dist_facts(dev, world, rank, opt.flat_grad.numel() * 4)
    # The other half of BASELINE.json's metric ("... & 512^3 sliding-window vols/sec"): part of the default line so that
    # the driver's run times both halves.  All ranks take part (window batches are sharded over them).
    sw_res = None
    if args.workload == "unet" and not args.no_sliding_window:
        sw_res = measure_sliding_window(args, dev, dtype, world, rank, args.sw_steps, 1, with_roofline=False)

    vols = args.batch * args.steps * world
    value = vols / dt
    w = WORK[args.workload]
    res = {
        "metric": w["metric"], "value": round(value, 3), "unit": "vol/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "launch": ("hipGraph replay" if graph_b is None else
                   "hipGraph replay (fwd+bwd head | all-reduce under bwd tail | all-reduce | optimiser)" if graph_tail is not None
                   else "hipGraph replay (fwd+bwd | all-reduce | optimiser)")
                  if graph is not None else "eager",
        "config": {"workload": w["name"].format(c=args.classes, s=args.size, b=args.batch), "global_batch": args.batch * world,
                   "parallelism": f"dp{world}", "final_loss": round(loss_v, 5), "distributed": facts,
                   "grad_sync": ("none (single rank)" if world == 1 else
                                 f"flat fp32 gradient buffer, {opt.flat_grad.numel() * 4} bytes per step, " +
                                 ("two all-reduces, the first under the backward tail" if gsync.overlapped else "one all-reduce"))},
    }
    if rank == 0:
        scale = (args.size / 96.0) ** 3
        step_ms = dt * 1e3 / args.steps
        res["roofline"] = roofline_from_timer(hip.TIMER.summary(), hip.ktimer_summary(), args.dtype, step_ms, instr_steps,
                                              top=1000 if args.all_groups else 5)
        if w["gflop_per_vol"] is not None:
            res["model_tflops"] = round(value / world * w["gflop_per_vol"] * scale / 1e3, 2)
            res["hbm_roofline_frac_algorithmic"] = round(value / world * w["gb_per_vol_bf16"] * scale *
                                                         (1 if args.dtype == "bf16" else 2) / HBM_PEAK_GBS, 4)
        if sw_res is not None:
            r, n_win = sw_res
            res["sliding_window"] = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "steps": r["steps"],
                                     "warmup": r["warmup"], "ms_per_step": r["ms_per_step"], "scaling": r["scaling"],
                                     "model_tflops": r["model_tflops"], "launch": r["launch"],
                                     "hbm_roofline_frac_algorithmic": r["hbm_roofline_frac_algorithmic"], "config": r["config"]}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_train(args.workload, args.batch, args.size, args.classes)
            if "counted_gflop_per_vol" in res["cpu_baseline"]:
                res["model_tflops"] = round(value * res["cpu_baseline"]["counted_gflop_per_vol"] / 1e3, 2)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
