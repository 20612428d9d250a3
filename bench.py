#!/usr/bin/env python
"""Headline benchmark: 96^3 volumes/s of UNet (MONAI BasicUNet 1->3) forward + DiceCE + backward + AdamW,
bf16, per-GPU batch 2 (BASELINE.json configs[1]; configs[2] under torchrun = weak scaling).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see DESIGN.md for the fields).  `--workload sliding_window` times the
512^3 sliding-window inference loop instead (reported in DESIGN.md, not the headline line).
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

# algorithmic work of BasicUNet(32,32,64,128,256,32) 1->3, one 96^3 sample
UNET_FWD_GFLOP_PER_VOL = 252.4
UNET_FWDBWD_GFLOP_PER_VOL = 757.0
UNET_FWDBWD_GB_PER_VOL_BF16 = 2.06
MFMA_PEAK_BF16_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


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


def cpu_baseline(batch, size, n_cls, budget_s=25.0):
    """The CPU oracle (torch fp32, all host cores) on the same synthetic step; bounded sample."""
    from oracle.blocks import BasicUNet
    from oracle.losses import dice_ce_loss
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)   # the GPU box grants one GPU's CPU share (16 cores); oversubscribing slows torch down
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    net = BasicUNet(1, n_cls)
    opt = torch.optim.AdamW(net.parameters(), lr=4e-4, betas=(0.9, 0.95), eps=1e-6)
    x, y = synth_batch(batch, size, n_cls, "cpu", 13)
    times = []
    t_all = time.perf_counter()
    for i in range(4):
        t0 = time.perf_counter()
        loss = dice_ce_loss(net(x), y)
        loss.backward()
        opt.step()
        opt.zero_grad()
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget_s and i >= 1:
            break
    steady = times[1:] if len(times) > 1 else times
    med = sorted(steady)[len(steady) // 2]
    return {"value": round(batch / med, 4), "unit": "vol/s", "cores": cores, "kind": "port",
            "sample": f"{len(steady)} timed step(s) (after 1 warm-up) of the same B={batch} {size}^3 fwd+DiceCE+bwd+AdamW "
                      f"step, oracle/ BasicUNet fp32 on torch-CPU, median {med:.2f} s/step"}


def bench_sliding_window(args, dev, dtype, world, rank):
    """512^3 sliding-window inference (roi 96^3, overlap 0.5, gaussian; 1000 windows) with the UNet in eval mode"""
    from medicalsemseg_amd import parallel
    from medicalsemseg_amd.engine.utils import sliding_window_inference
    from medicalsemseg_amd.models.unet import UNet
    net = UNet(1, args.classes, compute_dtype=dtype).to(dev).eval()
    g = torch.Generator().manual_seed(13)
    vol = torch.randn(1, 1, args.sw_size, args.sw_size, args.sw_size, generator=g).to(dev)
    aff = torch.ones(1, 3, device=dev)

    def run():
        with torch.no_grad():
            return sliding_window_inference(vol, aff, (args.size,) * 3, args.sw_batch, net, overlap=0.5, mode="gaussian")
    for _ in range(max(args.warmup, 1)):
        out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": f"{args.sw_size}^3 sliding-window vols/sec", "value": round(args.steps / dt, 4),
                          "unit": "vol/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong",
                          "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                          "config": {"workload": f"UNet base 1->{args.classes}, {args.sw_size}^3 volume, roi {args.size}^3, "
                                                 f"overlap 0.5, gaussian, sw_batch {args.sw_batch}", "parallelism": f"windows/{world}",
                                     "out_mean": round(float(out.mean()), 5)}}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="do not replay the step from a captured hipGraph")
    ap.add_argument("--split-graph", action="store_true",
                    help="single GPU: use the multi-GPU replay structure (graph A | eager gap | graph B)")
    ap.add_argument("--workload", default="unet", choices=["unet", "swin_unetr", "sliding_window"],
                    help="unet = the headline (BASELINE configs[1]); swin_unetr = configs[3]; sliding_window = configs[4]")
    ap.add_argument("--sw-size", type=int, default=512)
    ap.add_argument("--sw-batch", type=int, default=8)   # windows per forward (1.48 / 1.64 / 1.51 vol/s at 4 / 8 / 16)
    args = ap.parse_args()

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
    else:
        net = UNet(1, args.classes, compute_dtype=dtype).to(dev)
    opt = FlatAdamW(add_weight_decay(net, 1e-5), lr=4e-4, betas=(0.9, 0.95), eps=1e-6)
    if world > 1:   # replicas start from rank 0's weights (as run_training.py does); data stays per-rank
        from medicalsemseg_amd import layers as _layers
        torch.distributed.broadcast(opt.flat_param, src=0)
        _layers.bump_weights_epoch()
    crit = DiceCELoss(smooth_nr=1e-5, smooth_dr=1e-5)
    x, y = synth_batch(args.batch, args.size, args.classes, dev, 13 + rank)

    # Data parallel: parallel.GradSync averages the flat gradient buffer over the ranks; with the UNet's two-phase
    # backward the first (large) all-reduce runs under the tail of the backward.  --split-graph rehearses the same
    # step structure on one GPU (no collective is issued on a single rank).
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
    # (forward + loss + backward + AdamW).  Multi-GPU (or --split-graph): graph A = forward + loss + backward, then
    # the RCCL all-reduce of the flat gradient buffer as an ordinary eager call, then graph B = AdamW + zero_grad --
    # the collective stays outside the graphs, the ~200 kernel launches of the step do not pay Python per launch.
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
        for _ in range(args.steps):
            loss = step()
    sync()
    dt = time.perf_counter() - t0
    hip.TIMER.enabled = False
    instr_steps = args.steps
    if graph is not None:
        instr_steps = min(args.steps, 5)
        # per-kernel HIP-event timing needs eager launches: instrument a few extra steps right after the timed region
        hip.TIMER.enabled = True
        for _ in range(min(args.steps, 5)):
            step()
        sync()
        hip.TIMER.enabled = False
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss_v = float(loss.detach())

    vols = args.batch * args.steps * world
    value = vols / dt
    res = {
        "metric": "96^3 vols/sec fwd+bwd (train)" if args.workload == "unet" else
                  "96^3 vols/sec fwd+bwd (train), Swin-UNETR-48 (reference encoder, window 6/6/6/3)",
        "value": round(value, 3), "unit": "vol/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "launch": ("hipGraph replay" if graph_b is None else
                   "hipGraph replay (fwd+bwd head | all-reduce under bwd tail | all-reduce | optimiser)" if graph_tail is not None
                   else "hipGraph replay (fwd+bwd | all-reduce | optimiser)")
                  if graph is not None else "eager",
        "config": {"workload": f"UNet base (MONAI BasicUNet 32-32-64-128-256-32) 1->{args.classes}cls, {args.size}^3 "
                               f"patches, DiceCE + AdamW, per-GPU batch {args.batch}", "global_batch": args.batch * world,
                   "parallelism": f"dp{world}", "final_loss": round(loss_v, 5)},
    }
    if rank == 0:
        summ = hip.TIMER.summary()
        # dominant kernel = the conv3d k3 forward / input-gradient kernel of the 32-channel stages (the 96^3 and 48^3
        # levels): v3 = LDS-DMA ping-pong kernel (conv3d_k3_pp.hip); falls back to the generic big-tile kernel (v0)
        kid = "conv3d_k3_fwd/v3" if "conv3d_k3_fwd/v3" in summ else "conv3d_k3_fwd/v0"
        k = summ.get(kid)
        if k:
            tf = k["flops"] / (k["total_ms"] * 1e-3) / 1e12
            peak = MFMA_PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
            allk = [v for kk, v in summ.items() if kk.startswith("conv3d_k3_fwd")]
            name = ("k3pp_kernel<STATS> (conv3d k3 fwd + dgrad, bf16, 32-channel stages, 4x4x16 tiles, LDS-DMA ping-pong)"
                    if kid.endswith("v3") else
                    "igemm_fwd_kernel<27,DIRECT,STORE,4,8,16,8,NT=2> (conv3d k3 fwd + dgrad, 4x8x16 tiles)")
            traffic = None
            try:   # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/README.md): FETCH_SIZE x2 + WRITE_SIZE
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_traffic.json")) as fh:
                    tj = json.load(fh)
                ent = tj.get(kid)
                if ent and args.dtype == "bf16" and args.size == ent.get("size") and args.batch == ent.get("batch"):
                    traffic = ent["hbm_bytes_per_launch"]
            except (OSError, ValueError):
                pass
            res["roofline"] = {"bound": "mfma", "kernel": name,
                               "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                               "traffic": traffic, "traffic_shape": "32->32 @96^3 B=2 launch (97.8 GFLOP, 226.5 MB algorithmic)" if traffic else None,
                               "launches": k["launches"], "avg_ms": round(k["avg_ms"], 4),
                               "flops_per_launch_avg": round(k["flops"] / k["launches"]),
                               "share_of_step": round((k["total_ms"] / instr_steps) / (dt * 1e3 / args.steps), 3),
                               "all_k3_variants_tflops": round(sum(v["flops"] for v in allk) / (sum(v["total_ms"] for v in allk) * 1e-3) / 1e12, 2)}
            w = summ.get("conv3d_k3_wgrad")
            if w:
                res["roofline"]["wgrad_tflops"] = round(w["flops"] / (w["total_ms"] * 1e-3) / 1e12, 2)
                res["roofline"]["wgrad_share_of_step"] = round((w["total_ms"] / instr_steps) / (dt * 1e3 / args.steps), 3)
        res["model_tflops"] = round(value / world * UNET_FWDBWD_GFLOP_PER_VOL / 1e3, 2)
        res["hbm_roofline_frac_algorithmic"] = round(value / world * UNET_FWDBWD_GB_PER_VOL_BF16 / HBM_PEAK_GBS, 4)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.batch, args.size, args.classes)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
