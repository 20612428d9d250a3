"""Per-shape timing of the deep-level 3x3x3 convolutions of Swin-UNETR-48 (3^3 ... 12^3 grids, 96 ... 768 channels, batch 2):
forward with statistics and input gradient, us per call, with the kernel the planner picks.  usage: python tools/bench_deep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medicalsemseg_amd import hip
from medicalsemseg_amd.layers import Conv3

REP = 10
dev = torch.device("cuda:0")
dt = torch.bfloat16


def timed(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * REP) * 1e3


print("grid  cin->cout | fwd+stats  dgrad  wgrad | weights MB (bf16) -> us at 4 TB/s")
for s, cin, cout in [(3, 768, 768), (6, 768, 384), (6, 384, 384), (12, 384, 192), (12, 192, 192), (24, 192, 96), (24, 96, 96), (12, 192, 192)]:
    x = torch.randn(2, s, s, s, cin, device=dev).to(dt)
    dy = torch.randn(2, s, s, s, cout, device=dev).to(dt)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02)
    op = Conv3(w, None)
    w.requires_grad_(False)
    tf = timed(lambda: op.fwd(x, want_stats=True))
    td = timed(lambda: op.bwd(x, dy, True))
    w.requires_grad_(True)
    w.grad = torch.zeros_like(w)
    tw = timed(lambda: hip.conv3d_k3_wgrad(x, dy, w.grad, cin, cout, False))
    mb = 27 * cin * cout * 2 / 1e6
    print(f"{s:3d}^3 {cin:4d}->{cout:4d} | {tf:7.1f} {td:7.1f} {tw:7.1f} | {mb:6.1f} MB -> {mb / 4.0:5.1f} us", flush=True)
