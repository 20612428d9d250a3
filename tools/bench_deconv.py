"""Per-shape timing of the transposed convolutions (k = s = 2) of BasicUNet and Swin-UNETR-48 at 96^3, batch 2: forward, input
gradient, weight gradient; us per call (hipGraph of REP calls) and the HBM-bound figure (fine tensor once at 8 TB/s).
usage: [MSSEG_NO_DECONV_GEN=1] [MSSEG_NO_DECONV_LWG=1] python tools/bench_deconv.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medicalsemseg_amd import hip

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


# (coarse edge, cin, cout)
SHAPES = [(48, 32, 32), (24, 64, 32), (12, 128, 64), (6, 256, 128),            # BasicUNet upcat_1 .. upcat_4
          (48, 48, 48), (24, 96, 48), (12, 192, 96), (6, 384, 192), (3, 768, 384)]   # Swin-UNETR decoder1 .. decoder5
tot = [0.0] * 3
print(" coarse  cin->cout |   fwd  dgrad  wgrad | HBM-bound (us)")
for e, cin, cout in SHAPES:
    N = 2
    x = torch.randn(N, e, e, e, cin, device=dev).to(dt)
    dy = torch.randn(N, 2 * e, 2 * e, 2 * e, cout, device=dev).to(dt)
    w = torch.randn(cin, cout, 2, 2, 2, device=dev) * 0.05
    wp = hip.pack_deconv(w, dt)
    wpd = hip.pack_deconv(w, dt, bwd=True)
    y = torch.empty_like(dy)
    dx = torch.empty_like(x)
    dw = torch.zeros_like(w)
    t = [timed(lambda: hip.deconv_k2s2(x, wp, None, y, cin, cout)), timed(lambda: hip.deconv_k2s2_bwd_data(dy, wpd, dx, cin, cout)),
         timed(lambda: hip.deconv_k2s2_wgrad(x, dy, dw, cin, cout))]
    for i in range(3):
        tot[i] += t[i]
    hb = dy.numel() * 2 / 8e6
    print(f"{e:4d}^3 {cin:4d}->{cout:4d} | {t[0]:5.1f}  {t[1]:5.1f}  {t[2]:5.1f} | {hb:5.1f}", flush=True)
print(f"sums: fwd {tot[0]:.1f}  dgrad {tot[1]:.1f}  wgrad {tot[2]:.1f}")
