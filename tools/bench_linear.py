"""Per-shape timing of the Linear layers of the Swin stages (embedding 48, 96^3 input, batch 2): forward, input gradient,
weight gradient and bias gradient; us per call (hipGraph of REP calls) and the HBM-bound figure (algorithmic bytes at 8 TB/s).
usage: python tools/bench_linear.py"""
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


SHAPES = []
for tok, C in ((221184, 48), (27648, 96), (3456, 192), (432, 384)):
    SHAPES += [(tok, C, 3 * C), (tok, C, C), (tok, C, 4 * C), (tok, 4 * C, C)]
SHAPES += [(27648, 384, 96), (3456, 768, 192), (432, 1536, 384)]
tot = [0.0] * 4
print("tokens  cin->cout |   fwd  dgrad  wgrad  dbias | HBM-bound fwd / wgrad (us)")
for tok, cin, cout in SHAPES:
    x = torch.randn(tok, cin, device=dev).to(dt)
    dy = torch.randn(tok, cout, device=dev).to(dt)
    w = torch.randn(cout, cin, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    wp = hip.pack_conv_k1(w, dt)
    wpd = hip.pack_conv_k1(w, dt, dgrad=True)
    y = torch.empty(tok, cout, device=dev, dtype=dt)
    dx = torch.empty(tok, cin, device=dev, dtype=dt)
    dw = torch.zeros(cout, cin, device=dev)
    db = torch.zeros(cout, device=dev)
    t = [timed(lambda: hip.conv3d_k1(x, wp, b, y, cin, cout)), timed(lambda: hip.conv3d_k1(dy, wpd, None, dx, cout, cin)),
         timed(lambda: hip.conv3d_k1_wgrad(x, dy, dw, cin, cout)), timed(lambda: hip.channel_sum(dy, db))]
    tf = timed(lambda: hip.linear_wgrad(x, dy, dw, db, cin, cout)) if hip.linear_wgrad_ok(x, cin, cout) else float("nan")
    totf = globals().get("totf", 0.0) + tf
    globals()["totf"] = totf
    for i in range(4):
        tot[i] += t[i]
    hb = tok * (cin + cout) * 2 / 8e6
    print(f"{tok:6d} {cin:4d}->{cout:4d} | {t[0]:5.1f}  {t[1]:5.1f}  {t[2]:5.1f}  {t[3]:5.1f} | {hb:5.1f} | one-pass wgrad+dbias {tf:5.1f}", flush=True)
print(f"one-pass wgrad+dbias sum {totf:6.1f}")
print(f"sum (x2 blocks per stage for the first 16 rows not applied) | {tot[0]:6.1f} {tot[1]:6.1f} {tot[2]:6.1f} {tot[3]:6.1f}")
