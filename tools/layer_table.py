"""Per-layer timing table of the conv k3 kernels over the BasicUNet shapes (B = 2 at 96^3 patches, or B = 8 windows).
Each entry is timed as a hipGraph of REP back-to-back launches so host launch cost is excluded.
usage: python tools/layer_table.py [batch] [top_size]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medicalsemseg_amd import hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else 96
REP = 10
dev = torch.device("cuda:0")
dt = torch.bfloat16
# (size divisor, cin, cout)
LAYERS = [(1, 32, 32), (1, 64, 32), (2, 32, 32), (2, 64, 32), (4, 32, 64), (4, 64, 64), (4, 128, 64),
          (8, 64, 128), (8, 128, 128), (8, 256, 128), (16, 128, 256), (16, 256, 256)]


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


print(f"B={B} top={TOP}  us per launch (TFLOP/s)")
tot = [0.0, 0.0, 0.0]
for div, cin, cout in LAYERS:
    s = TOP // div
    x = torch.randn(B, s, s, s, cin, device=dev).to(dt)
    dy = torch.randn(B, s, s, s, cout, device=dev).to(dt)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    y = torch.empty(B, s, s, s, cout, dtype=dt, device=dev)
    dx = torch.empty(B, s, s, s, cin, dtype=dt, device=dev)
    wp = hip.pack_conv_k3(w, dt, vol=(B, s, s, s))
    wpd = hip.pack_conv_k3(w, dt, dgrad=True, vol=(B, s, s, s))
    dw = torch.empty_like(w)
    stats = torch.empty(B, cout, 2, device=dev)
    fl = 2.0 * B * s ** 3 * 27 * cin * cout
    t_f = timed(lambda: hip.conv3d_k3(x, wp, None, y, cin, cout, stats))
    t_d = timed(lambda: hip.conv3d_k3(dy, wpd, None, dx, cout, cin))
    t_w = timed(lambda: hip.conv3d_k3_wgrad(x, dy, dw, cin, cout))
    tot[0] += t_f; tot[1] += t_d; tot[2] += t_w
    print(f"{s:3d}^3 {cin:3d}->{cout:3d}  {fl/1e9:7.2f} GF   fwd+stats {t_f:7.1f} ({fl/t_f/1e6:6.0f})   dgrad {t_d:7.1f} ({fl/t_d/1e6:6.0f})"
          f"   wgrad {t_w:7.1f} ({fl/t_w/1e6:6.0f})", flush=True)
print(f"sum fwd {tot[0]:.0f} us  dgrad {tot[1]:.0f} us  wgrad {tot[2]:.0f} us")
