"""Per-shape timing of the small-grid conv path (csrc/conv3d_k3_small.hip) against the tile kernels it replaces:
one conv + InstanceNorm + LeakyReLU unit forward, and the input-gradient + receiving unit's backward; each entry is a
hipGraph of REP launches.  usage: python tools/bench_small.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medicalsemseg_amd import hip
from medicalsemseg_amd.layers import Conv3, ConvNormAct, InstNormAct

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
REP = 10
dev = torch.device("cuda:0")
dt = torch.bfloat16
LAYERS = [(12, 128, 128)] if os.environ.get("MSSEG_K3S_DBG") else [(12, 64, 128), (12, 128, 128), (12, 256, 128), (6, 128, 256), (6, 256, 256), (24, 32, 64), (24, 64, 64), (24, 128, 64)]


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


print(f"B={B}: us per call   [small path: partials + finish | tile kernels: conv + finalize + normalise]")
for s, cin, cout in LAYERS:
    x = torch.randn(B, s, s, s, cin, device=dev).to(dt)
    P = lambda *sh: torch.nn.Parameter(torch.randn(*sh, device=dev) * 0.05)   # noqa: E731
    u0 = ConvNormAct(Conv3(P(cin, cin, 3, 3, 3), P(cin)), InstNormAct(P(cin), P(cin), 0.1))     # the receiving unit
    u1 = ConvNormAct(Conv3(P(cout, cin, 3, 3, 3), P(cout)), InstNormAct(P(cout), P(cout), 0.1))
    u1.conv.w.requires_grad_(False); u1.conv.b.requires_grad_(False)                               # time the dgrad side only
    res = {}
    for mode in ("small", "tile"):
        if mode == "tile":
            os.environ["MSSEG_NO_K3_SMALL"] = "1"
        else:
            os.environ.pop("MSSEG_NO_K3_SMALL", None)
        if mode == "small" and not hip.conv3d_k3_small_ok(x, cin, cout):
            res[mode] = (float("nan"),) * 4
            continue
        a0, s0 = u0.fwd(x)
        a1, s1 = u1.fwd(a0)
        dy = torch.randn_like(s1[1])
        t_f = timed(lambda: u1.fwd(a0))
        t_b = timed(lambda: u1.conv.bwd(a0, dy, True, bias_grad_is_zero=True, next_norm=(u0.norm, s0[1], s0[2], s0[3])))
        t_bn = 0.0
        if mode == "tile":
            da, red = u1.conv.bwd(a0, dy, True, bias_grad_is_zero=True, next_norm=(u0.norm, s0[1], s0[2], s0[3]))
            t_bn = timed(lambda: u0.norm.bwd(s0[1], s0[2], s0[3], da, red=red))
        if mode == "small":
            wp = u1.conv.cache.get(u1.conv.w, dt, "fs", lambda: hip.pack_conv_k3(u1.conv.w.detach(), dt, cb=32))
            t_p = timed(lambda: hip.conv3d_k3_small_partials(a0, wp, cin, cout))
        else:
            t_p = 0.0
        res[mode] = (t_f, t_b, t_p, t_bn)
    os.environ.pop("MSSEG_NO_K3_SMALL", None)
    fl = 2.0 * B * s ** 3 * 27 * cin * cout
    sm, tl = res["small"], res["tile"]
    print(f"{s:3d}^3 {cin:3d}->{cout:3d} {fl/1e9:6.2f} GF | small: fwd unit {sm[0]:6.1f} (partials alone {sm[2]:5.1f})  dgrad+unit bwd {sm[1]:6.1f} | "
          f"tile: fwd unit {tl[0]:6.1f}  dgrad+sums {tl[1]:6.1f} + apply ~{tl[3]:5.1f}", flush=True)
