"""Per-shape timing of the 48-input-channel ping-pong kernel (csrc/conv3d_k3_c48.hip) on the Swin-UNETR decoder shapes:
forward with statistics, plain input gradient, input gradient with the fused InstanceNorm-backward sums, the accumulate
launch; each entry is a hipGraph of REP launches (the statistics-finalize launch of the fused modes included).
usage: python tools/bench_c48.py [batch]        (MSSEG_NO_K3C48=1: the generic kernel on the same shapes)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medicalsemseg_amd import hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
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


def cycles(fn, what):
    """MSSEG_K3C48_TIMING build: per-phase role cycles of workgroup 0 / wave 0 (group 0: MFMA role in even phases)"""
    import ctypes
    fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    L = ctypes.CDLL(os.path.join(os.path.dirname(hip.__file__), "libmsseg_hip.so"))
    if L.msseg_debug_k3c48_cycles(buf) == 0:
        t = list(buf)
        n = max(t[5], 1)
        half = n / 2
        print(f"    {what}: {t[5]} tiles; per phase: MFMA role {t[0]/half:7.0f} cyc | memory role: DMA issue {t[1]/half:6.0f} + epilogue "
              f"{t[2]/half:6.0f} | vmcnt wait {t[3]/(n+1):6.0f}  barrier wait {t[4]/(n+1):6.0f}  (averages over both roles' phases)", flush=True)


print(f"B={B}: kernel variant, us per call, TFLOP/s", flush=True)
for s, cout in [(96, 48), (96, 96), (48, 48), (48, 96), (64, 16), (64, 48)]:
    vol = (B, s, s, s)
    x = torch.randn(*vol, 48, device=dev).to(dt)
    w = torch.randn(cout, 48, 3, 3, 3, device=dev) * 0.03
    wp = hip.pack_conv_k3(w, dt, vol=vol)
    y = torch.empty(*vol, cout, device=dev, dtype=dt)
    yraw = torch.randn(*vol, cout, device=dev).to(dt)
    act = torch.randn(*vol, cout, device=dev).to(dt)
    stats = torch.empty(B, cout, 2, device=dev)
    fst = hip.channel_stats(yraw)
    var = hip.lib().msseg_conv3d_k3_kernel(*vol, 48, cout, hip.BF16)
    fl = 2.0 * B * s ** 3 * 27 * 48 * cout
    t0 = timed(lambda: hip.conv3d_k3(x, wp, None, y, 48, cout))
    t1 = timed(lambda: hip.conv3d_k3(x, wp, None, y, 48, cout, stats))
    t2 = timed(lambda: hip.conv3d_k3_dgrad_inbwd(x, wp, y, 48, cout, yraw, act, fst, 0.01, 1e-5))
    t3 = timed(lambda: hip.conv3d_k3_accumulate(x, wp, y, 48, cout, stats)) if var == 4 else float("nan")
    if os.environ.get("MSSEG_K3C48_TIMING") and var == 4:
        cycles(lambda: hip.conv3d_k3(x, wp, None, y, 48, cout), "plain")
        cycles(lambda: hip.conv3d_k3_dgrad_inbwd(x, wp, y, 48, cout, yraw, act, fst, 0.01, 1e-5), "in-bwd sums")
    print(f"{s:3d}^3 48->{cout:3d} v{var} {fl/1e9:6.1f} GF | plain {t0:6.1f} ({fl/t0/1e6:6.0f})  +stats {t1:6.1f} ({fl/t1/1e6:6.0f})  "
          f"+in-bwd sums {t2:6.1f} ({fl/t2/1e6:6.0f})  accumulate+stats {t3:6.1f} ({fl/t3/1e6:6.0f})", flush=True)
