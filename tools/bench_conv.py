"""Micro-benchmark of single kernels (for rocprofv3 counter passes and quick A/B timing).
usage: python tools/bench_conv.py [fwd|wgrad|norm] [cin] [cout] [size] [iters]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medicalsemseg_amd import hip

what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 32
size = int(sys.argv[4]) if len(sys.argv) > 4 else 96
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda:0")
N = int(os.environ.get("MSSEG_BENCH_N", "2"))
dt = torch.bfloat16
x = torch.randn(N, size, size, size, cin, device=dev).to(dt)
dy = torch.randn(N, size, size, size, cout, device=dev).to(dt)
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
y = torch.empty(N, size, size, size, cout, dtype=dt, device=dev)
wp = hip.pack_conv_k3(w, dt, vol=(N, size, size, size))
dw = torch.empty_like(w)
stats = torch.empty(N, cout, 2, device=dev)
flops = 2.0 * N * size ** 3 * 27 * cin * cout


def run():
    if what == "fwd":
        hip.conv3d_k3(x, wp, None, y, cin, cout)
    elif what == "fwdstats":
        hip.conv3d_k3(x, wp, None, y, cin, cout, stats)
    elif what == "wgrad":
        hip.conv3d_k3_wgrad(x, dy, dw, cin, cout)
    elif what == "norm":
        hip.instnorm_act_fwd(dy, stats, None, None, y, 0.1)
    elif what == "normbwd":
        hip.instnorm_act_bwd(dy, stats, None, y, dy, y, 0.1)
    elif what == "stats":
        hip.channel_stats(dy, stats)


hip.channel_stats(dy, stats)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
nbytes = x.numel() * 2 + y.numel() * 2
print(f"{what} cin={cin} cout={cout} size={size}: {ms*1e3:.1f} us/launch  {flops/ms/1e9:.1f} TFLOP/s  ({nbytes/ms/1e6:.0f} GB/s in+out)")
if os.environ.get("MSSEG_DIAG") == "5":
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    lib = hip.load_library()
    lib.msseg_debug_phase_cycles.argtypes = [ctypes.c_void_p]
    if lib.msseg_debug_phase_cycles(buf) == 0:
        names = ["barrier0(wait readers)", "vmcnt(0) wait", "commit ds_write", "barrier1", "stores+fetch issue", "MFMA phase", "epilogue", "-"]
        tot = sum(buf[:7])
        for n_, v in zip(names, buf):
            print(f"  {n_:26s} {v:10d} ticks  {100.0 * v / max(tot, 1):5.1f} %")
        print(f"  total {tot} ticks over the launch of workgroup 0 / wave 0 ({ms*1e3:.1f} us)")
if os.environ.get("MSSEG_K3PP_TIMING"):
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    lib = hip.load_library()
    lib.msseg_debug_k3pp_cycles.argtypes = [ctypes.c_void_p]
    if lib.msseg_debug_k3pp_cycles(buf) == 0:
        names = ["MFMA role", "epilogue", "final vmcnt wait", "barrier wait", "halo load issue", "halo load wait"]
        print("  wave 0 ticks: " + "  ".join(f"{n_} {v}" for n_, v in zip(names, buf)) + f"  total {sum(buf[:6])}")
if os.environ.get("MSSEG_K3PP_TIMING") and what == "wgrad":
    buf4 = (ctypes.c_ulonglong * 4)()
    lib.msseg_debug_k3wg_cycles.argtypes = [ctypes.c_void_p]
    if lib.msseg_debug_k3wg_cycles(buf4) == 0:
        print("  wgrad wave 0 ticks: MFMA role %d  memory role %d  vmcnt wait %d  barrier wait %d" % tuple(buf4))
