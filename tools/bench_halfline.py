"""does reading the 32 used channels of 64-channel rows cost more HBM time than reading dense 32-channel rows?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medicalsemseg_amd import hip
dev = torch.device("cuda:0")
N, R, C = 2, 96, 32
dense = torch.randn(N, R, R, R, C, device=dev).bfloat16()
cat = torch.randn(N, R, R, R, 2 * C, device=dev).bfloat16()
flush = torch.empty(1 << 28, dtype=torch.uint8, device=dev)   # 256 MB: evict the Infinity Cache between runs
out = torch.empty(N, R // 2, R // 2, R // 2, C, device=dev, dtype=torch.bfloat16)
y = torch.empty_like(dense)


def timeit(name, fn, nbytes):
    ts = []
    for _ in range(6):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(ts)[len(ts) // 2]
    print(f"{name}: {t:.1f} us  ({nbytes / t / 1e6:.2f} TB/s on the bytes used)")


used = dense.numel() * 2
timeit("channel_stats dense rows", lambda: hip.channel_stats(dense), used)
timeit("channel_stats half of 64-ch rows", lambda: hip.channel_stats(cat[..., :C]), used)
timeit("maxpool dense rows", lambda: hip.maxpool2_fwd(dense, out), used * 1.125)
timeit("maxpool half of 64-ch rows", lambda: hip.maxpool2_fwd(cat[..., :C], out), used * 1.125)
st = hip.channel_stats(dense)
timeit("instnorm fwd dense -> dense", lambda: hip.instnorm_act_fwd(dense, st, None, None, y, 0.1), used * 2)
timeit("instnorm fwd dense -> half rows", lambda: hip.instnorm_act_fwd(dense, st, None, None, cat[..., :C], 0.1), used * 2)
