"""micro-benchmark of the window-attention forward/backward (stage-0 shape of Swin-UNETR-48 by default)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medicalsemseg_amd import hip
B, R, C, heads, ws = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 48, 48, 3, 6
dev = torch.device("cuda:0")
qkv = torch.randn(B, R, R, R, 3 * C, device=dev).bfloat16()
qb = torch.randn(3 * C, device=dev)
tab = torch.randn((2 * ws - 1) ** 3, heads, device=dev) * 0.1
out = torch.empty(B, R, R, R, C, device=dev, dtype=torch.bfloat16)
dout = torch.randn_like(out)
dqkv = torch.empty_like(qkv)
dtab = torch.zeros_like(tab)
for shift in (0, 3):
    for name, fn in (("fwd", lambda: hip.window_attention_fwd(qkv, qb, tab, out, heads, ws, shift)),
                     ("bwd", lambda: hip.window_attention_bwd(qkv, qb, tab, out, lse, dout, dqkv, dtab if not os.environ.get("NO_DTAB") else None, heads, ws, shift))):
        lse = hip.window_attention_fwd(qkv, qb, tab, out, heads, ws, shift)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        nwin = B * (R // ws) ** 3
        flops = nwin * heads * 4.0 * 216 * 216 * 16 * (1 if name == "fwd" else 2.5)
        ms = e0.elapsed_time(e1) / 10
        print(f"window attention {name} shift={shift} ({nwin} windows x {heads} heads, N=216, d=16, "
              f"MFMA={'off' if os.environ.get('MSSEG_ATTN_NO_MFMA') else 'on'}): {ms*1e3:.1f} us  {flops/ms/1e9:.2f} TFLOP/s")
