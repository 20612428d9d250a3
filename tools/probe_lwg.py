"""probe: kernels behind hip.linear_wgrad for one shape (run under rocprofv3 --kernel-trace --stats)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medicalsemseg_amd import hip
dev = torch.device("cuda:0"); dt = torch.bfloat16
tok, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(tok, cin, device=dev).to(dt); dy = torch.randn(tok, cout, device=dev).to(dt)
dw = torch.zeros(cout, cin, device=dev); db = torch.zeros(cout, device=dev)
for _ in range(20):
    hip.linear_wgrad(x, dy, dw, db, cin, cout)
torch.cuda.synchronize()
