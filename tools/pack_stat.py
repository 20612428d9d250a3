"""bytes of packed weight images per (device, dtype) group after one training step of Swin-UNETR-48 (how much the batched repack
moves per step): python tools/pack_stat.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medicalsemseg_amd import layers
from medicalsemseg_amd.losses import DiceCELoss
from medicalsemseg_amd.models.swin_unetr import SwinTransformerNNFormer, SwinUNETRCustom
from medicalsemseg_amd.optim import FlatAdamW, add_weight_decay
dev = torch.device("cuda:0")
enc = SwinTransformerNNFormer((96,) * 3, (2, 2, 2), 1, 48, (2, 2, 2, 2), (3, 6, 12, 24), (6, 6, 6, 3), drop_path_rate=0.0,
                              compute_dtype=torch.bfloat16)
net = SwinUNETRCustom(enc, 1, 3, (96,) * 3, 48, (2, 2, 2), compute_dtype=torch.bfloat16).to(dev)
opt = FlatAdamW(add_weight_decay(net, 1e-5), lr=1e-4)
x = torch.randn(2, 1, 96, 96, 96, device=dev); y = torch.randint(0, 3, (2, 1, 96, 96, 96), device=dev).float()
DiceCELoss()(net((x, None, None)), y).backward(); opt.step(); opt.zero_grad()
nparam = sum(p.numel() for p in net.parameters())
for (d, dt), g in layers.PACK_REGISTRY.groups.items():
    tot = sum(int(r["job"].total) for r in g["jobs"])
    kinds = {}
    print(d, dt, "jobs", len(g["jobs"]), "image elements", tot, "= %.2f x the %d parameters" % (tot / nparam, nparam))
