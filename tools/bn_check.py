import torch, sys
sys.path.insert(0, '.')
from medicalsemseg_amd import layers, hip
torch.manual_seed(0)
dev = 'cuda:0'
N, C, S = 2, 64, (16, 16, 16)
x = torch.randn(N, C, *S, device=dev) * 1.5 + 0.3
bn = torch.nn.BatchNorm3d(C).to(dev)
with torch.no_grad():
    bn.weight.copy_(1 + 0.1 * torch.randn(C)); bn.bias.copy_(0.1 * torch.randn(C))
bn2 = torch.nn.BatchNorm3d(C).to(dev); bn2.load_state_dict(bn.state_dict())
xr = x.clone().requires_grad_(True)
y = torch.relu(bn(xr))
r = torch.randn_like(y)
(y * r).sum().backward()
op = layers.BatchNormAct(bn2, 0.0)
xcl = x.permute(0, 2, 3, 4, 1).contiguous()
a, s = op.fwd(xcl)
print('fwd', float((a.permute(0, 4, 1, 2, 3) - y).abs().max()))
dy = op.bwd(xcl, s, r.permute(0, 2, 3, 4, 1).contiguous())
e = (dy.permute(0, 4, 1, 2, 3) - xr.grad)
print('bwd rel', float(e.norm() / xr.grad.norm()), 'dgamma', float((bn2.weight.grad - bn.weight.grad).abs().max()), float(bn.weight.grad.abs().max()),
      'dbeta', float((bn2.bias.grad - bn.bias.grad).abs().max()))
print('rm', float((bn.running_mean - bn2.running_mean).abs().max()), 'rv', float((bn.running_var - bn2.running_var).abs().max()))
# same with InstanceNorm for reference
inn = layers.InstNormAct(bn2.weight, bn2.bias, 0.0)
bn2.weight.grad = None; bn2.bias.grad = None
a2, s2 = inn.fwd(xcl)
xr2 = x.clone().requires_grad_(True)
y2 = torch.relu(torch.nn.functional.instance_norm(xr2, weight=bn.weight, bias=bn.bias))
(y2 * r).sum().backward()
d2 = inn.bwd(xcl, s2, None, r.permute(0, 2, 3, 4, 1).contiguous())
print('IN fwd', float((a2.permute(0, 4, 1, 2, 3) - y2).abs().max()), 'bwd rel', float((d2.permute(0, 4, 1, 2, 3) - xr2.grad).norm() / xr2.grad.norm()))
