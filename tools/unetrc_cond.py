"""conditioning of the UNETRC gradient probes: the oracle (stock torch ops) run on the GPU in fp32 vs the CPU golden"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle.unetrc import UNETRC
from tests.golden_util import UNETRC_PROBES, ToyTokenEncoder, det_fill_, det_tensor, probe
g = np.load('tests/golden/unetrc_ref.npz')
torch.backends.cudnn.allow_tf32 = False
net = UNETRC(ToyTokenEncoder(1, 48, (32, 32, 32), (16, 16, 16)), 1, 2)
det_fill_(net, "unetrc.")
net = net.cuda().train()
x = det_tensor("unetrc_x", (2, 1, 32, 32, 32)).cuda()
y = net(x)
print('logits', float((y.detach().cpu() - torch.from_numpy(g['logits'])).abs().max() / np.abs(g['logits']).max()))
(y * det_tensor("unetrc_r", tuple(y.shape)).cuda()).sum().backward()
P = dict(net.named_parameters())
for k in UNETRC_PROBES:
    w = torch.from_numpy(g['g:' + k])
    if float(w.norm()) < 1e-6: continue
    print(k, float((probe(P[k].grad).cpu() - w).norm() / w.norm()))
