"""Run under torch.distributed.run with 2 ranks on ONE GPU over gloo (tests/test_gpu_engine.py): BatchNorm with
cross-rank statistics (what torch.nn.SyncBatchNorm.convert_sync_batchnorm gives the reference under DDP,
/root/reference/run_training.py:83).  Each rank holds half of a batch; output, input gradient, affine gradients and
running statistics must equal a single-process BatchNorm over the whole batch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from medicalsemseg_amd import ops, parallel  # noqa: E402


def main():
    parallel.init_from_env()
    rk, ws = parallel.rank(), parallel.world_size()
    assert ws == 2
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    C = 64
    x = (torch.randn(4, 6, 8, 10, C, generator=g) * 1.7 + 0.4).to(dev)
    r = torch.randn(4, 6, 8, 10, C, generator=g).to(dev)
    mk = lambda: torch.nn.BatchNorm3d(C, eps=1e-3).to(dev)
    whole, mine = mk(), mk()
    with torch.no_grad():
        whole.weight.copy_(1 + 0.1 * torch.randn(C, generator=g).to(dev))
        whole.bias.copy_(0.1 * torch.randn(C, generator=g).to(dev))
    mine.load_state_dict(whole.state_dict())
    xa = x.clone().requires_grad_(True)
    ya = ops.batch_norm(xa, whole, None)                  # the whole batch, no synchronisation
    (ya * r).sum().backward()
    sl = slice(2 * rk, 2 * rk + 2)
    xb = x[sl].clone().requires_grad_(True)
    yb = ops.batch_norm(xb, mine, True)                   # this rank's half, statistics over both ranks
    (yb * r[sl]).sum().backward()
    assert torch.allclose(yb, ya[sl], rtol=1e-5, atol=1e-5), float((yb - ya[sl]).abs().max())
    assert torch.allclose(xb.grad, xa.grad[sl], rtol=1e-4, atol=1e-5), float((xb.grad - xa.grad[sl]).abs().max())
    gw, gb = mine.weight.grad.clone(), mine.bias.grad.clone()
    torch.distributed.all_reduce(gw); torch.distributed.all_reduce(gb)      # the gradient exchange of data parallelism
    assert torch.allclose(gw, whole.weight.grad, rtol=1e-4, atol=1e-3) and torch.allclose(gb, whole.bias.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(mine.running_mean, whole.running_mean, atol=1e-6)
    assert torch.allclose(mine.running_var, whole.running_var, atol=1e-6)
    model_level(rk, dev)
    torch.distributed.barrier()
    if rk == 0:
        print("SYNCBN_CHECK_OK")


def model_level(rk, dev):
    """the hook run_training.py calls under --distributed (`parallel.convert_sync_batchnorm`, the reference's
    run_training.py:83) on a whole model: SegFormer3D (BatchNorm in the fusion head) with one sample per rank must
    reproduce the single-process model on both samples -- logits, running statistics, one encoder gradient after the
    gradient exchange."""
    from medicalsemseg_amd.models.segformer3d import MixVisionTransformer, SegFormerHeadOfficial

    def mk():
        torch.manual_seed(3)
        enc = MixVisionTransformer(64, 16, 1, 32, (1, 2, 4, 8), (4, 4, 4, 4), True, 0.0, (1, 1, 1, 1), (8, 4, 2, 1),
                                   compute_dtype=torch.float32)
        return SegFormerHeadOfficial(enc, [32, 64, 128, 256], 3, 0.0, 64, compute_dtype=torch.float32).to(dev)

    whole, mine = mk(), mk()
    assert parallel.convert_sync_batchnorm(mine) >= 1 and parallel.convert_sync_batchnorm(whole, None) >= 1
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 1, 64, 64, 64, generator=g).to(dev)
    r = torch.randn(2, 3, 64, 64, 64, generator=g).to(dev)
    ya = whole((x, None, None))
    (ya.float() * r).sum().backward()
    yb = mine((x[rk:rk + 1], None, None))
    (yb.float() * r[rk:rk + 1]).sum().backward()
    err = float((yb.float() - ya[rk:rk + 1].float()).abs().max()) / float(ya.float().abs().max())
    assert err < 1e-4, err
    bw, bm = whole.linear_fuse.bn, mine.linear_fuse.bn
    assert torch.allclose(bm.running_mean, bw.running_mean, atol=1e-5) and torch.allclose(bm.running_var, bw.running_var, atol=1e-5)
    pw = dict(whole.named_parameters())
    for name, p in mine.named_parameters():
        if name.endswith("linear_pred.weight") or name.endswith("patch_embed1.proj.weight"):
            gm = p.grad.clone()
            torch.distributed.all_reduce(gm)
            ref = pw[name].grad
            e = float((gm - ref).norm() / ref.norm())
            assert e < 1e-3, (name, e)


if __name__ == "__main__":
    main()
