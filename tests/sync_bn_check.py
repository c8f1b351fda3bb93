"""Two-rank check of the fused training BatchNorm in sync mode (run by tests/test_gpu_train.py under
torch.distributed.run, gloo backend, both ranks on cuda:0): norm.bn1d_relu on a NaiveSyncBatchNorm1d against the stock
composition norm._sync_bn + add + relu (the restatement of the reference's naiveSyncBN1d, ops/norm.py:186-211), ranks
holding DIFFERENT row counts: outputs, input / residual gradients, parameter gradients, running statistics."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from isfusion_amd import norm
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.manual_seed(100 + rank)
    worst = 0.0
    for c, n0, relu, with_res in ((64, 5000, True, False), (128, 777, True, True), (32, 20000, False, False)):
        n = n0 + 1234 * rank                                   # ranks disagree on the row count
        x0 = torch.randn(n, c, device=dev) * 2.0 + 0.5
        r0 = torch.randn(n, c, device=dev) if with_res else None
        g = torch.randn(n, c, device=dev)
        outs = []
        for fused in (True, False):
            torch.manual_seed(7)
            bn = norm.NaiveSyncBatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).train()
            with torch.no_grad():
                bn.weight.copy_(torch.rand(c) + 0.5)
                bn.bias.copy_(torch.randn(c) * 0.1)
            x = x0.clone().requires_grad_()
            r = r0.clone().requires_grad_() if with_res else None
            norm.FUSED_BN_TRAIN = fused
            y = norm.bn1d_relu(bn, x, residual=r, relu=relu)
            y.backward(g)
            outs.append([y.detach(), x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()] +
                        ([r.grad] if with_res else []))
        norm.FUSED_BN_TRAIN = True
        for a, b in zip(*outs):
            worst = max(worst, float((a - b).abs().max() / (b.abs().max() + 1e-6)))
    t = torch.tensor([worst], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(f"SYNC_BN_MAX_REL_ERR {float(t):.3e}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
