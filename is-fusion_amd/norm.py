"""naiveSyncBN1d / naiveSyncBN2d (drop-in for mmdet3d/ops/norm.py:136-263).

Same parameters / buffers / state-dict keys as ``nn.BatchNorm1d``.  In eval mode, or when
``torch.distributed`` is not initialised or world_size == 1, it is plain BatchNorm (norm.py:172-175).
In distributed training the per-channel mean and mean-of-squares are exchanged with ONE all_reduce of a
[2C] vector (the reference does all_gather + sum, norm.py:12-19,186); on ROCm backend "nccl" is RCCL.
"""
import torch
import torch.distributed as dist
from torch import nn
from torch.autograd import Function


class _AllReduceSum(Function):
    """Differentiable sum-all-reduce: backward all-reduces the gradient (norm.py:21-24)."""

    @staticmethod
    def forward(ctx, x):
        y = x.clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g


def _sync_bn(mod, x, reduce_dims):
    C = x.shape[1]
    mean = x.mean(dim=reduce_dims)
    meansqr = (x * x).mean(dim=reduce_dims)
    vec = _AllReduceSum.apply(torch.cat([mean, meansqr], dim=0)) * (1.0 / dist.get_world_size())
    mean, meansqr = vec[:C], vec[C:]
    var = meansqr - mean * mean
    with torch.no_grad():
        mod.running_mean += mod.momentum * (mean.detach() - mod.running_mean)
        mod.running_var += mod.momentum * (var.detach() - mod.running_var)
    shape = [1, C] + [1] * (x.dim() - 2)
    scale = mod.weight * torch.rsqrt(var + mod.eps)
    shift = mod.bias - mean * scale
    return x * scale.reshape(shape) + shift.reshape(shape)


def _needs_sync(mod):
    return mod.training and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class NaiveSyncBatchNorm1d(nn.BatchNorm1d):

    def forward(self, input):
        assert input.dtype == torch.float32, f"input should be in float32 type, got {input.dtype}"
        if not _needs_sync(self):
            return super().forward(input)
        assert input.shape[0] > 0, "SyncBN does not support empty inputs"
        if input.dim() == 3:  # [N,C,L]
            return _sync_bn(self, input, (0, 2))
        return _sync_bn(self, input, (0,))


class NaiveSyncBatchNorm2d(nn.BatchNorm2d):

    def forward(self, input):
        assert input.dtype == torch.float32, f"input should be in float32 type, got {input.dtype}"
        if not _needs_sync(self):
            return super().forward(input)
        assert input.shape[0] > 0, "SyncBN does not support empty inputs"
        return _sync_bn(self, input, (0, 2, 3))


def build_norm_layer(cfg, num_features):
    """Tiny stand-alone equivalent of mmcv.cnn.build_norm_layer for the norm types the config uses
    (BN1d / BN / BN2d / naiveSyncBN1d / naiveSyncBN2d); returns (name, module) like mmcv."""
    cfg = dict(cfg)
    t = cfg.pop("type")
    cfg.pop("requires_grad", None)
    table = {"BN1d": nn.BatchNorm1d, "BN": nn.BatchNorm2d, "BN2d": nn.BatchNorm2d,
             "naiveSyncBN1d": NaiveSyncBatchNorm1d, "naiveSyncBN2d": NaiveSyncBatchNorm2d}
    if t not in table:
        raise KeyError(f"norm type {t} not available in isfusion_amd")
    cfg.setdefault("eps", 1e-5)
    return "bn", table[t](num_features, **cfg)


def fold_bn(bn):
    """eval BatchNorm -> (scale, shift) fp32 with y = x*scale + shift;  scale = gamma * rsqrt(var + eps)."""
    with torch.no_grad():
        w = bn.weight if bn.weight is not None else torch.ones_like(bn.running_var)
        b = bn.bias if bn.bias is not None else torch.zeros_like(bn.running_var)
        scale = (w.float() * torch.rsqrt(bn.running_var.float() + bn.eps)).contiguous()
        shift = (b.float() - bn.running_mean.float() * scale).contiguous()
    return scale, shift
