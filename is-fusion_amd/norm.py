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


# ---------------------------------------------------------------------------------------------- fused training BN
FUSED_BN_TRAIN = True   # False: the stock composition (nn.BatchNorm1d -> ReLU -> add) everywhere


class _BN1dReLUFunction(Function):
    """relu?(BatchNorm1d_train(x) + residual?) on [N, C] fp32 rows through isf_bn1d_* (isf_bn_train.hip): two launches +
    an ordered second-level sum per direction instead of ~10; the statistics (and the backward sums) are all-reduced
    across ranks when the module is a naiveSyncBN in a multi-rank job -- one all_reduce of [2C] per direction, what the
    reference's naiveSyncBN1d does (ops/norm.py:186-190; equal weight per rank, biased running variance)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, mod, relu, sync):
        from . import _lib
        lib = _lib.load()
        n, c = x.shape
        x = x.contiguous()
        res = residual.contiguous() if residual is not None else None
        stats = torch.empty(2 * c, dtype=torch.float32, device=x.device)
        # single process: sums about the batch's first row (no cancellation when |mean| >> std; torch's native batch_norm is
        # two-pass).  Multi-rank naiveSyncBN: plain (sum x, sum x^2) -- the reference's own formulation (ops/norm.py:186-190)
        pivot = None if sync else x[0]
        _lib.check(lib.isf_bn1d_stats_pivot(_lib.ptr(x), n, c, _lib.ptr(pivot), _lib.ptr(stats), _lib.stream()),
                   "isf_bn1d_stats_pivot")
        count, bwd_count = float(n), float(n)
        if sync:
            # the reference averages the per-rank mean / mean-of-squares with EQUAL weights (ops/norm.py:186-190), whatever
            # the ranks' row counts: (sum / n_r) summed over ranks with count = world size is exactly that
            world = dist.get_world_size()
            stats.mul_(1.0 / n)
            dist.all_reduce(stats, op=dist.ReduceOp.SUM)
            count, bwd_count = float(world), float(world) * n
        y = torch.empty_like(x)
        saved = torch.empty(2 * c, dtype=torch.float32, device=x.device)
        rm, rv = (mod.running_mean, mod.running_var) if mod.track_running_stats else (None, None)
        mom = mod.momentum if mod.momentum is not None else 0.1
        # (the module's num_batches_tracked counter is bumped inside the same launch: 41 one-element add_ launches per
        # training step otherwise, on a step that is paced by the host)
        nbt = mod.num_batches_tracked if mod.track_running_stats else None
        counted = nbt is not None and nbt.is_cuda and nbt.dtype == torch.int64
        _lib.check(lib.isf_bn1d_apply_pivot_counted(
            _lib.ptr(x), n, c, _lib.ptr(stats), _lib.ptr(pivot), count, _lib.ptr(gamma), _lib.ptr(beta), float(mod.eps),
            float(mom), 0 if sync else 1, _lib.ptr(rm), _lib.ptr(rv), _lib.ptr(nbt) if counted else None, _lib.ptr(res),
            int(bool(relu)), _lib.ptr(y), _lib.ptr(saved), _lib.stream()), "isf_bn1d_apply_pivot_counted")
        if nbt is not None and not counted:
            nbt.add_(1)
        ctx.save_for_backward(x, y if relu else None, gamma, saved)
        ctx.geom = (n, c, bwd_count, bool(relu), residual is not None, sync)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        lib = _lib.load()
        x, y, gamma, saved = ctx.saved_tensors
        n, c, count, relu, has_res, sync = ctx.geom
        dy = dy.contiguous().float()
        sums = torch.empty(2 * c, dtype=torch.float32, device=x.device)
        _lib.check(lib.isf_bn1d_backward_sums(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(y), n, c, _lib.ptr(saved),
                                              _lib.ptr(sums), _lib.stream()), "isf_bn1d_backward_sums")
        local = sums
        if sync:                      # dgamma / dbeta stay the LOCAL sums (DDP averages parameter gradients itself);
            local = sums.clone()      # dx needs the global ones
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        dgamma = torch.empty(c, dtype=torch.float32, device=x.device) if gamma is not None else None
        dbeta = torch.empty(c, dtype=torch.float32, device=x.device) if gamma is not None else None
        _lib.check(lib.isf_bn1d_backward_apply(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(y), n, c, _lib.ptr(saved),
                                               _lib.ptr(gamma), _lib.ptr(sums), count, _lib.ptr(dx), _lib.ptr(dres),
                                               _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.stream()),
                   "isf_bn1d_backward_apply")
        if sync and gamma is not None:
            dbeta, dgamma = local[:c].contiguous(), local[c:].contiguous()
        return dx, dgamma, dbeta, dres, None, None, None


def bn1d_relu(mod, x, residual=None, relu=True):
    """relu?(mod(x) + residual?) for a BatchNorm1d-like module on [N, C] rows.  Training mode on CUDA fp32 rows with a
    channel count the kernels tile: the fused HIP path above; anything else (eval mode, odd channel counts, CPU tensors,
    FUSED_BN_TRAIN = False): the stock composition, op for op what the reference runs."""
    c = x.shape[1] if x.dim() == 2 else 0
    # the choice must be the same on every rank of a synchronised module: the fused and the stock path all-reduce different
    # quantities in backward ((sum g, sum g xhat) vs d / d(mean, meansqr)), both [2C], so a rank-local row count would
    # pair them up silently.  Synchronised: any row count >= 1 takes the fused kernels (count = all ranks' rows);
    # single-process BatchNorm1d keeps torch's n > 1 requirement
    # (a BatchNorm2d on token rows -- dense_train.py: channels-last BatchNorm2d is BatchNorm1d over the tokens -- qualifies)
    synced = isinstance(mod, (NaiveSyncBatchNorm1d, NaiveSyncBatchNorm2d)) and _needs_sync(mod)
    ok = (FUSED_BN_TRAIN and mod.training and isinstance(mod, (nn.BatchNorm1d, nn.BatchNorm2d)) and x.dim() == 2 and
          x.is_cuda and
          x.dtype == torch.float32 and (x.shape[0] > 1 or (synced and x.shape[0] == 1)) and c % 4 == 0 and
          4 <= c <= 1024 and 256 % (c // 4) == 0 and torch.is_grad_enabled() and mod.momentum is not None)
    if not ok:
        if isinstance(mod, nn.BatchNorm2d) and x.dim() == 2:      # token rows through a 2-d module: the same arithmetic
            out = _bn2d_on_rows(mod, x, synced)
        else:
            out = mod(x)
        if residual is not None:
            out = out + residual
        return torch.relu(out) if relu else out
    sync = synced
    with torch.autocast("cuda", enabled=False):
        return _BN1dReLUFunction.apply(x, mod.weight, mod.bias, residual.float() if residual is not None else None,
                                       mod, relu, sync)


def _bn2d_on_rows(mod, x, synced):
    """nn.BatchNorm2d / naiveSyncBN2d semantics on [tokens, C] rows (the stock composition: eval mode, FUSED_BN_TRAIN off,
    channel counts the fused kernels do not tile)"""
    if synced:
        return _sync_bn(mod, x, (0,))
    use_batch = mod.training or not mod.track_running_stats
    mom = mod.momentum
    if mod.training and mod.track_running_stats and mod.num_batches_tracked is not None:
        mod.num_batches_tracked.add_(1)
        if mom is None:
            mom = 1.0 / float(mod.num_batches_tracked)
    return torch.nn.functional.batch_norm(x, mod.running_mean if mod.track_running_stats else None,
                                          mod.running_var if mod.track_running_stats else None, mod.weight, mod.bias,
                                          use_batch, 0.0 if mom is None else mom, mod.eps)


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
