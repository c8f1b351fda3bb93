"""Voxel feature encoders (drop-in for mmdet3d/models/voxel_encoders: DynamicVFE, HardSimpleVFE).

State-dict keys equal the reference's (``vfe_layers.{i}.linear.weight``, ``vfe_layers.{i}.norm.*``), so a
released IS-Fusion checkpoint loads unchanged.  Eval-mode forward = ONE fused C call
(isf_dynamic_vfe_forward); training-mode forward (batch-statistics BN) composes the HIP DynamicScatter op
with stock torch Linear/BatchNorm so that autograd works.
"""
import ctypes

import torch
from torch import nn
from torch.nn import functional as F

from . import _lib
from .norm import bn1d_relu, build_norm_layer, fold_bn
from .scatter_points import dynamic_point_to_voxel_forward, dynamic_scatter


class HardSimpleVFE(nn.Module):
    """mean of the points of each voxel (voxel_encoder.py:14-45)."""

    def __init__(self, num_features=4):
        super().__init__()
        self.num_features = num_features
        self.fp16_enabled = False

    def forward(self, features, num_points, coors=None):
        _lib.require_cuda(features, num_points)
        features = features.contiguous().float()
        M, T, C = features.shape
        out = torch.empty((M, self.num_features), dtype=torch.float32, device=features.device)
        lib = _lib.load()
        npts = num_points.contiguous().int()                                     # local: must outlive the C call
        _lib.check(lib.isf_hard_simple_vfe(_lib.ptr(features), _lib.ptr(npts), M, T,
                                           C, self.num_features, _lib.ptr(out), _lib.stream()),
                   "isf_hard_simple_vfe")
        return out.contiguous()


class _RowsLinear(torch.autograd.Function):
    """y = x W^T on [P, K] point rows (P in the hundreds of thousands, K, N <= 128).  autograd's weight gradient of a stock
    F.linear is ONE library GEMM whose reduction runs over all P rows inside a couple of macro tiles (8.4 ms per training step
    at 2 x 300 k points, profiles/r05_train_step_300k.txt); here it is the chunked batched form the fused linears use
    (fusion_train._weight_grad: row chunks reduced side by side, then summed)."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return x.matmul(weight.t())

    @staticmethod
    def backward(ctx, g):
        from .fusion_train import _weight_grad
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g.matmul(weight) if ctx.needs_input_grad[0] else None
        gw = _weight_grad(g, x) if ctx.needs_input_grad[1] else None
        return gx, gw


class DynamicVFELayer(nn.Module):
    """Linear(no bias) + norm + ReLU (voxel_encoders/utils.py:116-144); parameter holder + torch forward."""

    def __init__(self, in_channels, out_channels, norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01)):
        super().__init__()
        self.fp16_enabled = False
        self.norm = build_norm_layer(norm_cfg, out_channels)[1]
        self.linear = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, inputs):
        if self.training and inputs.is_cuda and torch.is_grad_enabled() and inputs.dtype == torch.float32:
            y = _RowsLinear.apply(inputs.contiguous(), self.linear.weight)
        else:
            y = self.linear(inputs)
        return bn1d_relu(self.norm, y)   # fused BN + ReLU pass in training mode (norm.py)


class DynamicVFE(nn.Module):
    """voxel_encoder.py:287-547, the configuration IS-Fusion uses (cluster + voxel centre, max pooling)."""

    def __init__(self, in_channels=4, feat_channels=[], with_distance=False, with_cluster_center=False,
                 with_voxel_center=False, voxel_size=(0.2, 0.2, 4), point_cloud_range=(0, -40, -3, 70.4, 40, 1),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), mode="max", fusion_layer=None,
                 pointaugmenting=False, fusion_channels=None, return_point_feats=False):
        super().__init__()
        assert mode in ["avg", "max"]
        assert len(feat_channels) > 0
        if fusion_layer is not None:
            raise NotImplementedError("DynamicVFE fusion_layer (MVXNet point fusion) is outside the IS-Fusion path")
        self.raw_in_channels = in_channels
        if with_cluster_center:
            in_channels += 3
        if with_voxel_center:
            in_channels += 3
        if with_distance:
            in_channels += 3
        self.in_channels = in_channels
        self._with_distance = with_distance
        self._with_cluster_center = with_cluster_center
        self._with_voxel_center = with_voxel_center
        self.return_point_feats = return_point_feats
        self.fp16_enabled = False
        self.mode = mode
        self.vx, self.vy, self.vz = voxel_size
        self.x_offset = self.vx / 2 + point_cloud_range[0]
        self.y_offset = self.vy / 2 + point_cloud_range[1]
        self.z_offset = self.vz / 2 + point_cloud_range[2]
        self.voxel_size = list(voxel_size)
        self.point_cloud_range = list(point_cloud_range)
        chans = [self.in_channels] + list(feat_channels)
        layers = []
        for i in range(len(chans) - 1):
            cin = chans[i] * (2 if i > 0 else 1)
            layers.append(DynamicVFELayer(cin, chans[i + 1], norm_cfg))
        self.vfe_layers = nn.ModuleList(layers)
        self.num_vfe = len(layers)
        self.fusion_layer = None
        self.pointaugmenting = pointaugmenting

    # ------------------------------------------------------------------ fused eval path
    def _fusable(self):
        return (not self.training and self.num_vfe == 2 and self._with_cluster_center and self._with_voxel_center
                and not self._with_distance and self.mode == "max" and not self.return_point_feats
                and self.vfe_layers[0].linear.out_features == 64 and self.vfe_layers[1].linear.out_features == 64
                and self.raw_in_channels in (4, 5))

    def _forward_fused(self, features, coors):
        P = features.size(0)
        batch_size = int(coors[-1, 0]) + 1  # same host read as the reference (voxel_encoder.py:429)
        l1, l2 = self.vfe_layers
        s1, b1 = fold_bn(l1.norm)
        s2, b2 = fold_bn(l2.norm)
        w1 = l1.linear.weight.detach().float().contiguous()
        w2 = l2.linear.weight.detach().float().contiguous()
        vf = torch.empty((P, 64), dtype=torch.float32, device=features.device)
        vc = torch.empty((P, 4), dtype=torch.int32, device=features.device)
        n = ctypes.c_int(0)
        lib = _lib.load()
        _lib.check(lib.isf_dynamic_vfe_forward(
            _lib.ptr(features), _lib.ptr(coors), P, features.size(1), batch_size, _lib.f3(self.voxel_size),
            _lib.f6(self.point_cloud_range), _lib.ptr(w1), _lib.ptr(s1), _lib.ptr(b1), 64, _lib.ptr(w2),
            _lib.ptr(s2), _lib.ptr(b2), 64, _lib.ptr(vf), _lib.ptr(vc), None, ctypes.byref(n), _lib.stream()),
            "isf_dynamic_vfe_forward")
        return vf[:n.value], vc[:n.value]

    # ------------------------------------------------------------------ composed (training) path
    def _scatter_batched(self, feats, coors, reduce):
        """DynamicScatter over (b,z,y,x) rows returning the point->voxel map as well."""
        dz = int(coors[:, 1].max()) + 1
        bad = (coors[:, 1:] < 0).any(dim=1)
        folded = torch.stack([coors[:, 0] * dz + coors[:, 1], coors[:, 2], coors[:, 3]], dim=1).int()
        folded[bad] = -1
        with torch.no_grad():
            _, vc, cmap, _ = dynamic_point_to_voxel_forward(feats.detach(), folded, reduce)
        voxel, _ = dynamic_scatter(feats, folded, reduce)  # autograd-aware
        b = torch.div(vc[:, 0], dz, rounding_mode="floor")
        vcoors = torch.stack([b, vc[:, 0] - b * dz, vc[:, 1], vc[:, 2]], dim=1).to(coors.dtype)
        return voxel, vcoors, cmap.long()

    def _forward_composed(self, features, coors):
        feats_ls = [features]
        valid = None
        if self._with_cluster_center:
            voxel_mean, _, cmap = self._scatter_batched(features, coors, "mean")
            valid = cmap >= 0
            points_mean = voxel_mean[cmap.clamp(min=0)]
            feats_ls.append(features[:, :3] - points_mean[:, :3])
        if self._with_voxel_center:
            f_center = features.new_zeros((features.size(0), 3))
            f_center[:, 0] = features[:, 0] - (coors[:, 3].type_as(features) * self.vx + self.x_offset)
            f_center[:, 1] = features[:, 1] - (coors[:, 2].type_as(features) * self.vy + self.y_offset)
            f_center[:, 2] = features[:, 2] - (coors[:, 1].type_as(features) * self.vz + self.z_offset)
            feats_ls.append(f_center)
        if self._with_distance:
            feats_ls.append(torch.norm(features[:, :3], 2, 1, keepdim=True))
        x = torch.cat(feats_ls, dim=-1)
        reduce = "max" if self.mode == "max" else "mean"
        for i, vfe in enumerate(self.vfe_layers):
            point_feats = vfe(x)
            voxel_feats, voxel_coors, cmap = self._scatter_batched(point_feats, coors, reduce)
            if i != self.num_vfe - 1:
                x = torch.cat([point_feats, voxel_feats[cmap.clamp(min=0)]], dim=1)
        if self.return_point_feats:
            return point_feats
        return voxel_feats, voxel_coors

    def forward(self, features, coors, points=None, img_feats=None, img_metas=None, mode=None, **kwargs):
        if mode in ("MVXNet", "AutoAlign"):
            raise NotImplementedError("DynamicVFE mode=%s is dead code on the IS-Fusion path" % mode)
        _lib.require_cuda(features, coors)
        features = features.contiguous().float()
        coors = coors.contiguous().int()
        if self._fusable() and not features.requires_grad:
            return self._forward_fused(features, coors)
        # the reference forces this module to fp32 under mixed precision (@force_fp32, voxel_encoder.py:452): its
        # BatchNorm (naiveSyncBN) asserts fp32 inputs.  Same here under torch.autocast.
        with torch.autocast("cuda", enabled=False):
            return self._forward_composed(features, coors)
