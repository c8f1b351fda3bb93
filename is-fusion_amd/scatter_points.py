"""DynamicScatter (drop-in for ``mmdet3d.ops.DynamicScatter`` / ``dynamic_scatter``).

Mirrors mmdet3d/ops/voxel/scatter_points.py:9-105 including the autograd contract (forward saves
feats / voxel_feats / point2voxel_map / count; backward routes to the HIP backward kernel).
"""
import ctypes

import torch
from torch import nn
from torch.autograd import Function

from . import _lib

# the HIP kernels compute in fp32: under torch.autocast (the reference trains with mixed precision) inputs are cast
# to fp32 on the way in and autocast is off inside forward / backward
_amp_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")


def dynamic_point_to_voxel_forward(feats, coors, reduce_type):
    """-> [reduced_feats, out_coors, coors_map int32, reduce_count int32] (voxelization.h:108-121)."""
    _lib.require_cuda(feats, coors)
    if reduce_type not in _lib.REDUCE:
        raise RuntimeError("do not support reduce type " + str(reduce_type))
    P, C = feats.size(0), feats.size(1)
    if P == 0:  # scatter_points_cuda.cu:193-197
        return [feats.clone().detach(), coors.clone().detach(),
                coors.new_empty((0,), dtype=torch.int32), coors.new_empty((0,), dtype=torch.int32)]
    feats_c = feats.contiguous().float()
    coors_c = coors.contiguous().int()
    red = torch.empty((P, C), dtype=torch.float32, device=feats.device)
    out_coors = torch.empty((P, 3), dtype=torch.int32, device=feats.device)
    cmap = torch.empty((P,), dtype=torch.int32, device=feats.device)
    cnt = torch.empty((P,), dtype=torch.int32, device=feats.device)
    m = ctypes.c_int(0)
    lib = _lib.load()
    _lib.check(lib.isf_dynamic_point_to_voxel_forward(
        _lib.ptr(feats_c), _lib.ptr(coors_c), P, C, _lib.REDUCE[reduce_type], _lib.ptr(red),
        _lib.ptr(out_coors), _lib.ptr(cmap), _lib.ptr(cnt), ctypes.byref(m), _lib.stream()),
        "isf_dynamic_point_to_voxel_forward")
    M = m.value
    return [red[:M], out_coors[:M], cmap, cnt[:M]]


def dynamic_point_to_voxel_backward(grad_feats, grad_reduced_feats, feats, reduced_feats, coors_idx,
                                    reduce_count, reduce_type):
    """In-place into grad_feats (voxelization.h:123-140)."""
    _lib.require_cuda(grad_feats)
    P, C = feats.size(0), feats.size(1)
    M = reduced_feats.size(0)
    if P == 0:
        return
    lib = _lib.load()
    grf, f, rf = grad_reduced_feats.contiguous(), feats.contiguous(), reduced_feats.contiguous()   # outlive the C call
    _lib.check(lib.isf_dynamic_point_to_voxel_backward(
        _lib.ptr(grad_feats), _lib.ptr(grf), _lib.ptr(f), _lib.ptr(rf), _lib.ptr(coors_idx), _lib.ptr(reduce_count), P, M, C,
        _lib.REDUCE[reduce_type], _lib.stream()), "isf_dynamic_point_to_voxel_backward")


class _dynamic_scatter(Function):

    @staticmethod
    @_amp_fwd
    def forward(ctx, feats, coors, reduce_type="max"):
        voxel_feats, voxel_coors, point2voxel_map, voxel_points_count = dynamic_point_to_voxel_forward(
            feats, coors, reduce_type)
        ctx.reduce_type = reduce_type
        ctx.save_for_backward(feats, voxel_feats, point2voxel_map, voxel_points_count)
        ctx.mark_non_differentiable(voxel_coors)
        return voxel_feats, voxel_coors

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_voxel_feats, grad_voxel_coors=None):
        feats, voxel_feats, point2voxel_map, voxel_points_count = ctx.saved_tensors
        grad_feats = torch.zeros_like(feats)
        dynamic_point_to_voxel_backward(grad_feats, grad_voxel_feats.contiguous(), feats, voxel_feats,
                                        point2voxel_map, voxel_points_count, ctx.reduce_type)
        return grad_feats, None, None


dynamic_scatter = _dynamic_scatter.apply


class DynamicScatter(nn.Module):
    """Same constructor / forward as mmdet3d.ops.DynamicScatter (scatter_points.py:52-105)."""

    def __init__(self, voxel_size, point_cloud_range, average_points: bool):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.average_points = average_points

    def forward_single(self, points, coors):
        reduce = "mean" if self.average_points else "max"
        return dynamic_scatter(points.contiguous(), coors.contiguous(), reduce)

    def forward(self, points, coors):
        if coors.size(-1) == 3:
            return self.forward_single(points, coors)
        # (b,z,y,x) rows: the reference loops over samples and concatenates (scatter_points.py:82-96), which
        # equals one scatter over rows sorted by (b,z,y,x).  Fold b into z (z' = b*Dz + z) so a single
        # 3-column call yields that order, then unfold.  Rows with a negative z/y/x stay invalid.
        if coors.size(0) == 0:
            return points.clone(), coors.clone()
        dz = int(coors[:, 1].max()) + 1
        if dz <= 0:
            return points.new_zeros((0, points.size(1))), coors.new_zeros((0, 4))
        bad = (coors[:, 1:] < 0).any(dim=1)
        folded = torch.stack([coors[:, 0] * dz + coors[:, 1], coors[:, 2], coors[:, 3]], dim=1)
        folded[bad] = -1
        voxel, vc = self.forward_single(points, folded.int())
        b = torch.div(vc[:, 0], dz, rounding_mode="floor")
        return voxel, torch.stack([b, vc[:, 0] - b * dz, vc[:, 1], vc[:, 2]], dim=1).to(coors.dtype)

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range="
                f"{self.point_cloud_range}, average_points={self.average_points})")
