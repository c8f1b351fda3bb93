"""Voxelization (drop-in for ``mmdet3d.ops.Voxelization`` / ``voxelization``).

Mirrors mmdet3d/ops/voxel/voxelize.py:10-148: same constructor arguments, same ``forward`` return
conventions (dynamic: coors [P,3] int32 (z,y,x); hard: (voxels, coors, num_points_per_voxel)).
The arithmetic runs in libisf_hip.so (isf_dynamic_voxelize / isf_hard_voxelize).
"""
import ctypes

import torch
from torch import nn
from torch.nn.modules.utils import _pair

from . import _lib


def dynamic_voxelize(points, voxel_size, coors_range):
    """voxel_layer.dynamic_voxelize: points [P,C] -> coors [P,3] int32 (z,y,x), invalid rows -1."""
    _lib.require_cuda(points)
    points = points.contiguous().float()
    coors = torch.empty((points.size(0), 3), dtype=torch.int32, device=points.device)
    lib = _lib.load()
    _lib.check(lib.isf_dynamic_voxelize(_lib.ptr(points), points.size(0), points.size(1),
                                        _lib.f3(voxel_size), _lib.f6(coors_range), _lib.ptr(coors),
                                        _lib.stream()), "isf_dynamic_voxelize")
    return coors


def dynamic_voxelize_batched(points_list, voxel_size, coors_range):
    """ISFusionDetector.dynamic_voxelize (detectors/isfusion.py:123-146):
    list of [P_i,C] -> (points [sum P,C], coors [sum P,4] (b,z,y,x))."""
    points = torch.cat(points_list, dim=0).contiguous().float()
    _lib.require_cuda(points)
    offs = [0]
    for p in points_list:
        offs.append(offs[-1] + p.size(0))
    coors = torch.empty((points.size(0), 4), dtype=torch.int32, device=points.device)
    lib = _lib.load()
    arr = (ctypes.c_int64 * len(offs))(*offs)
    _lib.check(lib.isf_dynamic_voxelize_batched(_lib.ptr(points), arr, len(points_list), points.size(1),
                                                _lib.f3(voxel_size), _lib.f6(coors_range), _lib.ptr(coors),
                                                _lib.stream()), "isf_dynamic_voxelize_batched")
    return points, coors


def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    """voxel_layer.hard_voxelize (deterministic) -> (voxels[M,T,C], coors[M,3], num_points[M])."""
    _lib.require_cuda(points)
    points = points.contiguous().float()
    # outputs pre-allocated at max size and zero-filled, exactly like voxelize.py:57-61
    voxels = points.new_zeros((max_voxels, max_points, points.size(1)))
    coors = points.new_zeros((max_voxels, 3), dtype=torch.int32)
    num = points.new_zeros((max_voxels,), dtype=torch.int32)
    n = ctypes.c_int(0)
    lib = _lib.load()
    _lib.check(lib.isf_hard_voxelize(_lib.ptr(points), points.size(0), points.size(1),
                                     _lib.f3(voxel_size), _lib.f6(coors_range), int(max_points),
                                     int(max_voxels), _lib.ptr(voxels), _lib.ptr(coors), _lib.ptr(num),
                                     ctypes.byref(n), _lib.stream()), "isf_hard_voxelize")
    m = n.value
    return voxels[:m], coors[:m], num[:m]


class PendingVoxels:
    """Hard voxelization queued without a host wait (isf_hard_voxelize_device): capacity-sized outputs, the voxel count in
    device memory and on its way to pinned host memory.  `result()` -- call it when something else has kept the host busy
    in between -- waits for that copy (normally long finished) and returns the reference's (voxels[:M], coors[:M],
    num_points[:M])."""

    def __init__(self, voxels, coors, num, count_host, event):
        self.voxels, self.coors, self.num, self._count_host, self._event = voxels, coors, num, count_host, event

    def result(self):
        self._event.synchronize()
        m = int(self._count_host[0])
        return self.voxels[:m], self.coors[:m], self.num[:m]


def hard_voxelize_async(points, voxel_size, coors_range, max_points, max_voxels):
    """hard_voxelize with a DEVICE-RESIDENT count: every kernel is queued on the current stream, nothing waits on the host
    (the reference reads voxel_num back synchronously, voxelization_cuda.cu:366-371).  The outputs are allocated at
    capacity and not zero-filled -- the kernel writes the padding slots of every voxel it emits.  -> PendingVoxels."""
    _lib.require_cuda(points)
    points = points.contiguous().float()
    voxels = points.new_empty((max_voxels, max_points, points.size(1)))
    coors = points.new_empty((max_voxels, 3), dtype=torch.int32)
    num = points.new_empty((max_voxels,), dtype=torch.int32)
    count = points.new_empty((1,), dtype=torch.int32)
    _lib.check(_lib.load().isf_hard_voxelize_device(
        _lib.ptr(points), points.size(0), points.size(1), _lib.f3(voxel_size), _lib.f6(coors_range), int(max_points),
        int(max_voxels), _lib.ptr(voxels), _lib.ptr(coors), _lib.ptr(num), _lib.ptr(count), _lib.stream()),
        "isf_hard_voxelize_device")
    count_host = torch.empty((1,), dtype=torch.int32, pin_memory=True)
    count_host.copy_(count, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return PendingVoxels(voxels, coors, num, count_host, ev)


class PendingBatchVoxels:
    """hard_voxelize_batched_async's handle: result() -> (voxels [M, T, C], num_points [M], coors [M, 4] = (sample, z, y,
    x)), M = the samples' voxels one behind the other -- the triple ISFusionDetector.voxelize returns for pillars."""

    def __init__(self, voxels, coors4, num, count_host, event, batch_size):
        self.voxels, self.coors4, self.num = voxels, coors4, num
        self._count_host, self._event, self.batch_size = count_host, event, batch_size

    def result(self):
        self._event.synchronize()
        m = int(self._count_host[self.batch_size])
        return self.voxels[:m], self.num[:m], self.coors4[:m]

    def counts(self):
        self._event.synchronize()
        return [int(v) for v in self._count_host[:self.batch_size]]


def hard_voxelize_batched_async(points_list, voxel_size, coors_range, max_points, max_voxels):
    """The samples of a batch voxelized in one pass (isf_hard_voxelize_batched_device): per sample exactly
    hard_voxelize_async's voxels in the same order, concatenated, with (sample, z, y, x) coordinates; one set of launches
    for the batch instead of one per sample, nothing waits on the host.  -> PendingBatchVoxels."""
    B = len(points_list)
    _lib.require_cuda(*points_list)
    pts = torch.cat([p.contiguous().float() for p in points_list], 0) if B > 1 else points_list[0].contiguous().float()
    offs = [0]
    for p in points_list:
        offs.append(offs[-1] + p.size(0))
    rows = B * int(max_voxels)
    voxels = pts.new_empty((rows, max_points, pts.size(1)))
    coors4 = pts.new_empty((rows, 4), dtype=torch.int32)
    num = pts.new_empty((rows,), dtype=torch.int32)
    count = pts.new_empty((B + 1,), dtype=torch.int32)
    _lib.check(_lib.load().isf_hard_voxelize_batched_device(
        _lib.ptr(pts), (ctypes.c_int64 * (B + 1))(*offs), B, pts.size(1), _lib.f3(voxel_size), _lib.f6(coors_range),
        int(max_points), int(max_voxels), _lib.ptr(voxels), _lib.ptr(coors4), _lib.ptr(num), _lib.ptr(count), _lib.stream()),
        "isf_hard_voxelize_batched_device")
    count_host = torch.empty((B + 1,), dtype=torch.int32, pin_memory=True)
    count_host.copy_(count, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return PendingBatchVoxels(voxels, coors4, num, count_host, ev, B)


def voxelization(points, voxel_size, coors_range, max_points=35, max_voxels=20000, deterministic=True):
    """Functional form (voxelize.py:10-76).  ``deterministic=False`` selects the same deterministic kernel:
    the non-deterministic CUDA variant exists only to dodge the O(P^2) kernel this build does not have."""
    if max_points == -1 or max_voxels == -1:
        return dynamic_voxelize(points, voxel_size, coors_range)
    return hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels)


class Voxelization(nn.Module):
    """Same constructor / forward as mmdet3d.ops.Voxelization (voxelize.py:79-148)."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, deterministic=True):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.max_num_points = max_num_points
        self.max_voxels = max_voxels if isinstance(max_voxels, tuple) else _pair(max_voxels)
        self.deterministic = deterministic
        pcr = torch.tensor(point_cloud_range, dtype=torch.float32)
        vs = torch.tensor(voxel_size, dtype=torch.float32)
        grid_size = torch.round((pcr[3:] - pcr[:3]) / vs).long()
        self.grid_size = grid_size
        self.pcd_shape = [*grid_size[:2], 1][::-1]

    def forward(self, input):
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        return voxelization(input, self.voxel_size, self.point_cloud_range, self.max_num_points, max_voxels,
                            self.deterministic)

    def forward_async(self, input):
        """forward() queued without a host wait -> PendingVoxels (hard voxelization only)."""
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        assert self.max_num_points != -1 and max_voxels != -1, "dynamic voxelization has no count to wait for"
        return hard_voxelize_async(input, self.voxel_size, self.point_cloud_range, self.max_num_points, max_voxels)

    def forward_batch_async(self, inputs):
        """the samples of a batch in one pass -> PendingBatchVoxels (hard voxelization only, <= 16 samples)"""
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        assert self.max_num_points != -1 and max_voxels != -1, "dynamic voxelization has no count to wait for"
        return hard_voxelize_batched_async(inputs, self.voxel_size, self.point_cloud_range, self.max_num_points, max_voxels)

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range="
                f"{self.point_cloud_range}, max_num_points={self.max_num_points}, max_voxels="
                f"{self.max_voxels}, deterministic={self.deterministic})")
