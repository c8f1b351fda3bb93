"""Host side of the input pre-pass: the point path of the reference's data pipeline --
LoadPointsFromFile + LoadPointsFromMultiSweeps (datasets/pipelines/loading.py:1345-1516, :735-907), the point side of
GlobalRotScaleTransV2 / RandomFlip3DV2 (datasets/pipelines/transforms_3d.py:1871-1915, :1163-1204),
PointsRangeFilter (:2002-2037) and PointShuffle (:1918-1943) -- for a whole batch on the GPU.

The host only reads the sweep files into ONE pinned buffer (no per-point work on the CPU) and fills one descriptor per
file; libisf_hip.so (isf_assemble_points) applies time column / remove_close / sensor pose / augmentation / range
filter and compacts per sample in the reference's order.  No CPU fallback: without a GPU this raises."""
import ctypes

import numpy as np
import torch

from . import _lib

POINT_DIM = 5


def rotation_matrix_T(angle):
    """rot_mat_T of BasePoints.rotate(angle) about the z axis (core/points/base_points.py:156-173, float32 sin / cos as
    the reference computes them)."""
    a = torch.tensor(float(angle), dtype=torch.float32)
    s, c = torch.sin(a), torch.cos(a)
    return torch.tensor([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=torch.float32).T.contiguous().numpy()


def draw_train_aug(resize_lim=(0.9, 1.1), rot_lim=(-0.78539816, 0.78539816), trans_lim=0.5, flip=True):
    """One draw of the training augmentation in the reference's numpy RNG order: GlobalRotScaleTransV2 (scale, theta,
    3 x normal; transforms_3d.py:1882-1884; points are rotated by -theta, :1888) then RandomFlip3DV2 (two
    np.random.choice draws, :1167-1168)."""
    scale = np.random.uniform(*resize_lim)
    theta = np.random.uniform(*rot_lim)
    translation = np.array([np.random.normal(0, trans_lim) for _ in range(3)])
    fh = fv = 0
    if flip:
        fh, fv = int(np.random.choice([0, 1])), int(np.random.choice([0, 1]))
    return dict(rot_mat_T=rotation_matrix_T(-theta), translation=translation, scale=scale,
                flip_horizontal=bool(fh), flip_vertical=bool(fv))


class MultiSweepPointLoader:
    """Batch replacement for the LoadPointsFromFile -> LoadPointsFromMultiSweeps -> [GlobalRotScaleTransV2 ->
    RandomFlip3DV2] -> PointsRangeFilter [-> PointShuffle] chain of configs/isfusion/isfusion_0075voxel.py:238-352.
    Constructor arguments carry the reference's names (LoadPointsFromMultiSweeps: sweeps_num, remove_close,
    test_mode; PointsRangeFilter: point_cloud_range)."""

    def __init__(self, sweeps_num=10, remove_close=False, test_mode=False, point_cloud_range=None, close_radius=1.0,
                 shuffle=False, device="cuda"):
        self.sweeps_num, self.remove_close, self.test_mode = sweeps_num, remove_close, test_mode
        self.point_cloud_range = None if point_cloud_range is None else [float(v) for v in point_cloud_range]
        self.close_radius, self.shuffle = float(close_radius), shuffle
        self.device = torch.device(device)
        self._pinned = None

    # ------------------------------------------------------------------------------------------------ host side
    @staticmethod
    def _read(src):
        """a sweep file path (flat float32, loading.py:797) or an already loaded array -> flat float32 view"""
        if isinstance(src, (str, bytes)):
            return np.fromfile(src, dtype=np.float32)
        return np.ascontiguousarray(src, dtype=np.float32).reshape(-1)

    def _choose(self, num):
        """sweep choice of loading.py:871-877"""
        if num <= self.sweeps_num:
            return np.arange(num)
        if self.test_mode:
            return np.arange(self.sweeps_num)
        return np.random.choice(num, self.sweeps_num, replace=False)

    def _stage(self, arrays):
        total = sum(a.size for a in arrays)
        if self._pinned is None or self._pinned.numel() < total:
            self._pinned = torch.empty(max(total, 1), dtype=torch.float32).pin_memory()
        host = self._pinned[:total]
        view, at = host.numpy(), 0
        for a in arrays:
            view[at:at + a.size] = a
            at += a.size
        return host

    def __call__(self, results_list, aug=None):
        """results_list: one dict per sample with the reference's keys -- 'pts_filename' (path or float32 [P, 5]
        array), 'timestamp' (s), 'sweeps': [dict(data_path (path or array), timestamp (us), sensor2lidar_rotation,
        sensor2lidar_translation)].  aug: None or one dict per sample (draw_train_aug).
        -> list of float32 [N_b, 5] device tensors (views of one buffer), in the reference's point order."""
        if self.device.type != "cuda":
            raise _lib.IsfError("MultiSweepPointLoader runs on the GPU only (isf_assemble_points); no CPU fallback")
        lib = _lib.load()
        B = len(results_list)
        arrays, descs, row = [], [], 0

        def add(arr, sample, is_sweep, lag=0.0, rot=None, trans=None):
            nonlocal row
            assert arr.size % POINT_DIM == 0, "sweep files hold float32 [P, 5]"
            d = _lib.Sweep()
            d.first_point, d.num_points, d.sample, d.is_sweep = row, arr.size // POINT_DIM, sample, int(is_sweep)
            d.remove_close, d.close_radius = int(self.remove_close and is_sweep), self.close_radius
            d.time_lag = float(np.float32(lag))
            r = np.eye(3) if rot is None else np.asarray(rot, dtype=np.float64)
            t = np.zeros(3) if trans is None else np.asarray(trans, dtype=np.float64)
            d.rotation = (ctypes.c_double * 9)(*r.reshape(-1))
            d.translation = (ctypes.c_double * 3)(*t.reshape(-1))
            arrays.append(arr)
            descs.append(d)
            row += d.num_points

        for b, res in enumerate(results_list):
            add(self._read(res["pts_filename"]), b, False)
            sweeps = res.get("sweeps", [])
            for i in self._choose(len(sweeps)):
                sw = sweeps[int(i)]
                add(self._read(sw["data_path"]), b, True, res["timestamp"] - sw["timestamp"] / 1e6,
                    sw["sensor2lidar_rotation"], sw["sensor2lidar_translation"])
        raw = self._stage(arrays).to(self.device, non_blocking=True)
        out = torch.empty((max(row, 1), POINT_DIM), dtype=torch.float32, device=self.device)
        offsets = torch.empty((B + 1,), dtype=torch.int32, device=self.device)
        host_offsets = (ctypes.c_int32 * (B + 1))()
        sw_arr = (_lib.Sweep * len(descs))(*descs)
        aug_arr = None
        if aug is not None:
            aug_arr = (_lib.PointAug * B)()
            for b, a in enumerate(aug):
                if not a:
                    continue
                g = aug_arr[b]
                g.enabled = 1
                g.rot_mat_T = (ctypes.c_float * 9)(*np.asarray(a.get("rot_mat_T", np.eye(3)), np.float32).reshape(-1))
                g.translation = (ctypes.c_float * 3)(*np.asarray(a.get("translation", np.zeros(3)), np.float32))
                g.scale = float(a.get("scale", 1.0))
                g.flip_horizontal, g.flip_vertical = int(a.get("flip_horizontal", 0)), int(a.get("flip_vertical", 0))
        rng = None
        if self.point_cloud_range is not None:
            rng = (ctypes.c_float * 6)(*self.point_cloud_range)
        with torch.cuda.device(self.device):
            _lib.check(lib.isf_assemble_points(_lib.ptr(raw), sw_arr, len(descs), B, aug_arr, rng, _lib.ptr(out),
                                               _lib.ptr(offsets), host_offsets, _lib.stream()), "isf_assemble_points")
        pts = [out[host_offsets[b]:host_offsets[b + 1]] for b in range(B)]
        if self.shuffle:        # PointShuffle: BasePoints.shuffle = tensor[randperm] (base_points.py, torch RNG)
            pts = [p[torch.randperm(p.shape[0], device=p.device)] for p in pts]
        return pts
