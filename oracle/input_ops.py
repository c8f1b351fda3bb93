"""CPU restatement of the reference's point-cloud INPUT pipeline -- TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py).  numpy / torch on the CPU, each function citing the reference lines it follows; pinned by
tests/golden/input_ref.npz, which tests/golden/make_golden_input.py produced by running the reference's own pipeline
classes (imported from /root/reference through ref_harness.install_pipelines) on seeded sweep files.

The arithmetic types matter for bit-exactness and follow the reference: raw points are float32 [P, 5]
(x, y, z, intensity, ring/time); the sensor2lidar rotation / translation are float64 (as the nuScenes info files hold
them), so numpy computes `p @ R.T` and `+= t` in float64 and rounds to float32 on each assignment; the augmentation of
GlobalRotScaleTransV2 / RandomFlip3DV2 runs in float32 torch ops on the already assembled cloud."""
import numpy as np
import torch


def remove_close(points, radius=1.0):
    """LoadPointsFromMultiSweeps._remove_close (datasets/pipelines/loading.py:824-844): drop points whose |x| AND |y|
    are both below the radius."""
    near = (np.abs(points[:, 0]) < radius) & (np.abs(points[:, 1]) < radius)
    return points[~near]


def assemble_sweeps(key_points, sweeps, timestamp, drop_close=False, radius=1.0):
    """LoadPointsFromMultiSweeps.__call__ after the sweep choice (loading.py:860-903, use_dim = all 5 columns,
    pad_empty_sweeps False).  key_points float32 [P0, 5]; sweeps = list of dict(points float32 [P, 5],
    sensor2lidar_rotation float64 [3, 3], sensor2lidar_translation float64 [3], timestamp (microseconds));
    timestamp = the key frame's, in seconds.  -> float32 [sum P, 5], key frame first, its time column zeroed."""
    key = np.array(key_points, dtype=np.float32, copy=True)
    key[:, 4] = 0
    parts = [key]
    for sw in sweeps:
        p = np.array(sw["points"], dtype=np.float32, copy=True).reshape(-1, 5)
        if drop_close:
            p = remove_close(p, radius)
        lag = timestamp - sw["timestamp"] / 1e6
        p[:, :3] = p[:, :3] @ np.asarray(sw["sensor2lidar_rotation"]).T       # float64 product, rounded on store
        p[:, :3] += np.asarray(sw["sensor2lidar_translation"])                # float64 sum, rounded on store
        p[:, 4] = lag
        parts.append(p)
    return np.concatenate(parts, 0)


def augment(points, rot_mat_T=None, translation=None, scale=None, flip_horizontal=False, flip_vertical=False):
    """The point side of GlobalRotScaleTransV2 (transforms_3d.py:1887-1890: rotate, translate, scale -- BasePoints
    :178, :206, :270) and RandomFlip3DV2 (:1171-1183; LiDARPoints.flip lidar_points.py:29-34), float32 torch ops in
    that order.  points float32 [P, 5] (numpy) -> new array."""
    t = torch.from_numpy(np.array(points, dtype=np.float32, copy=True))
    if rot_mat_T is not None:
        t[:, :3] = t[:, :3] @ torch.as_tensor(rot_mat_T, dtype=torch.float32)
    if translation is not None:
        t[:, :3] += torch.as_tensor(np.asarray(translation), dtype=torch.float32)
    if scale is not None:
        t[:, :3] *= scale
    if flip_horizontal:
        t[:, 1] = -t[:, 1]
    if flip_vertical:
        t[:, 0] = -t[:, 0]
    return t.numpy()


def range_filter(points, point_cloud_range):
    """PointsRangeFilter (transforms_3d.py:2012-2025) with BasePoints.in_range_3d (base_points.py:224-229): STRICT
    inequalities against the float32 range."""
    r = np.asarray(point_cloud_range, dtype=np.float32)
    keep = ((points[:, 0] > r[0]) & (points[:, 1] > r[1]) & (points[:, 2] > r[2]) &
            (points[:, 0] < r[3]) & (points[:, 1] < r[4]) & (points[:, 2] < r[5]))
    return points[keep]


def load_frame(key_points, sweeps, timestamp, point_cloud_range, drop_close=False, aug=None):
    """The point path of one sample through test_pipeline / train_pipeline (configs/isfusion/isfusion_0075voxel.py
    :238-352) up to -- not including -- PointShuffle: assemble, augment, range-filter."""
    pts = assemble_sweeps(key_points, sweeps, timestamp, drop_close)
    if aug:
        pts = augment(pts, **aug)
    return range_filter(pts, point_cloud_range)
