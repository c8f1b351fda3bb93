"""Seeded inputs for the input pre-pass tests, shared by tests/golden/make_golden_input.py and the
parity tests: synthetic nuScenes-shaped sweep files (float32 [P, 5]) with float64 sensor2lidar poses."""
import numpy as np

PC_RANGE = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]

# name -> (seed, key-frame points, points per previous sweep)
INPUT_CONFIGS = {
    "a": (11, 1500, (1000, 900, 1100)),
    "b": (12, 1200, (800, 1000)),
    "key_only": (13, 700, ()),
}
# reference pipeline variants the goldens cover
INPUT_CASES = ("test", "remove_close", "train_aug")


def sweep_inputs(seed, key_n, sweep_ns):
    """-> key_points float32 [key_n, 5], sweeps [dict(points, sensor2lidar_rotation f64 [3,3],
    sensor2lidar_translation f64 [3], timestamp us)], timestamp (s).  Points spill over PC_RANGE on every side and
    crowd the sensor origin so that the range filter and remove_close both drop some."""
    rng = np.random.default_rng(seed)

    def cloud(n):
        p = np.empty((n, 5), np.float32)
        p[:, 0] = rng.uniform(-60, 60, n)
        p[:, 1] = rng.uniform(-60, 60, n)
        p[:, 2] = rng.uniform(-6, 4, n)
        near = rng.random(n) < 0.08
        p[near, :2] = rng.uniform(-1.5, 1.5, (int(near.sum()), 2))
        p[:, 3] = rng.integers(0, 256, n)
        p[:, 4] = rng.integers(0, 32, n)
        return p

    ts_us = 1533151603547590 + int(rng.integers(0, 10**6))
    sweeps = []
    for k, n in enumerate(sweep_ns):
        yaw, pitch = rng.uniform(-0.05, 0.05), rng.uniform(-0.01, 0.01)
        cz, sz, cy, sy = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        rot = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]]) @ np.array([[cy, 0, sy], [0, 1.0, 0], [-sy, 0, cy]])
        sweeps.append(dict(points=cloud(n), sensor2lidar_rotation=rot,
                           sensor2lidar_translation=rng.uniform(-1.0, 1.0, 3) * np.array([1.0, 1.0, 0.05]),
                           timestamp=ts_us - (k + 1) * 50000 - int(rng.integers(0, 2000))))
    return cloud(key_n), sweeps, ts_us / 1e6


def train_aug(seed):
    """One draw of GlobalRotScaleTransV2(is_train=True; transforms_3d.py:1882-1890) + RandomFlip3DV2 (:1167-1183)
    parameters, as the golden script replays them through the reference with the same numpy seed.
    -> dict for oracle.input_ops.augment / isfusion_amd.input_pipeline (rot_mat_T from LiDARPoints.rotate(-theta),
    base_points.py:156-173 with axis 2)."""
    import torch
    st = np.random.RandomState(seed)
    scale = st.uniform(0.9, 1.1)
    theta = st.uniform(-0.78539816, 0.78539816)
    translation = np.array([st.normal(0, 0.5) for _ in range(3)])
    flip_h, flip_v = int(st.choice([0, 1])), int(st.choice([0, 1]))
    ang = torch.tensor(-theta, dtype=torch.float32)
    s, c = torch.sin(ang), torch.cos(ang)
    rot_mat_T = torch.tensor([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=torch.float32).T.contiguous()
    return dict(rot_mat_T=rot_mat_T.numpy(), translation=translation, scale=scale, flip_horizontal=bool(flip_h),
                flip_vertical=bool(flip_v))
