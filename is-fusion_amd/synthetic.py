"""Seeded synthetic nuScenes-shaped inputs: there is no dataset / network here.

``lidar_sweeps(seed, num_points)``: a 32-beam spinning LiDAR (elevations -30.67..+10.67 deg, 1084 azimuth
steps) ray-cast against a ground plane, 40 random axis-aligned boxes and 8 building facades, 10 sweeps with
small ego motion, range-filtered to the point-cloud range, re-sampled to exactly ``num_points`` rows of
(x, y, z, intensity, dt) and shuffled (PointShuffle).  ``uniform_cloud`` is BASELINE config-1's degenerate
uniform-random cloud.
"""
import numpy as np

PC_RANGE = (-54.0, -54.0, -5.0, 54.0, 54.0, 3.0)


def _ray_boxes(origin, dirs, lo, hi):
    """slab test: origin [3], dirs [R,3], boxes lo/hi [M,3] -> nearest positive hit distance [R] (inf = miss)."""
    inv = 1.0 / np.where(np.abs(dirs) < 1e-9, 1e-9, dirs)           # [R,3]
    t0 = (lo[None, :, :] - origin[None, None, :]) * inv[:, None, :]  # [R,M,3]
    t1 = (hi[None, :, :] - origin[None, None, :]) * inv[:, None, :]
    tmin = np.minimum(t0, t1).max(axis=2)
    tmax = np.maximum(t0, t1).min(axis=2)
    hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0.05)
    t = np.where(hit, tmin, np.inf)
    return t.min(axis=1)


def lidar_sweeps(seed, num_points, pc_range=PC_RANGE, num_sweeps=10):
    rng = np.random.default_rng(seed)
    # scene
    nb = 40
    ctr = np.stack([rng.uniform(-50, 50, nb), rng.uniform(-50, 50, nb)], 1)
    size = np.stack([rng.uniform(1.5, 10, nb), rng.uniform(1.5, 3, nb), rng.uniform(1.0, 3.5, nb)], 1)
    ground = -1.84
    lo = np.concatenate([ctr - size[:, :2] / 2, np.full((nb, 1), ground)], 1)
    hi = np.concatenate([ctr + size[:, :2] / 2, ground + size[:, 2:3]], 1)
    nf = 8
    fy = rng.uniform(25, 50, nf) * rng.choice([-1.0, 1.0], nf)
    fx = rng.uniform(-45, 45, nf)
    fh = rng.uniform(5, 20, nf)
    flo = np.stack([fx - fh, fy - 0.5, np.full(nf, ground)], 1)
    fhi = np.stack([fx + fh, fy + 0.5, np.full(nf, ground + 6.0)], 1)
    lo = np.concatenate([lo, flo], 0)
    hi = np.concatenate([hi, fhi], 0)
    # keep the sensor out of every box
    inside = (lo[:, 0] < 3) & (hi[:, 0] > -3) & (lo[:, 1] < 3) & (hi[:, 1] > -3)
    lo, hi = lo[~inside], hi[~inside]
    elev = np.deg2rad(np.linspace(-30.67, 10.67, 32))
    azim = np.linspace(-np.pi, np.pi, 1084, endpoint=False)
    ce, se = np.cos(elev), np.sin(elev)
    dirs = np.stack([np.outer(ce, np.cos(azim)).ravel(), np.outer(ce, np.sin(azim)).ravel(),
                     np.repeat(se, azim.size)], 1)
    pts = []
    for k in range(num_sweeps):
        origin = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), 0.0])
        t = _ray_boxes(origin, dirs, lo, hi)
        tg = np.where(dirs[:, 2] < -1e-6, (ground - origin[2]) / np.minimum(dirs[:, 2], -1e-6), np.inf)
        t = np.minimum(t, tg)
        keep = (t < 70.0) & (rng.random(t.size) > 0.03)
        p = origin[None, :] + dirs[keep] * t[keep, None] + rng.normal(0, 0.01, (int(keep.sum()), 3))
        pts.append(np.concatenate([p, rng.random((p.shape[0], 1)), np.full((p.shape[0], 1), 0.05 * k)], 1))
    pts = np.concatenate(pts, 0)
    r = pc_range
    m = ((pts[:, 0] >= r[0]) & (pts[:, 0] < r[3]) & (pts[:, 1] >= r[1]) & (pts[:, 1] < r[4])
         & (pts[:, 2] >= r[2]) & (pts[:, 2] < r[5]))
    pts = pts[m]
    n = pts.shape[0]
    if n >= num_points:
        pts = pts[rng.choice(n, num_points, replace=False)]
    else:
        extra = pts[rng.integers(0, n, num_points - n)].copy()
        extra[:, :3] += rng.normal(0, 0.02, (extra.shape[0], 3))
        pts = np.concatenate([pts, extra], 0)
    pts = pts[rng.permutation(pts.shape[0])].astype(np.float32)
    # keep the jittered duplicates strictly inside the range (PointsRangeFilter runs before the detector)
    eps = 1e-3
    for j in range(3):
        pts[:, j] = np.clip(pts[:, j], r[j] + eps, r[3 + j] - eps)
    return np.ascontiguousarray(pts)


def uniform_cloud(seed, num_points, pc_range=PC_RANGE, num_features=5):
    rng = np.random.default_rng(seed)
    lo = np.array(pc_range[:3], np.float32)
    hi = np.array(pc_range[3:], np.float32)
    pts = rng.random((num_points, num_features), dtype=np.float32)
    pts[:, :3] = lo + pts[:, :3] * (hi - lo) * np.float32(0.999)
    return np.ascontiguousarray(pts)


def batch(cfg_id, batch_size, num_points, kind="lidar"):
    """SURVEY.md section 8d seeding: seed = 1234 + 1000*cfg + sample_idx."""
    fn = lidar_sweeps if kind == "lidar" else uniform_cloud
    return [fn(1234 + 1000 * cfg_id + i, num_points) for i in range(batch_size)]
