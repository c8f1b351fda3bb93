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


def camera_matrices(batch_size, num_cam=6, image_hw=(384, 1056), seed=0):
    """SURVEY.md section 8d cfg 3: num_cam pinhole cameras at 360/num_cam degree yaw spacing, fx = fy = 1266*0.48,
    principal point at the image centre, mounted 1.5 m above the LiDAR origin.  -> (lidar2img [B,cam,4,4],
    img_aug [B,cam,4,4] = small scale/shift augmentation, lidar_aug [B,4,4] = small yaw rotation + translation)."""
    rng = np.random.default_rng(seed)
    H, W = image_hw
    f = 1266.0 * 0.48
    K = np.array([[f, 0, W / 2, 0], [0, f, H / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)
    l2i = np.zeros((batch_size, num_cam, 4, 4), np.float32)
    for c in range(num_cam):
        yaw = 2 * np.pi * c / num_cam
        # camera axes in the lidar frame: z forward (cos yaw, sin yaw, 0), x right, y down
        fwd = np.array([np.cos(yaw), np.sin(yaw), 0.0])
        right = np.array([np.sin(yaw), -np.cos(yaw), 0.0])
        down = np.array([0.0, 0.0, -1.0])
        R = np.stack([right, down, fwd], 0)          # lidar -> camera
        t = -R @ np.array([0.0, 0.0, 1.5])
        E = np.eye(4)
        E[:3, :3], E[:3, 3] = R, t
        l2i[:, c] = (K @ E).astype(np.float32)
    img_aug = np.tile(np.eye(4, dtype=np.float32), (batch_size, num_cam, 1, 1))
    lidar_aug = np.tile(np.eye(4, dtype=np.float32), (batch_size, 1, 1))
    for b in range(batch_size):
        s = 1.0 + 0.05 * rng.standard_normal()
        img_aug[b, :, 0, 0] = img_aug[b, :, 1, 1] = s
        img_aug[b, :, 0, 3], img_aug[b, :, 1, 3] = 4.0 * rng.standard_normal(), 3.0 * rng.standard_normal()
        a = 0.1 * rng.standard_normal()
        lidar_aug[b, :2, :2] = [[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]]
        lidar_aug[b, :3, 3] = 0.3 * rng.standard_normal(3)
    return l2i, img_aug, lidar_aug


def fusion_inputs(seed, batch_size, bev_size=180, num_pillars=1500, num_cam=6, image_hw=(384, 1056), embed=256):
    """Random inputs of ISFusionEncoder.forward (BASELINE config 3: camera features random, LiDAR BEV random):
    img feats (two FPN levels, stride 8 / 16), lidar BEV features, pillars (T = 12 slots, zero padded),
    pillar coords (b, 0, y, x) unique per sample, camera / augmentation matrices."""
    rng = np.random.default_rng(seed)
    H, W = image_hw
    B = batch_size
    img0 = rng.standard_normal((B * num_cam, embed, H // 8, W // 8), dtype=np.float32)
    img1 = rng.standard_normal((B * num_cam, embed, H // 16, W // 16), dtype=np.float32)
    lidar = (rng.standard_normal((B, 2 * embed, bev_size, bev_size), dtype=np.float32) * 0.5)
    lidar *= (rng.random((B, 1, bev_size, bev_size)) < 0.35)  # sparse like a real BEV map
    coors, pillars = [], []
    cell = 108.0 / bev_size
    for b in range(B):
        lin = rng.choice(bev_size * bev_size, min(num_pillars, bev_size * bev_size), replace=False)
        y, x = lin // bev_size, lin % bev_size
        coors.append(np.stack([np.full_like(y, b), np.zeros_like(y), y, x], 1))
        n = rng.integers(1, 13, len(lin))
        p = np.zeros((len(lin), 12, 5), np.float32)
        cx, cy = -54.0 + (x + 0.5) * cell, -54.0 + (y + 0.5) * cell
        for t in range(12):
            m = n > t
            p[m, t, 0] = cx[m] + rng.uniform(-cell / 2, cell / 2, m.sum())
            p[m, t, 1] = cy[m] + rng.uniform(-cell / 2, cell / 2, m.sum())
            p[m, t, 2] = rng.uniform(-3.0, 1.0, m.sum())
            p[m, t, 3:] = rng.random((m.sum(), 2))
        pillars.append(p)
    l2i, img_aug, lidar_aug = camera_matrices(B, num_cam, image_hw, seed + 1)
    return dict(img_feats=(img0, img1), lidar_feats=lidar.astype(np.float32),
                pillars=np.concatenate(pillars).astype(np.float32),
                pillar_coors=np.concatenate(coors).astype(np.int32), lidar2img=l2i, img_aug_matrix=img_aug,
                lidar_aug_matrix=lidar_aug, input_shape=tuple(image_hw))
